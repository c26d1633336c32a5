"""CPU stand-in for GpuStripeEngine, backed by the oracle (test infrastructure): lets the
multi-process stripe orchestration run under gloo without a GPU."""
import ctypes as C

import numpy as np
import torch

from oracle import Oracle
from oracle.pyoracle import MODE, _Res
from sequencealigner_amd.engine import AlignResult

EVENT_FN = C.CFUNCTYPE(C.c_int, C.c_void_p, C.c_int, C.c_int64)
KIND = {"nw": 0, "sw": 1, "lcs": 2}


class OracleStripeEngine:
    def __init__(self):
        self.orc = Oracle()
        L = self.orc.lib
        vp = C.c_void_p
        L.orc_stripe_fill.argtypes = [C.c_int, vp, C.c_int64, vp, C.c_int64, C.c_int, C.c_double, C.c_double, C.c_int64,
                                      C.c_int64, C.c_int, C.c_int, vp, vp, C.c_int64, EVENT_FN, vp, C.POINTER(vp),
                                      C.POINTER(C.c_int64)]
        L.orc_stripe_fill.restype = C.c_int
        L.orc_stripe_traceback.argtypes = [vp, C.c_int64, C.c_int64, C.POINTER(_Res)]
        L.orc_stripe_traceback.restype = C.c_int
        L.orc_stripe_free.argtypes = [vp]
        L.orc_stripe_best.argtypes = [vp, C.POINTER(C.c_int64), C.POINTER(C.c_int64), C.POINTER(C.c_int64)]
        L.orc_stripe_best.restype = None
        self.handle = None
        self.fail_next = False  # tests: refuse the next stripe_fill before posting anything (as SA_ERR_NOMEM would)

    def alloc_border(self, m):
        return torch.zeros((m + 1, 2), dtype=torch.int32)

    def small(self, values, dtype=None):
        return torch.tensor(values, dtype=dtype or torch.int64)

    def bytes_tensor(self, arr):
        return torch.from_numpy(np.ascontiguousarray(arr).copy())

    def empty_bytes(self, n):
        return torch.empty(n, dtype=torch.uint8)

    def stripe_fill(self, kind, x, y_stripe, n_total, col_begin, is_first, is_last, left_ptr, right_ptr, chunk_rows,
                    on_event, mode="dna", gap_open=-5.0, gap_extend=-1.0):
        x = np.ascontiguousarray(x, np.uint8)
        y = np.ascontiguousarray(y_stripe, np.uint8)
        if self.fail_next:
            self.fail_next = False
            raise RuntimeError("injected failure before any transfer was posted")
        err = []

        def _cb(_u, ev, k):
            try:
                on_event(ev, k)
                return 0
            except BaseException as e:
                err.append(e)
                return 1

        cb = EVENT_FN(_cb)
        h = C.c_void_p()
        score = C.c_int64()
        if self.handle:
            self.orc.lib.orc_stripe_free(self.handle)
        rc = self.orc.lib.orc_stripe_fill(KIND[kind], x.ctypes.data, len(x), y.ctypes.data, len(y), MODE[mode], gap_open,
                                          gap_extend, n_total, col_begin, int(is_first), int(is_last), left_ptr, right_ptr,
                                          chunk_rows, cb, None, C.byref(h), C.byref(score))
        if err:
            raise err[0]
        assert rc == 0, rc
        self.handle = h
        end = (len(x), len(y))
        if kind == "sw":
            b, bi, bj = C.c_int64(), C.c_int64(), C.c_int64()
            self.orc.lib.orc_stripe_best(h, C.byref(b), C.byref(bi), C.byref(bj))
            end = (int(bi.value), int(bj.value))
        return AlignResult(int(score.value), end, (0, 0), np.zeros(0, np.uint8))

    def stripe_traceback(self, i, j):
        r = _Res()
        rc = self.orc.lib.orc_stripe_traceback(self.handle, i, j, C.byref(r))
        assert rc == 0
        ops = np.ctypeslib.as_array(r.ops, shape=(r.n_ops,)).copy() if r.n_ops else np.zeros(0, np.uint8)
        out = AlignResult(0, (i, j), (int(r.start_i), int(r.start_j)), ops)
        self.orc.lib.orc_free(C.byref(r))
        return out

    def whole(self, kind, x, y, mode, go, ge):
        o = self.orc.lcs(x, y) if kind == "lcs" else getattr(self.orc, kind)(x, y, mode, go, ge)
        return AlignResult(o.score, o.end, o.start, o.ops)
