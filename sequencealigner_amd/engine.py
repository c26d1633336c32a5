"""ctypes binding of include/seqalign_hip.h.

Function names mirror the reference's hot-path entry points (src/main.cpp):
  globalalign(x, y, mode, gap_open, gap_extend)  <- globalalign(), main.cpp:691
  localalign(...)                                <- localalign(),  main.cpp:1037
  lcs(x, y)                                      <- lcs(),         main.cpp:1562
Each returns an AlignResult (score, end/start cell, traceback ops) instead of writing
files; the C++ CLI (csrc/aligner_main.cpp) owns the byte-exact writers.
"""
from __future__ import annotations

import ctypes as C
from dataclasses import dataclass, field
from pathlib import Path
from typing import Optional

import numpy as np

_PKG = Path(__file__).resolve().parent
_LIB_NAME = "libseqalign_hip.so"

SA_OK = 0
SA_ERR_INVALID_CODE = -2
SA_ERR_NOMEM = -3
SA_ERR_HIP = -4
SA_ERR_ARG = -5
SA_ERR_UNSUPPORTED = -6
SA_FLAG_KEEP_DP = 1
SA_FLAG_NO_TRACEBACK = 2
MODES = {"dna": 0, "protein": 1}


class SeqAlignError(RuntimeError):
    def __init__(self, code, msg, bad=None):
        super().__init__(msg)
        self.code = code
        self.bad = bad


class _Scoring(C.Structure):
    _fields_ = [
        ("gap_open", C.c_double),
        ("gap_extend", C.c_double),
        ("sub", C.c_void_p),
        ("alpha", C.c_int32),
        ("idx256", C.c_void_p),
    ]


class _Result(C.Structure):
    _fields_ = [
        ("score", C.c_int64),
        ("end_i", C.c_int64),
        ("end_j", C.c_int64),
        ("start_i", C.c_int64),
        ("start_j", C.c_int64),
        ("ops", C.POINTER(C.c_uint8)),
        ("n_ops", C.c_int64),
        ("dp", C.POINTER(C.c_int32)),
        ("bad_x", C.c_ubyte),
        ("bad_y", C.c_ubyte),
        ("fill_ms", C.c_float),
        ("traceback_ms", C.c_float),
        ("kernel_fill_ms", C.c_float),
        ("n_launches", C.c_int32),
        ("rows_per_lane", C.c_int32),
        ("tile_steps", C.c_int32),
        ("cells_trace", C.c_int64),
        ("cells_notrace", C.c_int64),
        ("border_wait_ms", C.c_float),
    ]


STRIPE_EVENT_FN = C.CFUNCTYPE(C.c_int, C.c_void_p, C.c_int, C.c_int64)
SA_EV_NEED_LEFT = 0
SA_EV_RIGHT_READY = 1
KINDS = {"nw": 0, "sw": 1, "lcs": 2}


class _StripeDesc(C.Structure):
    _fields_ = [
        ("n_total", C.c_int64),
        ("col_begin", C.c_int64),
        ("is_first", C.c_int32),
        ("is_last", C.c_int32),
        ("left", C.c_void_p),
        ("right", C.c_void_p),
        ("chunk_rows", C.c_int64),
        ("on_event", STRIPE_EVENT_FN),
        ("user", C.c_void_p),
        ("stream_in", C.c_void_p),
        ("stream_out", C.c_void_p),
    ]


@dataclass
class AlignResult:
    score: int
    end: tuple
    start: tuple
    ops: np.ndarray
    dp: Optional[np.ndarray] = None
    fill_ms: float = 0.0
    traceback_ms: float = 0.0
    kernel_fill_ms: float = 0.0
    n_launches: int = 0
    rows_per_lane: int = 0
    tile_steps: int = 0
    cells_trace: int = 0
    cells_notrace: int = 0
    border_wait_ms: float = 0.0
    extra: dict = field(default_factory=dict)


def lib_path() -> Path:
    # SA_LIBRARY: another build of the same library (A/B measurements of kernel variants)
    import os

    return Path(os.environ["SA_LIBRARY"]) if os.environ.get("SA_LIBRARY") else _PKG / _LIB_NAME


_lib = None


def load_library() -> C.CDLL:
    """Load libseqalign_hip.so from the package directory.  No fallback: a missing
    library is an error (build it with `python -c 'import __graft_entry__ as g; g.build()'`
    or `make -C sequencealigner_amd/csrc`)."""
    global _lib
    if _lib is not None:
        return _lib
    p = lib_path()
    if not p.exists():
        raise SeqAlignError(SA_ERR_HIP, f"{p} is missing: the HIP extension was not built; there is no CPU fallback")
    L = C.CDLL(str(p))
    vp = C.c_void_p
    L.sa_create.argtypes = [C.c_int]
    L.sa_create.restype = vp
    L.sa_destroy.argtypes = [vp]
    L.sa_destroy.restype = None
    L.sa_last_error.argtypes = [vp]
    L.sa_last_error.restype = C.c_char_p
    L.sa_set_stream.argtypes = [vp, vp]
    L.sa_set_tuning.argtypes = [vp, C.c_int, C.c_int]
    L.sa_release_workspace.argtypes = [vp]
    L.sa_set_option.argtypes = [vp, C.c_char_p, C.c_int64]
    L.sa_set_option.restype = C.c_int
    L.sa_free_result.argtypes = [C.POINTER(_Result)]
    L.sa_free_result.restype = None
    L.sa_builtin_scoring.argtypes = [C.c_int]
    L.sa_builtin_scoring.restype = C.POINTER(_Scoring)
    pair = [vp, vp, C.c_int64, vp, C.c_int64, C.POINTER(_Scoring), C.c_uint32, C.POINTER(_Result)]
    L.sa_nw.argtypes = pair
    L.sa_sw.argtypes = pair
    L.sa_lcs.argtypes = [vp, vp, C.c_int64, vp, C.c_int64, C.c_uint32, C.POINTER(_Result)]
    L.sa_sw_batch.argtypes = [vp, vp, vp, vp, vp, vp, C.c_int64, C.POINTER(_Scoring), vp, vp, vp, C.POINTER(C.c_float)]
    L.sa_sw_batch_noalloc.argtypes = L.sa_sw_batch.argtypes
    L.sa_sw_batch_noalloc.restype = C.c_int
    L.sa_stripe_fill.argtypes = [vp, C.c_int, vp, C.c_int64, vp, C.c_int64, C.POINTER(_Scoring), C.POINTER(_StripeDesc),
                                 C.POINTER(_Result)]
    L.sa_stripe_traceback.argtypes = [vp, C.c_int64, C.c_int64, C.POINTER(_Result)]
    L.sa_sw_batch_align.argtypes = [vp, vp, vp, vp, vp, vp, C.c_int64, C.POINTER(_Scoring), vp, vp, vp, vp, vp, vp, vp,
                                    C.POINTER(vp), C.POINTER(C.c_int64), C.POINTER(C.c_float)]
    L.sa_sw_batch_align.restype = C.c_int
    L.sa_nw_batch_align.argtypes = [vp, vp, vp, vp, vp, vp, C.c_int64, C.POINTER(_Scoring), vp, vp, vp, C.POINTER(vp),
                                    C.POINTER(C.c_int64), C.POINTER(C.c_float)]
    L.sa_nw_batch_align.restype = C.c_int
    L.sa_lcs_batch_align.argtypes = [vp, vp, vp, vp, vp, vp, C.c_int64, vp, vp, vp, C.POINTER(vp), C.POINTER(C.c_int64),
                                     C.POINTER(C.c_float)]
    L.sa_lcs_batch_align.restype = C.c_int
    L.sa_sw_window_batch.argtypes = [vp, vp, vp, vp, vp, vp, C.c_int64, C.POINTER(_Scoring), vp, vp, vp, vp, vp, vp,
                                     C.POINTER(C.c_float)]
    L.sa_sw_window_batch.restype = C.c_int
    L.sa_batch_set_device_pool.argtypes = [vp, vp, C.c_int64]
    L.sa_batch_set_device_pool.restype = C.c_int
    L.sa_free_pool.argtypes = [vp]
    L.sa_free_pool.restype = None
    for f in ("sa_set_stream", "sa_set_tuning", "sa_release_workspace", "sa_nw", "sa_sw", "sa_lcs", "sa_sw_batch",
              "sa_stripe_fill", "sa_stripe_traceback"):
        getattr(L, f).restype = C.c_int
    _lib = L
    return L


def _u8(b) -> np.ndarray:
    if b is None:  # batch calls with a device-resident pool
        return np.zeros(1, dtype=np.uint8)
    if isinstance(b, str):
        b = b.encode("latin-1")
    if isinstance(b, (bytes, bytearray)):
        return np.frombuffer(bytes(b), dtype=np.uint8)
    return np.ascontiguousarray(b, dtype=np.uint8)


class Engine:
    """One sa_ctx on one GPU."""

    def __init__(self, device: int = 0):
        self.lib = load_library()
        self.ctx = self.lib.sa_create(device)
        if not self.ctx:
            raise SeqAlignError(SA_ERR_HIP, f"sa_create({device}) failed: no gfx950 GPU visible; there is no CPU fallback")

    def close(self):
        if getattr(self, "ctx", None):
            self.lib.sa_destroy(self.ctx)
            self.ctx = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    # ---- helpers -----------------------------------------------------------
    def _scoring(self, mode, gap_open, gap_extend, sub=None, idx256=None):
        base = self.lib.sa_builtin_scoring(MODES[mode]).contents
        sc = _Scoring(float(gap_open), float(gap_extend), base.sub, base.alpha, base.idx256)
        keep = []
        if sub is not None:
            sub = np.ascontiguousarray(sub, np.int8)
            idx256 = np.ascontiguousarray(idx256, np.uint8)
            sc.sub, sc.alpha, sc.idx256 = sub.ctypes.data, sub.shape[0], idx256.ctypes.data
            keep = [sub, idx256]
        return sc, keep

    def _err(self, rc, r):
        msg = self.lib.sa_last_error(self.ctx).decode()
        bad = (chr(r.bad_x), chr(r.bad_y)) if rc == SA_ERR_INVALID_CODE else None
        return SeqAlignError(rc, msg, bad)

    def _wrap(self, r, m, n, keep_dp) -> AlignResult:
        ops = np.ctypeslib.as_array(r.ops, shape=(r.n_ops,)).copy() if r.n_ops else np.zeros(0, np.uint8)
        dp = None
        if keep_dp and r.dp:
            dp = np.ctypeslib.as_array(r.dp, shape=(m + 1, n + 1)).copy()
        out = AlignResult(int(r.score), (int(r.end_i), int(r.end_j)), (int(r.start_i), int(r.start_j)), ops, dp,
                          r.fill_ms, r.traceback_ms, r.kernel_fill_ms, r.n_launches, r.rows_per_lane, r.tile_steps,
                          int(r.cells_trace), int(r.cells_notrace), float(r.border_wait_ms))
        self.lib.sa_free_result(C.byref(r))
        return out

    def set_tuning(self, rows_per_lane=0, tile_steps=0):
        rc = self.lib.sa_set_tuning(self.ctx, rows_per_lane, tile_steps)
        if rc:
            raise SeqAlignError(rc, self.lib.sa_last_error(self.ctx).decode())

    def set_option(self, name, value):
        rc = self.lib.sa_set_option(self.ctx, name.encode(), int(value))
        if rc:
            raise SeqAlignError(rc, self.lib.sa_last_error(self.ctx).decode())

    def set_stream(self, stream_handle):
        self.lib.sa_set_stream(self.ctx, stream_handle)

    def release_workspace(self):
        self.lib.sa_release_workspace(self.ctx)

    # ---- hot path -------------------------------------------------------------
    def _pair(self, fn, x, y, mode, gap_open, gap_extend, flags, sub=None, idx256=None):
        x, y = _u8(x), _u8(y)
        sc, keep = self._scoring(mode, gap_open, gap_extend, sub, idx256)
        r = _Result()
        rc = fn(self.ctx, x.ctypes.data, len(x), y.ctypes.data, len(y), C.byref(sc), flags, C.byref(r))
        del keep
        if rc:
            raise self._err(rc, r)
        return self._wrap(r, len(x), len(y), bool(flags & SA_FLAG_KEEP_DP))

    def nw(self, x, y, mode="dna", gap_open=-5.0, gap_extend=-1.0, flags=0, **kw) -> AlignResult:
        return self._pair(self.lib.sa_nw, x, y, mode, gap_open, gap_extend, flags, **kw)

    def sw(self, x, y, mode="dna", gap_open=-5.0, gap_extend=-1.0, flags=0, **kw) -> AlignResult:
        return self._pair(self.lib.sa_sw, x, y, mode, gap_open, gap_extend, flags, **kw)

    def lcs(self, x, y, flags=0) -> AlignResult:
        x, y = _u8(x), _u8(y)
        r = _Result()
        rc = self.lib.sa_lcs(self.ctx, x.ctypes.data, len(x), y.ctypes.data, len(y), flags, C.byref(r))
        if rc:
            raise self._err(rc, r)
        return self._wrap(r, len(x), len(y), bool(flags & SA_FLAG_KEEP_DP))

    # ---- column stripes (one process per GPU; the caller moves the border chunks) ----------
    def stripe_fill(self, kind, x, y_stripe, n_total, col_begin, is_first, is_last, left_ptr, right_ptr, chunk_rows,
                    on_event, mode="dna", gap_open=-5.0, gap_extend=-1.0, stream_in=None, stream_out=None) -> AlignResult:
        """on_event(event, chunk) -> None is called while the fill is being enqueued (see
        include/seqalign_hip.h); left_ptr / right_ptr are device addresses of int32[m+1][2]."""
        x, y = _u8(x), _u8(y_stripe)
        sc, keep = self._scoring(mode, gap_open, gap_extend)
        err = []

        def _cb(_user, event, chunk):
            try:
                on_event(event, chunk)
                return 0
            except BaseException as e:  # never let an exception cross the C frame
                err.append(e)
                return 1

        cb = STRIPE_EVENT_FN(_cb)
        d = _StripeDesc(int(n_total), int(col_begin), int(is_first), int(is_last), left_ptr, right_ptr, int(chunk_rows), cb, None,
                        stream_in, stream_out)
        r = _Result()
        rc = self.lib.sa_stripe_fill(self.ctx, KINDS[kind], x.ctypes.data, len(x), y.ctypes.data, len(y), C.byref(sc),
                                     C.byref(d), C.byref(r))
        del keep
        if err:
            raise err[0]
        if rc:
            raise self._err(rc, r)
        return self._wrap(r, len(x), len(y), False)

    def stripe_traceback(self, start_i, start_j) -> AlignResult:
        r = _Result()
        rc = self.lib.sa_stripe_traceback(self.ctx, int(start_i), int(start_j), C.byref(r))
        if rc:
            raise self._err(rc, r)
        return self._wrap(r, 0, 0, False)

    def set_device_pool(self, data_ptr, nbytes):
        """The batch calls read their sequences from this device buffer (e.g. tensor.data_ptr(), tensor.numel())
        instead of copying the host pool in; set_device_pool(None, 0) switches back."""
        rc = self.lib.sa_batch_set_device_pool(self.ctx, data_ptr, int(nbytes))
        if rc:
            raise SeqAlignError(rc, self.lib.sa_last_error(self.ctx).decode())

    def sw_batch_noalloc(self, *a, **kw):
        """sa_sw_batch_noalloc: sw_batch through the entry point that allocates nothing (static device scratch)."""
        return self.sw_batch(*a, _fn="sa_sw_batch_noalloc", **kw)

    def sw_batch(self, seqs, off_x, len_x, off_y, len_y, mode="protein", gap_open=-5.0, gap_extend=-1.0, sub=None, idx256=None,
                 _fn="sa_sw_batch"):
        seqs = _u8(seqs)
        off_x = np.ascontiguousarray(off_x, np.int64)
        off_y = np.ascontiguousarray(off_y, np.int64)
        len_x = np.ascontiguousarray(len_x, np.int32)
        len_y = np.ascontiguousarray(len_y, np.int32)
        n = len(off_x)
        score = np.zeros(n, np.int32)
        ei = np.zeros(n, np.int32)
        ej = np.zeros(n, np.int32)
        sc, keep = self._scoring(mode, gap_open, gap_extend, sub, idx256)
        ms = C.c_float(0)
        rc = getattr(self.lib, _fn)(self.ctx, seqs.ctypes.data, off_x.ctypes.data, len_x.ctypes.data, off_y.ctypes.data,
                                    len_y.ctypes.data, n, C.byref(sc), score.ctypes.data, ei.ctypes.data, ej.ctypes.data,
                                    C.byref(ms))
        if rc and rc != SA_ERR_INVALID_CODE:
            raise SeqAlignError(rc, self.lib.sa_last_error(self.ctx).decode())
        return score, ei, ej, ms.value, rc

    def sw_batch_align(self, seqs, off_x, len_x, off_y, len_y, mode="protein", gap_open=-5.0, gap_extend=-1.0):
        """Full local alignments of many pairs (sa_sw_batch_align).  Returns a BatchAlignment:
        per-pair score / end / start arrays and `ops(k)` = the op list of pair k, end -> start."""
        seqs = _u8(seqs)
        off_x = np.ascontiguousarray(off_x, np.int64)
        off_y = np.ascontiguousarray(off_y, np.int64)
        len_x = np.ascontiguousarray(len_x, np.int32)
        len_y = np.ascontiguousarray(len_y, np.int32)
        n = len(off_x)
        i32 = lambda: np.zeros(n, np.int32)  # noqa: E731
        score, ei, ej, si, sj, nops = i32(), i32(), i32(), i32(), i32(), i32()
        ops_off = np.zeros(n, np.int64)
        sc, keep = self._scoring(mode, gap_open, gap_extend)
        ms = C.c_float(0)
        pool = C.c_void_p()
        pool_bytes = C.c_int64(0)
        rc = self.lib.sa_sw_batch_align(self.ctx, seqs.ctypes.data, off_x.ctypes.data, len_x.ctypes.data, off_y.ctypes.data,
                                        len_y.ctypes.data, n, C.byref(sc), score.ctypes.data, ei.ctypes.data, ej.ctypes.data,
                                        si.ctypes.data, sj.ctypes.data, ops_off.ctypes.data, nops.ctypes.data,
                                        C.byref(pool), C.byref(pool_bytes), C.byref(ms))
        del keep
        if rc and rc != SA_ERR_INVALID_CODE:
            raise SeqAlignError(rc, self.lib.sa_last_error(self.ctx).decode())
        nb = int(pool_bytes.value)
        out = BatchAlignment(score, ei, ej, si, sj, ops_off, nops, np.zeros(0, np.uint8), ms.value, rc)
        if nb:  # a view of the library's pool (no copy); freed when the BatchAlignment goes away
            out.ops_pool = np.ctypeslib.as_array(C.cast(pool, C.POINTER(C.c_uint8)), shape=(nb,))
            out._release = (self.lib.sa_free_pool, pool)
        else:
            self.lib.sa_free_pool(pool)
        return out

    def sw_window_batch(self, seqs, off_x, len_x, off_y, len_y, mode="dna", gap_open=-5.0, gap_extend=-1.0, sub=None, idx256=None):
        """The seed_aligner's windowed SW for many (query, window) pairs (sa_sw_window_batch): returns
        score, end_i, end_j, start_i, start_j, aligned_len (arrays), kernel ms, rc."""
        seqs = _u8(seqs)
        off_x = np.ascontiguousarray(off_x, np.int64)
        off_y = np.ascontiguousarray(off_y, np.int64)
        len_x = np.ascontiguousarray(len_x, np.int32)
        len_y = np.ascontiguousarray(len_y, np.int32)
        n = len(off_x)
        i32 = lambda: np.zeros(n, np.int32)  # noqa: E731
        score, ei, ej, si, sj, ln = i32(), i32(), i32(), i32(), i32(), i32()
        sc, keep = self._scoring(mode, gap_open, gap_extend, sub, idx256)
        ms = C.c_float(0)
        rc = self.lib.sa_sw_window_batch(self.ctx, seqs.ctypes.data, off_x.ctypes.data, len_x.ctypes.data, off_y.ctypes.data,
                                         len_y.ctypes.data, n, C.byref(sc), score.ctypes.data, ei.ctypes.data, ej.ctypes.data,
                                         si.ctypes.data, sj.ctypes.data, ln.ctypes.data, C.byref(ms))
        del keep
        if rc and rc != SA_ERR_INVALID_CODE:
            raise SeqAlignError(rc, self.lib.sa_last_error(self.ctx).decode())
        return score, ei, ej, si, sj, ln, ms.value, rc

    def nw_batch_align(self, seqs, off_x, len_x, off_y, len_y, mode="protein", gap_open=-5.0, gap_extend=-1.0, sub=None, idx256=None):
        """Global alignments of many pairs (sa_nw_batch_align): BatchAlignment with end = (m, n), start = (0, 0)."""
        seqs = _u8(seqs)
        off_x = np.ascontiguousarray(off_x, np.int64)
        off_y = np.ascontiguousarray(off_y, np.int64)
        len_x = np.ascontiguousarray(len_x, np.int32)
        len_y = np.ascontiguousarray(len_y, np.int32)
        n = len(off_x)
        score, nops = np.zeros(n, np.int32), np.zeros(n, np.int32)
        ops_off = np.zeros(n, np.int64)
        sc, keep = self._scoring(mode, gap_open, gap_extend, sub, idx256)
        ms = C.c_float(0)
        pool = C.c_void_p()
        pool_bytes = C.c_int64(0)
        rc = self.lib.sa_nw_batch_align(self.ctx, seqs.ctypes.data, off_x.ctypes.data, len_x.ctypes.data, off_y.ctypes.data,
                                        len_y.ctypes.data, n, C.byref(sc), score.ctypes.data, ops_off.ctypes.data,
                                        nops.ctypes.data, C.byref(pool), C.byref(pool_bytes), C.byref(ms))
        del keep
        if rc and rc != SA_ERR_INVALID_CODE:
            raise SeqAlignError(rc, self.lib.sa_last_error(self.ctx).decode())
        z = np.zeros(n, np.int32)
        out = BatchAlignment(score, len_x.copy(), len_y.copy(), z, z.copy(), ops_off, nops, np.zeros(0, np.uint8), ms.value, rc)
        nb = int(pool_bytes.value)
        if nb:
            out.ops_pool = np.ctypeslib.as_array(C.cast(pool, C.POINTER(C.c_uint8)), shape=(nb,))
            out._release = (self.lib.sa_free_pool, pool)
        else:
            self.lib.sa_free_pool(pool)
        return out


    def lcs_batch_align(self, seqs, off_x, len_x, off_y, len_y):
        """LCS of many pairs with their tracebacks (sa_lcs_batch_align)."""
        seqs = _u8(seqs)
        off_x = np.ascontiguousarray(off_x, np.int64)
        off_y = np.ascontiguousarray(off_y, np.int64)
        len_x = np.ascontiguousarray(len_x, np.int32)
        len_y = np.ascontiguousarray(len_y, np.int32)
        n = len(off_x)
        score, nops = np.zeros(n, np.int32), np.zeros(n, np.int32)
        ops_off = np.zeros(n, np.int64)
        ms = C.c_float(0)
        pool = C.c_void_p()
        pool_bytes = C.c_int64(0)
        rc = self.lib.sa_lcs_batch_align(self.ctx, seqs.ctypes.data, off_x.ctypes.data, len_x.ctypes.data, off_y.ctypes.data,
                                         len_y.ctypes.data, n, score.ctypes.data, ops_off.ctypes.data, nops.ctypes.data,
                                         C.byref(pool), C.byref(pool_bytes), C.byref(ms))
        if rc:
            raise SeqAlignError(rc, self.lib.sa_last_error(self.ctx).decode())
        z = np.zeros(n, np.int32)
        out = BatchAlignment(score, len_x.copy(), len_y.copy(), z, z.copy(), ops_off, nops, np.zeros(0, np.uint8), ms.value, rc)
        nb = int(pool_bytes.value)
        if nb:
            out.ops_pool = np.ctypeslib.as_array(C.cast(pool, C.POINTER(C.c_uint8)), shape=(nb,))
            out._release = (self.lib.sa_free_pool, pool)
        else:
            self.lib.sa_free_pool(pool)
        return out


@dataclass
class BatchAlignment:
    score: np.ndarray
    end_i: np.ndarray
    end_j: np.ndarray
    start_i: np.ndarray
    start_j: np.ndarray
    ops_off: np.ndarray
    n_ops: np.ndarray
    ops_pool: np.ndarray
    kernel_ms: float
    rc: int
    _release: Optional[tuple] = None  # ops_pool is valid as long as this object lives

    def ops(self, k) -> np.ndarray:
        o = int(self.ops_off[k])
        return self.ops_pool[o:o + int(self.n_ops[k])].copy()

    def __del__(self):
        if self._release is not None:
            free, pool = self._release
            self._release = None
            self.ops_pool = np.zeros(0, np.uint8)
            free(pool)


_default: Optional[Engine] = None


def _engine() -> Engine:
    global _default
    if _default is None:
        _default = Engine(0)
    return _default


def globalalign(x, y, mode="dna", gap_open=-5.0, gap_extend=-1.0) -> AlignResult:
    return _engine().nw(x, y, mode, gap_open, gap_extend)


def localalign(x, y, mode="dna", gap_open=-5.0, gap_extend=-1.0) -> AlignResult:
    return _engine().sw(x, y, mode, gap_open, gap_extend)


def lcs(x, y) -> AlignResult:
    return _engine().lcs(x, y)
