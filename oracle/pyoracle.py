"""ctypes binding of oracle/cpu_ref.c plus helpers that turn an op list into the
reference's output artefacts.  TEST INFRASTRUCTURE ONLY (see oracle/__init__.py).
"""
from __future__ import annotations

import ctypes as C
import os
import subprocess
import tempfile
from dataclasses import dataclass
from pathlib import Path
from typing import Optional

import numpy as np

_HERE = Path(__file__).resolve().parent
_LIB = _HERE / "_build" / "libcpu_ref.so"
_REF = _HERE / "_ref" / "aligner_ref"

MODE = {"dna": 0, "protein": 1, "custom": 2}


class _Res(C.Structure):
    _fields_ = [
        ("score", C.c_int64),
        ("end_i", C.c_int64),
        ("end_j", C.c_int64),
        ("start_i", C.c_int64),
        ("start_j", C.c_int64),
        ("ops", C.POINTER(C.c_uint8)),
        ("n_ops", C.c_int64),
        ("dp", C.POINTER(C.c_int32)),
        ("ptr", C.POINTER(C.c_char)),
        ("bad_x", C.c_ubyte),
        ("bad_y", C.c_ubyte),
    ]


@dataclass
class OracleResult:
    rc: int
    score: int = 0
    end: tuple = (0, 0)
    start: tuple = (0, 0)
    ops: Optional[np.ndarray] = None
    dp: Optional[np.ndarray] = None
    ptr: Optional[np.ndarray] = None
    bad: tuple = (0, 0)


def build(force: bool = False) -> Path:
    """Compile the C restatement (and, when /root/reference is present, the
    unmodified reference binary).  Building the checker is not using it."""
    src_newer = (not _LIB.exists()) or any(
        (_HERE / f).stat().st_mtime > _LIB.stat().st_mtime for f in ("cpu_ref.c", "cpu_ref.h")
    )
    if force or src_newer:
        subprocess.run(["make", "-C", str(_HERE), "-s", "all"], check=True)
    if Path("/root/reference/src/main.cpp").exists() and not _REF.exists():
        subprocess.run(["make", "-C", str(_HERE), "-s", "ref"], check=False)
    return _LIB


def ref_binary() -> Optional[Path]:
    return _REF if _REF.exists() else None


def _u8(b) -> np.ndarray:
    if isinstance(b, str):
        b = b.encode("latin-1")
    if isinstance(b, (bytes, bytearray)):
        return np.frombuffer(bytes(b), dtype=np.uint8)
    return np.ascontiguousarray(b, dtype=np.uint8)


class Oracle:
    def __init__(self):
        build()
        self.lib = C.CDLL(str(_LIB))
        L = self.lib
        p8 = C.c_void_p
        L.orc_nw.argtypes = [p8, C.c_int64, p8, C.c_int64, C.c_int, C.c_double, C.c_double, C.c_int, C.POINTER(_Res)]
        L.orc_sw.argtypes = L.orc_nw.argtypes
        L.orc_lcs.argtypes = [p8, C.c_int64, p8, C.c_int64, C.c_int, C.POINTER(_Res)]
        L.orc_nw_packed.argtypes = [p8, C.c_int64, p8, C.c_int64, C.c_int, C.c_double, C.c_double, C.POINTER(_Res)]
        L.orc_lcs_packed.argtypes = [p8, C.c_int64, p8, C.c_int64, C.POINTER(_Res)]
        L.orc_nw_banded.argtypes = [p8, C.c_int64, p8, C.c_int64, C.c_int, C.c_double, C.c_double, C.c_int64,
                                    C.POINTER(_Res), C.POINTER(C.c_int)]
        L.orc_sw_banded.argtypes = L.orc_nw_banded.argtypes
        L.orc_sw_banded.restype = C.c_int
        L.orc_lcs_banded.argtypes = [p8, C.c_int64, p8, C.c_int64, C.c_int64, C.POINTER(_Res), C.POINTER(C.c_int)]
        L.orc_lcs_banded.restype = C.c_int
        L.orc_sw_score.argtypes = [p8, C.c_int64, p8, C.c_int64, C.c_int, C.c_double, C.c_double,
                                   C.POINTER(C.c_int32), C.POINTER(C.c_int32), C.POINTER(C.c_int32)]
        L.orc_sw_score_batch.argtypes = [p8, p8, p8, p8, p8, C.c_int64, C.c_int, C.c_double, C.c_double, C.c_int,
                                         p8, p8, p8]
        L.orc_sigma.argtypes = [C.c_int, C.c_ubyte, C.c_ubyte, C.POINTER(C.c_int)]
        L.orc_free.argtypes = [C.POINTER(_Res)]
        for f in ("orc_nw", "orc_sw", "orc_lcs", "orc_nw_packed", "orc_lcs_packed", "orc_nw_banded",
                  "orc_sw_score", "orc_sw_score_batch", "orc_sigma"):
            getattr(L, f).restype = C.c_int

    # ---- helpers -------------------------------------------------------
    def _wrap(self, rc, r, m, n, keep_dp, has_ptr=False) -> OracleResult:
        out = OracleResult(rc=rc, bad=(r.bad_x, r.bad_y))
        if rc == 0:
            out.score = int(r.score)
            out.end = (int(r.end_i), int(r.end_j))
            out.start = (int(r.start_i), int(r.start_j))
            out.ops = (np.ctypeslib.as_array(r.ops, shape=(r.n_ops,)).copy() if r.n_ops else np.zeros(0, np.uint8))
            if keep_dp and r.dp:
                out.dp = np.ctypeslib.as_array(r.dp, shape=(m + 1, n + 1)).copy()
            if has_ptr and r.ptr:
                out.ptr = np.frombuffer(C.string_at(r.ptr, (m + 1) * (n + 1)), dtype="S1").reshape(m + 1, n + 1).copy()
        self.lib.orc_free(C.byref(r))
        return out

    def set_custom_table(self, scores: dict):
        """scores: {"xy": value} for every valid residue pair; selects mode "custom" for later calls"""
        t = np.full((256, 256), np.iinfo(np.int16).min, np.int16)
        for pair, v in scores.items():
            t[ord(pair[0]), ord(pair[1])] = v
        self.lib.orc_set_custom_table.argtypes = [C.c_void_p]
        self.lib.orc_set_custom_table(t.ctypes.data)

    def sigma(self, mode, x, y):
        ok = C.c_int(0)
        v = self.lib.orc_sigma(MODE[mode], ord(x), ord(y), C.byref(ok))
        return v if ok.value else None

    def nw(self, x, y, mode="dna", go=-5.0, ge=-1.0, keep_dp=False, packed=False) -> OracleResult:
        x, y = _u8(x), _u8(y)
        r = _Res()
        if packed:
            rc = self.lib.orc_nw_packed(x.ctypes.data, len(x), y.ctypes.data, len(y), MODE[mode], go, ge, C.byref(r))
            return self._wrap(rc, r, len(x), len(y), False)
        rc = self.lib.orc_nw(x.ctypes.data, len(x), y.ctypes.data, len(y), MODE[mode], go, ge, int(keep_dp), C.byref(r))
        return self._wrap(rc, r, len(x), len(y), keep_dp)

    def sw(self, x, y, mode="dna", go=-5.0, ge=-1.0, keep_dp=False) -> OracleResult:
        x, y = _u8(x), _u8(y)
        r = _Res()
        rc = self.lib.orc_sw(x.ctypes.data, len(x), y.ctypes.data, len(y), MODE[mode], go, ge, int(keep_dp), C.byref(r))
        return self._wrap(rc, r, len(x), len(y), keep_dp)

    def lcs(self, x, y, keep_dp=False, packed=False) -> OracleResult:
        x, y = _u8(x), _u8(y)
        r = _Res()
        if packed:
            rc = self.lib.orc_lcs_packed(x.ctypes.data, len(x), y.ctypes.data, len(y), C.byref(r))
            return self._wrap(rc, r, len(x), len(y), False)
        rc = self.lib.orc_lcs(x.ctypes.data, len(x), y.ctypes.data, len(y), int(keep_dp), C.byref(r))
        return self._wrap(rc, r, len(x), len(y), keep_dp, has_ptr=keep_dp)

    def nw_banded(self, x, y, band, mode="dna", go=-5.0, ge=-1.0):
        x, y = _u8(x), _u8(y)
        r = _Res()
        touched = C.c_int(0)
        rc = self.lib.orc_nw_banded(x.ctypes.data, len(x), y.ctypes.data, len(y), MODE[mode], go, ge, band,
                                    C.byref(r), C.byref(touched))
        return self._wrap(rc, r, len(x), len(y), False), touched.value

    def sw_banded(self, x, y, band, mode="dna", go=-5.0, ge=-1.0):
        x, y = _u8(x), _u8(y)
        r = _Res()
        touched = C.c_int(0)
        rc = self.lib.orc_sw_banded(x.ctypes.data, len(x), y.ctypes.data, len(y), MODE[mode], go, ge, band,
                                    C.byref(r), C.byref(touched))
        return self._wrap(rc, r, len(x), len(y), False), touched.value

    def lcs_banded(self, x, y, band):
        x, y = _u8(x), _u8(y)
        r = _Res()
        touched = C.c_int(0)
        rc = self.lib.orc_lcs_banded(x.ctypes.data, len(x), y.ctypes.data, len(y), band, C.byref(r), C.byref(touched))
        return self._wrap(rc, r, len(x), len(y), False), touched.value

    def sw_score(self, x, y, mode="protein", go=-5.0, ge=-1.0):
        x, y = _u8(x), _u8(y)
        s, i, j = C.c_int32(), C.c_int32(), C.c_int32()
        rc = self.lib.orc_sw_score(x.ctypes.data, len(x), y.ctypes.data, len(y), MODE[mode], go, ge,
                                   C.byref(s), C.byref(i), C.byref(j))
        return rc, s.value, i.value, j.value

    def sw_score_batch(self, seqs, off_x, len_x, off_y, len_y, mode="protein", go=-5.0, ge=-1.0, threads=1):
        seqs = _u8(seqs)
        off_x = np.ascontiguousarray(off_x, np.int64)
        off_y = np.ascontiguousarray(off_y, np.int64)
        len_x = np.ascontiguousarray(len_x, np.int32)
        len_y = np.ascontiguousarray(len_y, np.int32)
        n = len(off_x)
        sc = np.zeros(n, np.int32)
        ei = np.zeros(n, np.int32)
        ej = np.zeros(n, np.int32)
        rc = self.lib.orc_sw_score_batch(seqs.ctypes.data, off_x.ctypes.data, len_x.ctypes.data, off_y.ctypes.data,
                                         len_y.ctypes.data, n, MODE[mode], go, ge, threads,
                                         sc.ctypes.data, ei.ctypes.data, ej.ctypes.data)
        return rc, sc, ei, ej


# ---- op list -> the reference's artefacts ------------------------------------

def ops_to_alignment(x, y, ops, end_i, end_j):
    """Aligned strings (start -> end), as main.cpp:782-795 builds then reverses them."""
    x, y = _u8(x), _u8(y)
    ops = np.asarray(ops, np.uint8)
    n = len(ops)
    ax = np.full(n, ord("-"), np.uint8)
    ay = np.full(n, ord("-"), np.uint8)
    takes_x = ops != 1
    takes_y = ops != 2
    # traceback order consumes x[end_i-1], x[end_i-2], ...
    xi = end_i - np.cumsum(takes_x)
    yj = end_j - np.cumsum(takes_y)
    ax[takes_x] = x[xi[takes_x]]
    ay[takes_y] = y[yj[takes_y]]
    return ax[::-1].tobytes().decode("latin-1"), ay[::-1].tobytes().decode("latin-1")


def ops_to_path(kind, ops, end_i, end_j):
    """Path points (j, i) in the order the reference writes them:
    global: end -> start, never reversed (main.cpp:776,817,833-836);
    local / lcs: start -> end (main.cpp:1262, 1640)."""
    ops = np.asarray(ops, np.uint8)
    di = np.concatenate([[0], np.cumsum(ops != 1)])
    dj = np.concatenate([[0], np.cumsum(ops != 2)])
    pts = np.stack([end_j - dj, end_i - di], axis=1)
    if kind in ("local", "lcs"):
        pts = pts[::-1]
    return pts


def run_reference(query_fa, target_fa, choice, mode="dna", go=None, ge=None, extra=()):
    """Run the compiled, unmodified reference (build container only)."""
    if not _REF.exists():
        raise RuntimeError("oracle/_ref/aligner_ref not built (make -C oracle ref)")
    out = tempfile.mkdtemp(prefix="ref_out_")
    cmd = [str(_REF), "--query", str(query_fa), "--target", str(target_fa), "--choice", str(choice),
           "--mode", mode, "--outdir", out]
    if go is not None:
        cmd += ["--gap_open", repr(go)]
    if ge is not None:
        cmd += ["--gap_extend", repr(ge)]
    cmd += list(extra)
    env = dict(os.environ)
    env["LD_LIBRARY_PATH"] = "/usr/lib/x86_64-linux-gnu:/opt/conda/lib"
    p = subprocess.run(cmd, capture_output=True, env=env, timeout=3600)
    files = {f.name: f.read_bytes() for f in Path(out).iterdir()}
    return p.returncode, p.stdout, p.stderr, files, out
