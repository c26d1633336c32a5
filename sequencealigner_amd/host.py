"""ctypes view of include/seqalign_host.h (libsa_host.so): the reference's FASTA input and
byte-exact result writers.  Host-only code, usable without a GPU."""
from __future__ import annotations

import ctypes as C
from pathlib import Path

import numpy as np

_PKG = Path(__file__).resolve().parent
MODES = {"dna": 0, "protein": 1}


class _Job(C.Structure):
    _fields_ = [
        ("outdir", C.c_char_p), ("header1", C.c_char_p), ("header2", C.c_char_p), ("mode", C.c_int),
        ("gap_open", C.c_double), ("gap_extend", C.c_double),
        ("x", C.c_void_p), ("m", C.c_int64), ("y", C.c_void_p), ("n", C.c_int64),
        ("score", C.c_int64), ("end_i", C.c_int64), ("end_j", C.c_int64),
        ("ops", C.c_void_p), ("n_ops", C.c_int64), ("time_ms", C.c_int64),
        ("verbose", C.c_int), ("binary", C.c_int), ("txt", C.c_int), ("dp", C.c_void_p),
    ]


_lib = None


def load_host_library() -> C.CDLL:
    global _lib
    if _lib is None:
        p = _PKG / "libsa_host.so"
        if not p.exists():
            raise RuntimeError(f"{p} missing: run `make -C sequencealigner_amd/csrc`")
        L = C.CDLL(str(p))
        L.sah_read_fasta.argtypes = [C.c_char_p, C.POINTER(C.c_void_p), C.POINTER(C.c_void_p), C.POINTER(C.c_int64)]
        L.sah_read_fasta.restype = C.c_int
        L.sah_free.argtypes = [C.c_void_p]
        L.sah_free.restype = None
        for f in ("sah_accession", "sah_gene_symbol"):
            getattr(L, f).argtypes = [C.c_char_p, C.c_int, C.c_char_p, C.c_int64]
            getattr(L, f).restype = None
        for f in ("sah_write_global", "sah_write_local", "sah_write_lcs"):
            getattr(L, f).argtypes = [C.POINTER(_Job)]
            getattr(L, f).restype = C.c_int
        _lib = L
    return _lib


def read_fasta(path):
    L = load_host_library()
    h, s, n = C.c_void_p(), C.c_void_p(), C.c_int64()
    if L.sah_read_fasta(str(path).encode(), C.byref(h), C.byref(s), C.byref(n)) != 0:
        raise FileNotFoundError(f"Error: Unable to open {path}")
    header = C.string_at(h).decode("latin-1")
    seq = C.string_at(s, n.value)
    L.sah_free(h)
    L.sah_free(s)
    return header, seq


def read_fasta_records(path):
    """every record of a multi-FASTA file: [(header, residues bytes)] (sah_read_fasta_records)"""
    L = load_host_library()
    L.sah_read_fasta_records.argtypes = [C.c_char_p, C.POINTER(C.c_int64), C.POINTER(C.POINTER(C.c_char_p)),
                                         C.POINTER(C.POINTER(C.c_void_p)), C.POINTER(C.POINTER(C.c_int64))]
    L.sah_read_fasta_records.restype = C.c_int
    L.sah_free_records.argtypes = [C.c_int64, C.POINTER(C.c_char_p), C.POINTER(C.c_void_p), C.POINTER(C.c_int64)]
    L.sah_free_records.restype = None
    n = C.c_int64()
    hs, ss, ls = C.POINTER(C.c_char_p)(), C.POINTER(C.c_void_p)(), C.POINTER(C.c_int64)()
    if L.sah_read_fasta_records(str(path).encode(), C.byref(n), C.byref(hs), C.byref(ss), C.byref(ls)) != 0:
        raise FileNotFoundError(f"Error: Unable to open {path}")
    out = [(hs[k].decode("latin-1"), C.string_at(ss[k], ls[k])) for k in range(n.value)]
    L.sah_free_records(n, hs, ss, ls)
    return out


def accession(header: str, mode: str) -> str:
    buf = C.create_string_buffer(4096)
    load_host_library().sah_accession(header.encode("latin-1"), MODES[mode], buf, 4096)
    return buf.value.decode("latin-1")


def gene_symbol(header: str, mode: str) -> str:
    buf = C.create_string_buffer(4096)
    load_host_library().sah_gene_symbol(header.encode("latin-1"), MODES[mode], buf, 4096)
    return buf.value.decode("latin-1")


def write_result(kind, outdir, header1, header2, mode, x, y, score, end, ops, gap_open=-5.0, gap_extend=-1.0,
                 time_ms=0, dp=None, binary=False, txt=False, verbose=False):
    """kind in {"global","local","lcs"}; x, y raw residue bytes; ops/end as returned by the engine."""
    L = load_host_library()
    xb = np.frombuffer(x if isinstance(x, (bytes, bytearray)) else str(x).encode("latin-1"), np.uint8) \
        if not isinstance(x, np.ndarray) else np.ascontiguousarray(x, np.uint8)
    yb = np.frombuffer(y if isinstance(y, (bytes, bytearray)) else str(y).encode("latin-1"), np.uint8) \
        if not isinstance(y, np.ndarray) else np.ascontiguousarray(y, np.uint8)
    ops = np.ascontiguousarray(ops, np.uint8)
    dpa = np.ascontiguousarray(dp, np.int32) if dp is not None else None
    job = _Job(str(outdir).encode(), header1.encode("latin-1"), header2.encode("latin-1"), MODES[mode],
               float(gap_open), float(gap_extend),
               xb.ctypes.data if len(xb) else None, len(xb), yb.ctypes.data if len(yb) else None, len(yb),
               int(score), int(end[0]), int(end[1]), ops.ctypes.data if len(ops) else None, len(ops), int(time_ms),
               int(verbose), int(binary), int(txt), dpa.ctypes.data if dpa is not None else None)
    fn = {"global": L.sah_write_global, "local": L.sah_write_local, "lcs": L.sah_write_lcs}[kind]
    return fn(C.byref(job))
