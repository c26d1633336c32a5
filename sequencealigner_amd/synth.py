"""Synthetic workloads of BASELINE.json's configs, defined once.

DNA pair: x uniform over ACGT; y = x with per-base 6 % substitution, 2 % deletion and 2 %
insertion (uniform base), i.e. about 90 % identity; `related=False` gives an independent
uniform y instead.  numpy PCG64 streams, so a (seed, n) pair names one input everywhere.
"""
from __future__ import annotations

import numpy as np

_ACGT = np.frombuffer(b"ACGT", dtype=np.uint8)
_AA20 = np.frombuffer(b"ARNDCQEGHILKMFPSTWYV", dtype=np.uint8)


def dna_pair(n: int, seed: int = 1234, related: bool = True, sub=0.06, dele=0.02, ins=0.02):
    rng = np.random.default_rng(seed)
    x = _ACGT[rng.integers(0, 4, size=n)]
    if not related:
        rng2 = np.random.default_rng(seed + 4444)
        return x.copy(), _ACGT[rng2.integers(0, 4, size=n)]
    r = rng.random(n)
    keep = r >= dele
    is_sub = keep & (r < dele + sub)
    base = x.copy()
    base[is_sub] = _ACGT[rng.integers(0, 4, size=int(is_sub.sum()))]
    has_ins = rng.random(n) < ins
    insb = _ACGT[rng.integers(0, 4, size=n)]
    two = np.stack([base, insb], axis=1).reshape(-1)
    mask = np.stack([keep, has_ins], axis=1).reshape(-1)
    return x.copy(), two[mask].copy()


def protein_pairs(n_pairs: int, seed: int = 42, lo: int = 300, hi: int = 1000):
    """n_pairs independent protein pairs, lengths iid uniform[lo, hi], residues uniform over the
    20 standard amino acids.  Returns (pool, off_x, len_x, off_y, len_y)."""
    rng = np.random.default_rng(seed)
    len_x = rng.integers(lo, hi + 1, size=n_pairs).astype(np.int32)
    len_y = rng.integers(lo, hi + 1, size=n_pairs).astype(np.int32)
    lens = np.stack([len_x, len_y], axis=1).reshape(-1).astype(np.int64)
    offs = np.concatenate([[0], np.cumsum(lens)[:-1]])
    pool = _AA20[rng.integers(0, 20, size=int(lens.sum()))]
    return pool, offs[0::2].copy(), len_x, offs[1::2].copy(), len_y
