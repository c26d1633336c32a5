"""Shared by the seed_aligner segment tests: the fixture produced from the reference's own
functions (tests/golden/make_golden_seed.py) and the glue that turns a traceback op list into the
gapped strings those functions return."""
import json
from pathlib import Path

import numpy as np

GOLD = Path(__file__).resolve().parent / "golden" / "seed_segments.json"


def load():
    return json.loads(GOLD.read_text())


def scoring_arrays(tab):
    """(sub int8 alpha x alpha, idx256 uint8) for sa_scoring from the fixture's score dictionary"""
    valid = tab["valid"]
    # residues that score identically against everything share an index (U/T, X/N in the DNA table)
    rows = {}
    idx = np.full(256, 255, np.uint8)
    reps = []
    for ch in valid:
        key = tuple(tab["score"][ch + o] for o in valid)
        if key not in rows:
            rows[key] = len(reps)
            reps.append(ch)
        idx[ord(ch)] = rows[key]
    a = len(reps)
    sub = np.zeros((a, a), np.int8)
    for i, x in enumerate(reps):
        for j, y in enumerate(reps):
            sub[i, j] = tab["score"][x + y]
    return sub, idx


def replay(x, y, ops, end=None):
    """gapped strings (start -> end order) from ops (end -> start order)"""
    x, y = bytes(x), bytes(y)
    i, j = end if end is not None else (len(x), len(y))
    a, b = [], []
    for op in ops:
        if op == 0:
            i -= 1
            j -= 1
            a.append(x[i])
            b.append(y[j])
        elif op == 1:
            j -= 1
            a.append(45)
            b.append(y[j])
        else:
            i -= 1
            a.append(x[i])
            b.append(45)
    return bytes(reversed(a)).decode("latin-1"), bytes(reversed(b)).decode("latin-1")
