#!/usr/bin/env python3
"""Smith-Waterman at 20-30 k through the UNMODIFIED reference (oracle/_ref/aligner_ref, --choice 2): the two
synthetic pairs big.json already holds for NW / LCS (20 000 seed 1234, 30 000 seed 77; ~90 % identity) and one
unrelated 20 000 x 20 000 pair (seed 5678), whose local alignment is short and whose first row-major maximum
is somewhere in the middle of the matrix; and two TALL matrices (70 000 rows x 4-5 000 columns), which at one row
per lane are more strips than a persistent launch has waves.  The reference keeps three int matrices for the fill and four
more for its prefix recomputation (main.cpp:1060-1065, 1197-1224): 11-25 GB, one run at a time.
Build container only; what travels is score, end cell and SHA-256 of the files it wrote
(tests/golden/big_sw.json).

    python tests/golden/make_golden_sw_big.py [case ...]
"""
from __future__ import annotations

import json
import os
import shutil
import sys
import tempfile
import time
from pathlib import Path

HERE = Path(__file__).resolve().parent
ROOT = HERE.parent.parent
sys.path.insert(0, str(ROOT))
sys.path.insert(0, str(HERE))

sys.path.insert(0, str(ROOT / "tests"))

from make_golden import summarize, write_fasta  # noqa: E402
from make_golden_shared import tall_pair  # noqa: E402

from oracle import run_reference  # noqa: E402
from sequencealigner_amd.synth import dna_pair  # noqa: E402

CASES = {
    "synth_20000_1234_c2": dict(n=20000, seed=1234, related=True),
    "synth_30000_77_c2": dict(n=30000, seed=77, related=True),
    "unrel_20000_5678_c2": dict(n=20000, seed=5678, related=False),
    # tall matrices: 70 000 rows are 1 094 one-row-per-lane strips, more than a persistent launch has waves
    "tall_70000x5000_99_c2": dict(m=70000, n=5000, seed=99, related=True),
    "tall_70000x4000_98_c2": dict(m=70000, n=4000, seed=98, related=False),
}


def make_pair(g):
    if "m" in g:
        return tall_pair(g["m"], g["n"], g["seed"], g["related"])
    return dna_pair(g["n"], seed=g["seed"], related=g["related"])


def main():
    out_path = HERE / "big_sw.json"
    done = {c["id"]: c for c in json.loads(out_path.read_text())} if out_path.exists() else {}
    for cid in (sys.argv[1:] or list(CASES)):
        g = CASES[cid]
        x, y = make_pair(g)
        tmp = tempfile.mkdtemp(prefix="goldsw_")
        try:
            write_fasta(os.path.join(tmp, "q.fasta"), "seqA synth", x.tobytes().decode())
            write_fasta(os.path.join(tmp, "t.fasta"), "seqB synth", y.tobytes().decode())
            t0 = time.time()
            rc, _so, se, files, outdir = run_reference(os.path.join(tmp, "q.fasta"), os.path.join(tmp, "t.fasta"), 2, "dna")
            shutil.rmtree(outdir, ignore_errors=True)
            done[cid] = {"id": cid, "synth": g, "m": len(x), "n": len(y), "mode": "dna", "choice": 2, "rc": rc,
                         "stderr": se.decode("latin-1"), "files": summarize(files, small=False)}
            print(cid, "rc", rc, f"{time.time() - t0:.0f} s", sorted(files), flush=True)
        finally:
            shutil.rmtree(tmp, ignore_errors=True)
        out_path.write_text(json.dumps(list(done.values()), indent=1))


if __name__ == "__main__":
    main()
