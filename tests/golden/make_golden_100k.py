#!/usr/bin/env python3
"""Config 3 at its own size through the UNMODIFIED reference (oracle/_ref/aligner_ref): the synthetic
100 000 x ~100 000 DNA pair of SURVEY.md 8d (seed 1234, ~90 % identity) and the unrelated variant
(seed 5678), --choice 1.  The reference holds (m+1)(n+1) x 5 bytes = 50 GB for each (main.cpp:706, 732)
and takes 4-5 minutes per pair on one core, so the two runs go one after the other and nothing else
that needs memory may run beside them in the 62 GiB build container.  Build container only; what
travels is score + SHA-256 of the files it wrote (tests/golden/big100k.json).

    python tests/golden/make_golden_100k.py [related] [unrelated]
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

from make_golden import summarize, write_fasta  # noqa: E402

from oracle import run_reference  # noqa: E402
from sequencealigner_amd.synth import dna_pair  # noqa: E402

CASES = {"related": dict(n=100_000, seed=1234, related=True), "unrelated": dict(n=100_000, seed=5678, related=False)}


def main():
    out_path = HERE / "big100k.json"
    done = {c["id"]: c for c in json.loads(out_path.read_text())} if out_path.exists() else {}
    for name in (sys.argv[1:] or list(CASES)):
        g = CASES[name]
        x, y = dna_pair(g["n"], seed=g["seed"], related=g["related"])
        tmp = tempfile.mkdtemp(prefix="gold100k_")
        try:
            write_fasta(os.path.join(tmp, "q.fasta"), "seqA synth", x.tobytes().decode())
            write_fasta(os.path.join(tmp, "t.fasta"), "seqB synth", y.tobytes().decode())
            t0 = time.time()
            rc, _so, se, files, outdir = run_reference(os.path.join(tmp, "q.fasta"), os.path.join(tmp, "t.fasta"), 1, "dna")
            shutil.rmtree(outdir, ignore_errors=True)
            cid = f"synth100k_{name}"
            done[cid] = {"id": cid, "synth": g, "m": len(x), "n": len(y), "mode": "dna", "choice": 1, "rc": rc,
                         "stderr": se.decode("latin-1"), "files": summarize(files, small=False)}
            print(cid, "rc", rc, f"{time.time() - t0:.0f} s", sorted(files), flush=True)
        finally:
            shutil.rmtree(tmp, ignore_errors=True)
        out_path.write_text(json.dumps(list(done.values()), indent=1))


if __name__ == "__main__":
    main()
