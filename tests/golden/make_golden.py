#!/usr/bin/env python3
"""Generate tests/golden/*.json by running the UNMODIFIED reference (oracle/_ref/aligner_ref,
built by `make -C oracle ref` from /root/reference/src/main.cpp).  Build container only: the
reference cannot travel, the vectors it produced can.

Every vector = inputs (sequences or a generator seed) + what the reference wrote for them:
scores, the SW end cell, SHA-256 of the alignment / path files (full text for small cases)
and the stats JSON with time_ms zeroed.

    python tests/golden/make_golden.py            # everything (~10 min on 8 cores)
"""
from __future__ import annotations

import hashlib
import json
import os
import re
import shutil
import sys
import tempfile
from concurrent.futures import ProcessPoolExecutor
from pathlib import Path

import numpy as np

HERE = Path(__file__).resolve().parent
ROOT = HERE.parent.parent
sys.path.insert(0, str(ROOT))
sys.path.insert(0, str(ROOT / "tests"))

from oracle import run_reference  # noqa: E402

DNA = "ACGT"
IUPAC = "ACGTURYSWKMBDHVNX"
PROT20 = "ARNDCQEGHILKMFPSTWYV"
PROT24 = "ARNDCQEGHILKMFPSTWYVBZX*"


def sha(b: bytes) -> str:
    return hashlib.sha256(b).hexdigest()


def norm_stats(b: bytes) -> str:
    return re.sub(r'("time_ms":\s*)\d+', r"\g<1>0", b.decode())


def write_fasta(path, header, seq, width=70):
    with open(path, "w") as f:
        f.write(">" + header + "\n")
        for i in range(0, len(seq), width):
            f.write(seq[i:i + width] + "\n")


def summarize(files: dict, small: bool) -> dict:
    out = {}
    for name, data in files.items():
        if name.endswith("_stats.json"):
            out[name] = norm_stats(data)
        else:
            out[name] = {"sha256": sha(data), "bytes": len(data)}
            if small:
                out[name]["text"] = data.decode("latin-1")
    return out


def run_case(case: dict) -> dict:
    """case: {id, x, y, mode, go, ge, choice}.  Sequences are written to temp FASTA files."""
    tmp = tempfile.mkdtemp(prefix="gold_")
    try:
        qa, ta = os.path.join(tmp, "q.fasta"), os.path.join(tmp, "t.fasta")
        write_fasta(qa, "seqA test query", case["x"])
        write_fasta(ta, "seqB test target", case["y"])
        rc, so, se, files, outdir = run_reference(qa, ta, case.get("choice", 4), case["mode"], case.get("go"), case.get("ge"))
        shutil.rmtree(outdir, ignore_errors=True)
        res = dict(case)
        res["rc"] = rc
        res["stderr"] = se.decode("latin-1")
        res["files"] = summarize(files, small=(len(case["x"]) + len(case["y"]) <= 200))
        return res
    finally:
        shutil.rmtree(tmp, ignore_errors=True)


def rand_seq(rng, n, alphabet):
    a = np.frombuffer(alphabet.encode(), dtype=np.uint8)
    return a[rng.integers(0, len(a), size=n)].tobytes().decode()


def mutate(rng, s, alphabet, sub=0.06, dele=0.02, ins=0.02):
    out = []
    for ch in s:
        r = rng.random()
        if r < dele:
            continue
        out.append(alphabet[rng.integers(0, len(alphabet))] if r < dele + sub else ch)
        if rng.random() < ins:
            out.append(alphabet[rng.integers(0, len(alphabet))])
    return "".join(out)


def random_cases():
    rng = np.random.default_rng(20261111)
    cases = []

    def add(x, y, mode, go=None, ge=None, tag=""):
        cases.append({"id": f"r{len(cases):04d}{tag}", "x": x, "y": y, "mode": mode, "go": go, "ge": ge})

    lens = [0, 1, 2, 3, 5, 17, 63, 64, 65, 127, 128, 129, 200, 255, 256, 257, 400, 511, 513, 700, 1000, 1500, 2100, 3000]
    for mode, full, core in (("dna", IUPAC, DNA), ("protein", PROT24, PROT20)):
        for k in range(90):
            m = int(lens[rng.integers(0, len(lens))])
            n = int(lens[rng.integers(0, len(lens))])
            if k % 3 == 0:   # related pair, core alphabet
                x = rand_seq(rng, m, core)
                y = mutate(rng, x, core, 0.08, 0.03, 0.03)
            elif k % 3 == 1:  # unrelated, full alphabet incl. ambiguity codes
                x, y = rand_seq(rng, m, full), rand_seq(rng, n, full)
            else:            # unrelated, core alphabet
                x, y = rand_seq(rng, m, core), rand_seq(rng, n, core)
            add(x, y, mode)
    # tie storms (low complexity / tandem repeats)
    for x, y in [("A" * 300, "A" * 257), ("AC" * 200, "CA" * 180), ("ACGT" * 100, "ACGT" * 90 + "TTTT"),
                 ("A" * 500, "T" * 400), ("AAAAAAAAAC" * 40, "AAAAAAAAAG" * 35), ("GATTACA" * 64, "GATACA" * 70),
                 ("ACGTRYSWKMBDHVNACGT", "ACGTSWRYMKVHDBNACGT"), ("", "ACGTACGT"), ("ACGTACGT", ""), ("A", "A"), ("A", "C")]:
        add(x, y, "dna", tag="t")
    for x, y in [("W" * 100, "W" * 90), ("ARND" * 50, "ARNE" * 55), ("MKV" * 100, "MKKV" * 70)]:
        add(x, y, "protein", tag="t")
    # integral non-default gaps and fractional gaps (quirk Q3)
    for go, ge in [(-10.0, -1.0), (-2.0, -2.0), (-1.0, -3.0), (0.0, 0.0), (-11.0, -1.0), (-3.0, 0.0),
                   (-5.5, -0.5), (-4.25, -1.75), (-0.5, -0.5), (-7.9, -0.1)]:
        x = rand_seq(rng, 420, DNA)
        add(x, mutate(rng, x, DNA, 0.1, 0.05, 0.05), "dna", go, ge, tag="g")
        xp = rand_seq(rng, 333, PROT20)
        add(xp, mutate(rng, xp, PROT20, 0.3, 0.05, 0.05), "protein", go, ge, tag="g")
    return cases


def error_cases():
    return [{"id": f"e{k}", "x": x, "y": y, "mode": mode, "go": None, "ge": None, "choice": 1}
            for k, (x, y, mode) in enumerate([("acgtACGT", "ACGTACGT", "dna"), ("ACGT", "ACzT", "dna"), ("ACGJ", "ACGT", "dna"),
                                              ("AJGT", "AqGT", "dna"), ("ARNDO", "ARND", "protein"), ("ARND", "ARJD", "protein")])]


def table_cases():
    cases = []
    for mode, letters in (("dna", "ACGTURYSWKMBDHVNX"), ("protein", PROT24)):
        for a in letters:
            for b in letters:
                cases.append({"id": f"s_{mode}_{a}{b}", "x": a, "y": b, "mode": mode, "go": None, "ge": None, "choice": 1})
    return cases


def bundled():
    out = []
    for a, b, mode in [("dna1", "dna2", "dna"), ("P00533", "P08581", "protein"), ("prot3", "prot4", "protein"),
                       ("prot1", "prot2", "protein"), ("dna2", "dna1", "dna")]:
        rc, so, se, files, outdir = run_reference(HERE / "inputs" / f"{a}.fasta", HERE / "inputs" / f"{b}.fasta", 4, mode)
        shutil.rmtree(outdir, ignore_errors=True)
        out.append({"id": f"{a}_{b}", "query": f"{a}.fasta", "target": f"{b}.fasta", "mode": mode, "rc": rc,
                    "stderr": se.decode("latin-1"), "files": summarize(files, small=False)})
        print("bundled", a, b, flush=True)
    return out


def big():
    """Large NW cases: score + hashes only.  dna3 x dna4 takes ~90 s and ~17 GB in the reference."""
    out = []
    rc, so, se, files, outdir = run_reference(HERE / "inputs" / "dna3.fasta", HERE / "inputs" / "dna4.fasta", 1, "dna")
    shutil.rmtree(outdir, ignore_errors=True)
    out.append({"id": "dna3_dna4", "query": "dna3.fasta", "target": "dna4.fasta", "mode": "dna", "rc": rc,
                "files": summarize(files, small=False)})
    print("big dna3 x dna4 done", flush=True)
    from sequencealigner_amd.synth import dna_pair

    for n, seed in ((20000, 1234), (30000, 77)):
        x, y = dna_pair(n, seed=seed)
        tmp = tempfile.mkdtemp(prefix="gold_")
        write_fasta(os.path.join(tmp, "q.fasta"), "seqA synth", x.tobytes().decode())
        write_fasta(os.path.join(tmp, "t.fasta"), "seqB synth", y.tobytes().decode())
        for choice in (1, 3):
            rc, so, se, files, outdir = run_reference(os.path.join(tmp, "q.fasta"), os.path.join(tmp, "t.fasta"), choice, "dna")
            shutil.rmtree(outdir, ignore_errors=True)
            out.append({"id": f"synth_{n}_{seed}_c{choice}", "synth": {"n": n, "seed": seed}, "mode": "dna", "choice": choice,
                        "rc": rc, "files": summarize(files, small=False)})
        shutil.rmtree(tmp, ignore_errors=True)
        print("big synth", n, flush=True)
    return out


def dumps():
    """--txt / --binary matrix dumps (main.cpp:422-528; call sites :825-831, :1364-1368, :1680-1691) exactly as the
    reference writes them: every file it produced, bytes in base64 (small cases only).  Three flag settings per pair:
    --txt, --binary, and both (the reference lets --binary win for the S / L matrices, main.cpp:825-829)."""
    import base64

    rng = np.random.default_rng(424242)
    xp = rand_seq(rng, 37, PROT20)
    cases = [("d0", "ACGTTGCAAGCT", "ACGTGCATAGT", "dna"), ("d1", rand_seq(rng, 70, DNA), rand_seq(rng, 23, DNA), "dna"),
             ("d2", "ACGTRYSWKMBDHVN", "NACGTAAC", "dna"), ("p0", xp, mutate(rng, xp, PROT20, 0.2, 0.1, 0.1), "protein"),
             ("p1", "W", "ARNDW", "protein"), ("e0", "", "ACGT", "dna"), ("e1", "ACG", "", "dna")]
    out = []
    for cid, x, y, mode in cases:
        tmp = tempfile.mkdtemp(prefix="gold_")
        try:
            qa, ta = os.path.join(tmp, "q.fasta"), os.path.join(tmp, "t.fasta")
            write_fasta(qa, "seqA dump query", x)
            write_fasta(ta, "seqB dump target", y)
            for flags in (["--txt"], ["--binary"], ["--txt", "--binary"]):
                rc, so, se, files, outdir = run_reference(qa, ta, 4, mode, extra=flags)
                shutil.rmtree(outdir, ignore_errors=True)
                out.append({"id": cid + "_" + "".join(f.strip("-")[0] for f in flags), "x": x, "y": y, "mode": mode, "flags": flags, "rc": rc,
                            "stderr": se.decode("latin-1"),
                            "files": {k: base64.b64encode(v).decode() for k, v in sorted(files.items()) if not k.endswith("_stats.json")}})
                print("dumps", out[-1]["id"], rc, sorted(files), flush=True)
        finally:
            shutil.rmtree(tmp, ignore_errors=True)
    return out


def main():
    only = sys.argv[1:] or ["random", "errors", "tables", "bundled", "big", "dumps"]
    with ProcessPoolExecutor(max_workers=6) as ex:
        if "random" in only:
            res = list(ex.map(run_case, random_cases(), chunksize=4))
            (HERE / "random_pairs.json").write_text(json.dumps(res, indent=0))
            print("random", len(res), flush=True)
        if "errors" in only:
            res = list(ex.map(run_case, error_cases()))
            (HERE / "error_cases.json").write_text(json.dumps(res, indent=1))
        if "tables" in only:
            res = list(ex.map(run_case, table_cases(), chunksize=16))
            tab = {}
            for r in res:
                m = re.search(r'"score":\s*(-?\d+)', r["files"]["global_stats.json"])
                tab[r["id"]] = int(m.group(1))
            (HERE / "sub_tables.json").write_text(json.dumps(tab, indent=0))
            print("tables", len(tab), flush=True)
    if "bundled" in only:
        (HERE / "bundled.json").write_text(json.dumps(bundled(), indent=1))
    if "big" in only:
        (HERE / "big.json").write_text(json.dumps(big(), indent=1))
    if "dumps" in only:
        (HERE / "dumps.json").write_text(json.dumps(dumps(), indent=0))


if __name__ == "__main__":
    main()
