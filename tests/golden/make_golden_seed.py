#!/usr/bin/env python3
"""Generates tests/golden/seed_segments.json from the reference's seed_aligner DP helpers
(build container only):

    make -C oracle ref_seed
    LD_LIBRARY_PATH=/usr/lib/x86_64-linux-gnu:/opt/conda/lib python tests/golden/make_golden_seed.py

oracle/_ref/libref_seed.so is the unmodified /root/reference/src/main_fmindex.cpp behind extern "C"
wrappers (oracle/ref_seed_harness.cpp).  The fixture holds DATA only: the two score tables as the
reference's score functions return them for every residue pair, and inputs + outputs of
align_segment_globally / compute_lcs_for_segment / perform_sw_in_window on seeded random segments.
"""
import ctypes as C
import json
from pathlib import Path

import numpy as np

ROOT = Path(__file__).resolve().parent.parent.parent
L = C.CDLL(str(ROOT / "oracle" / "_ref" / "libref_seed.so"))
CAP = 4096
DNA = "ACGT"
IUPAC = "ACGTURYSWKMBDHVNX"
PROT = "ARNDCQEGHILKMFPSTWYVBZX*"


def table(mode):
    valid = []
    out = C.c_int()
    for b in range(256):
        if L.ref_seed_score(mode, b, ord("A"), C.byref(out)) == 0:
            valid.append(b)
    sub = {}
    for x in valid:
        for y in valid:
            assert L.ref_seed_score(mode, x, y, C.byref(out)) == 0
            sub[chr(x) + chr(y)] = out.value
    return {"valid": "".join(chr(b) for b in valid), "score": sub}


def rand_seq(rng, n, alpha):
    return "".join(rng.choice(list(alpha), n)) if n else ""


def mutate(rng, s, alpha, ps=0.1, pd=0.04, pi=0.04):
    out = []
    for ch in s:
        r = rng.random()
        if r < pd:
            continue
        out.append(rng.choice(list(alpha)) if r < pd + ps else ch)
        if rng.random() < pi:
            out.append(rng.choice(list(alpha)))
    return "".join(out)


def seg_nw(s1, s2, mode, go, ge):
    sc = C.c_int()
    a1, a2 = C.create_string_buffer(CAP), C.create_string_buffer(CAP)
    rc = L.ref_seg_nw(s1.encode(), len(s1), s2.encode(), len(s2), mode, C.c_double(go), C.c_double(ge), C.byref(sc), a1, a2, CAP)
    assert rc == 0
    return {"score": sc.value, "a1": a1.value.decode(), "a2": a2.value.decode()}


def seg_lcs(s1, s2):
    n = C.c_int()
    l, g1, g2 = (C.create_string_buffer(CAP) for _ in range(3))
    L.ref_seg_lcs(s1.encode(), len(s1), s2.encode(), len(s2), C.byref(n), l, g1, g2, CAP)
    return {"len": n.value, "lcs": l.value.decode(), "g1": g1.value.decode(), "g2": g2.value.decode()}


def win_sw(s1, s2, mode, go, ge, qo, to):
    out = (C.c_int * 5)()
    a1, a2 = C.create_string_buffer(CAP), C.create_string_buffer(CAP)
    rc = L.ref_win_sw(s1.encode(), len(s1), s2.encode(), len(s2), mode, C.c_double(go), C.c_double(ge), qo, to, out, a1, a2, CAP)
    assert rc == 0
    return {"score": out[0], "q_end": out[1], "t_end": out[2], "q_start": out[3], "t_start": out[4],
            "a1": a1.value.decode(), "a2": a2.value.decode()}


def main():
    rng = np.random.default_rng(20250101)
    doc = {"tables": {"dna": table(0), "protein": table(1)}, "nw": [], "lcs": [], "sw": []}
    gaps = [(-5.0, -1.0), (-5.0, -1.0), (-2.0, -2.0), (-11.0, -1.0), (-3.0, -3.0), (-5.5, -0.5)]
    for k in range(150):
        mode = int(k % 3 == 2)
        alpha = PROT if mode else (IUPAC if k % 5 == 0 else DNA)
        m = int(rng.integers(0, 260)) if k % 11 else 0
        s1 = rand_seq(rng, m, alpha)
        s2 = mutate(rng, s1, alpha) if k % 4 else rand_seq(rng, int(rng.integers(0, 260)), alpha)
        if k % 13 == 0:
            s1, s2 = "A" * int(rng.integers(1, 80)), "A" * int(rng.integers(1, 80))  # tie storm
            mode = 0
        go, ge = gaps[k % len(gaps)]
        doc["nw"].append({"s1": s1, "s2": s2, "mode": mode, "go": go, "ge": ge, **seg_nw(s1, s2, mode, go, ge)})
    for k in range(80):
        alpha = DNA if k % 2 else "ACGTacgtN-"
        s1 = rand_seq(rng, int(rng.integers(0, 260)) if k % 9 else 0, alpha)
        s2 = mutate(rng, s1, alpha) if k % 3 else rand_seq(rng, int(rng.integers(0, 260)), alpha)
        doc["lcs"].append({"s1": s1, "s2": s2, **seg_lcs(s1, s2)})
    for k in range(150):
        mode = int(k % 3 == 2)
        alpha = PROT if mode else (IUPAC if k % 5 == 0 else DNA)
        core = rand_seq(rng, int(rng.integers(5, 120)), alpha)
        s1 = rand_seq(rng, int(rng.integers(0, 90)), alpha) + core + rand_seq(rng, int(rng.integers(0, 90)), alpha)
        s2 = rand_seq(rng, int(rng.integers(0, 90)), alpha) + mutate(rng, core, alpha, 0.08, 0.03, 0.03) + rand_seq(rng, int(rng.integers(0, 90)), alpha)
        if k % 17 == 0:
            s1, s2 = "A" * 40, "C" * 50  # no positive cell
            mode = 0
        go, ge = gaps[k % 5]
        doc["sw"].append({"s1": s1, "s2": s2, "mode": mode, "go": go, "ge": ge, "q_off": int(rng.integers(0, 1000)),
                          "t_off": int(rng.integers(0, 1000))})
        c = doc["sw"][-1]
        c.update(win_sw(s1, s2, mode, go, ge, c["q_off"], c["t_off"]))
    out = ROOT / "tests" / "golden" / "seed_segments.json"
    out.write_text(json.dumps(doc, separators=(",", ":")))
    print("wrote", out, out.stat().st_size, "bytes;", len(doc["nw"]), "nw,", len(doc["lcs"]), "lcs,", len(doc["sw"]), "sw cases")


if __name__ == "__main__":
    main()
