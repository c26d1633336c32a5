"""Randomised GPU-vs-oracle parity: shapes, alphabets, gap settings, rows per lane, both fill
schedules, fill_group, few workgroups (several rounds of tickets), checkpoint bands, score-only calls, bit-vector / cell-by-cell LCS.  `python tools/fuzz_gpu.py SECONDS [SEED]`; prints a
line per 200 cases and stops at the first mismatch with the reproducer."""
import sys
import time
from pathlib import Path

import numpy as np

ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))
sys.path.insert(0, str(ROOT / "tests"))
import sequencealigner_amd as sa
from conftest import DNA, PROT20, mutate, rand_seq
from oracle import Oracle

budget = float(sys.argv[1]) if len(sys.argv) > 1 else 60.0
seed = int(sys.argv[2]) if len(sys.argv) > 2 else 1
rng = np.random.default_rng(seed)
eng, orc = sa.Engine(0), Oracle()
t0, n_cases = time.time(), 0
GAPS = [(-5.0, -1.0), (-2.0, -2.0), (-11.0, -1.0), (-1.0, -3.0), (0.0, 0.0), (-5.5, -0.5), (-3.0, 0.0)]
while time.time() - t0 < budget:
    kind = ("nw", "sw", "lcs")[int(rng.integers(0, 3))]
    mode = "dna" if rng.random() < 0.7 else "protein"
    alpha = {"dna": DNA if rng.random() < 0.8 else "ACGTRYSWKMBDHVN", "protein": PROT20}[mode]
    big = rng.random() < 0.15
    m = int(rng.integers(1, 9000 if big else 1500))
    x = rand_seq(rng, m, alpha if rng.random() < 0.9 else alpha[:2])
    r = rng.random()
    if r < 0.6:
        y = mutate(rng, x, alpha, float(rng.random() * 0.3), float(rng.random() * 0.1), float(rng.random() * 0.1))
    elif r < 0.9:
        y = rand_seq(rng, int(rng.integers(1, 9000 if big else 1500)), alpha)
    else:
        y = x[int(rng.integers(0, m)):] + x[:int(rng.integers(0, m))]
    if not y:
        y = alpha[0]
    go, ge = GAPS[int(rng.integers(0, len(GAPS)))]
    K = int(rng.choice([0, 0, 0, 1, 2, 4, 8, 16]))  # 0: automatic (LCS: the bit-vector kernel); 16: LCS only, 8 elsewhere
    persistent = int(rng.random() < 0.7)
    band = int(rng.choice([0, 0, 0, 1, 2, 5]))
    flags = 2 if rng.random() < 0.1 else 0
    eng.set_tuning(K, 0)
    eng.set_option("persistent", persistent)
    eng.set_option("checkpoint_band", band)
    eng.set_option("lcs_bits", int(rng.random() < 0.85))
    group = int(rng.random() < 0.4)
    eng.set_option("group", group)  # fill_group instead of fill_persistent (NW / SW, persistent schedule)
    eng.set_option("max_workgroups", int(rng.choice([0, 0, 0, 1, 2, 7])))
    if kind == "lcs":
        if rng.random() < 0.15:  # byte text over a wide alphabet (up to 255 symbols: the bit-vector kernel; all 256: the cell kernels)
            nsym = int(rng.choice([64, 65, 127, 128, 129, 160, 161, 200, 255, 256]))
            syms = rng.permutation(256)[:nsym].astype(np.uint8)
            x = syms[rng.integers(0, nsym, len(x))]
            y = np.concatenate([x[len(x) // 5:], syms[rng.integers(0, nsym, int(rng.integers(1, 400)))], x[:len(x) // 7]])
            if rng.random() < 0.5 and len(x) >= nsym:
                x[:nsym] = syms
        g, o = eng.lcs(x, y, flags=flags), orc.lcs(x, y)
    else:
        g, o = getattr(eng, kind)(x, y, mode, go, ge, flags=flags), getattr(orc, kind)(x, y, mode, go, ge)
    ok = g.score == o.score and (kind != "sw" or tuple(g.end) == tuple(o.end))
    if not flags:
        ok = ok and tuple(g.start) == tuple(o.start) and np.array_equal(g.ops, o.ops)
    n_cases += 1
    if not ok:
        print(f"MISMATCH case {n_cases}: kind={kind} mode={mode} m={len(x)} n={len(y)} go={go} ge={ge} K={K} persistent={persistent} group={group} "
              f"band={band} flags={flags} seed={seed}: gpu score {g.score} end {g.end} vs oracle {o.score} {o.end}", flush=True)
        Path(ROOT / "gpurun_out").mkdir(exist_ok=True)
        (ROOT / "gpurun_out" / "fuzz_fail.txt").write_text(f"{kind}\n{mode}\n{go} {ge} {K} {persistent} {band} {flags}\n{x}\n{y}\n")
        sys.exit(1)
    if n_cases % 200 == 0:
        print(f"{n_cases} cases ok, {time.time() - t0:.0f} s", flush=True)
print(f"FUZZ_OK {n_cases} cases in {time.time() - t0:.0f} s (seed {seed})", flush=True)
