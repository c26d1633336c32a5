"""Randomised GPU-vs-oracle parity of the many-pair entry points (sa_sw_batch, sa_sw_batch_align,
sa_nw_batch_align, sa_lcs_batch_align, sa_sw_window_batch): random batch sizes, lengths around the
row-group and kernel-limit edges (0, 63..65, 1023..1025, > the batch kernels' limits), alphabets,
related / unrelated / tie-storm pairs, integral and fractional gap settings and code budgets that cut the batch into
chunks.  `python tools/fuzz_gpu_batch.py SECONDS [SEED]`; stops at the first mismatch."""
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
GAPS = [(-5.0, -1.0), (-2.0, -2.0), (-11.0, -1.0), (-1.0, -3.0), (0.0, 0.0), (-3.0, 0.0), (-10.0, -1.0), (-5.5, -0.5), (-4.25, -1.75),
        (-0.5, -0.5), (-7.9, -0.1), (-5.0, -0.5)]
EDGE = [0, 1, 2, 63, 64, 65, 127, 128, 129, 511, 512, 513, 1023, 1024]


def rand_len(limit):
    r = rng.random()
    if r < 0.15:
        return int(rng.choice(EDGE))
    if r < 0.17:
        return int(rng.integers(1025, limit))
    if r < 0.6:
        return int(rng.integers(1, 120))
    return int(rng.integers(1, 1025))


def fail(msg, pool, ox, lx, oy, ly, k):
    print("MISMATCH " + msg, flush=True)
    out = ROOT / "gpurun_out"
    out.mkdir(exist_ok=True)
    (out / "fuzz_batch_fail.txt").write_text(
        f"{msg}\n{bytes(pool[ox[k]:ox[k] + lx[k]]).decode()}\n{bytes(pool[oy[k]:oy[k] + ly[k]]).decode()}\n")
    sys.exit(1)


t0, n_batches, n_pairs_total = time.time(), 0, 0
while time.time() - t0 < budget:
    what = ("sw", "sw_align", "nw_align", "lcs_align", "window")[int(rng.integers(0, 5))]
    mode = "dna" if rng.random() < 0.5 else "protein"
    alpha = {"dna": DNA if rng.random() < 0.8 else "ACGTRYSWKMBDHVN", "protein": PROT20}[mode]
    if rng.random() < 0.1:
        alpha = alpha[:1 + int(rng.integers(0, 2))]  # tie storms
    n_pairs = int(rng.integers(1, 120))
    go, ge = GAPS[int(rng.integers(0, len(GAPS)))]
    if what == "window" and go == 0.0:
        go = -4.0
    seqs, ox, lx, oy, ly, pos = [], [], [], [], [], 0
    for _ in range(n_pairs):
        m = rand_len(1500)
        x = rand_seq(rng, m, alpha) if m else ""
        r = rng.random()
        if m and r < 0.5:
            y = mutate(rng, x, alpha, float(rng.random() * 0.3), float(rng.random() * 0.1), float(rng.random() * 0.1))
            if what == "window" and len(y) > 3000:
                y = y[:3000]
        elif m and r < 0.6:  # a window around the query, as the seed_aligner cuts it
            y = rand_seq(rng, int(rng.integers(0, 40)), alpha) + x + rand_seq(rng, int(rng.integers(0, 40)), alpha)
        else:
            n = rand_len(4300)
            y = rand_seq(rng, n, alpha) if n else ""
        if what == "window":  # beyond the batch kernels' limits the window entry point refuses
            x, y = x[:1024], y[:3800]
        for s, o, l in ((x, ox, lx), (y, oy, ly)):
            o.append(pos)
            l.append(len(s))
            seqs.append(s)
            pos += len(s)
    pool = np.frombuffer(("".join(seqs) + "A").encode(), np.uint8)
    ox, oy = np.array(ox), np.array(oy)
    lx, ly = np.array(lx, np.int32), np.array(ly, np.int32)
    budget_mb = int(rng.choice([32768, 32768, 1, 2, 8]))
    eng.set_option("batch_trace_budget_mb", budget_mb)
    tag = f"{what} batch {n_batches} seed {seed} mode {mode} go {go} ge {ge} budget {budget_mb} pairs {n_pairs}"

    def seq(k):
        return pool[ox[k]:ox[k] + lx[k]], pool[oy[k]:oy[k] + ly[k]]

    if what == "sw":
        sc, ei, ej, _ms, rc = eng.sw_batch(pool, ox, lx, oy, ly, mode, go, ge)
        orc_rc, osc, oei, oej = orc.sw_score_batch(pool, ox, lx, oy, ly, mode, go, ge, threads=8)
        assert rc == 0 and orc_rc == 0
        bad = np.nonzero((sc != osc) | (ei != oei) | (ej != oej))[0]
        if len(bad):
            k = int(bad[0])
            fail(f"{tag} pair {k} m {lx[k]} n {ly[k]}: gpu {sc[k], ei[k], ej[k]} oracle {osc[k], oei[k], oej[k]}", pool, ox, lx, oy, ly, k)
    elif what == "window":
        sc, ei, ej, si, sj, ln, _ms, rc = eng.sw_window_batch(pool, ox, lx, oy, ly, mode, go, ge)
        assert rc == 0
        for k in range(n_pairs):
            x, y = seq(k)
            if not len(x) or not len(y):
                want = (0, 0, 0, 0, 0, 0)
            else:
                w = orc.sw(x, y, mode, go, ge, keep_dp=True)
                i, j = w.end
                n = 0
                while w.score > 0 and i - n > 0 and j - n > 0 and w.dp[i - n, j - n] > 0:
                    n += 1
                want = (w.score, i, j, i - n, j - n, n) if w.score > 0 else (0, 0, 0, 0, 0, 0)
            got = tuple(int(v[k]) for v in (sc, ei, ej, si, sj, ln))
            if got != want:
                fail(f"{tag} pair {k} m {lx[k]} n {ly[k]}: gpu {got} oracle {want}", pool, ox, lx, oy, ly, k)
    else:
        if what == "sw_align":
            b = eng.sw_batch_align(pool, ox, lx, oy, ly, mode, go, ge)
        elif what == "nw_align":
            b = eng.nw_batch_align(pool, ox, lx, oy, ly, mode, go, ge)
        else:
            b = eng.lcs_batch_align(pool, ox, lx, oy, ly)
        assert b.rc == 0
        for k in range(n_pairs):
            x, y = seq(k)
            if what == "sw_align":
                if not len(x) or not len(y):
                    ok = b.score[k] == 0 and b.n_ops[k] == 0
                    w = None
                else:
                    w = orc.sw(x, y, mode, go, ge)
                    ok = (int(b.score[k]), int(b.end_i[k]), int(b.end_j[k]), int(b.start_i[k]), int(b.start_j[k])) == (
                        w.score, *w.end, *w.start) and np.array_equal(b.ops(k), w.ops)
            else:
                if not len(x) and not len(y):
                    ok, w = b.n_ops[k] == 0, None
                else:
                    w = orc.nw(x, y, mode, go, ge) if what == "nw_align" else orc.lcs(x, y)
                    ok = np.array_equal(b.ops(k), w.ops) and int(b.score[k]) == w.score
            if not ok:
                fail(f"{tag} pair {k} m {lx[k]} n {ly[k]}: gpu score {b.score[k]} n_ops {b.n_ops[k]} oracle {w.score if w else None}",
                     pool, ox, lx, oy, ly, k)
        del b
    n_batches += 1
    n_pairs_total += n_pairs
    if n_batches % 20 == 0:
        print(f"{n_batches} batches / {n_pairs_total} pairs ok, {time.time() - t0:.0f} s", flush=True)
eng.set_option("batch_trace_budget_mb", 32768)
print(f"FUZZ_BATCH_OK {n_batches} batches, {n_pairs_total} pairs in {time.time() - t0:.0f} s (seed {seed})", flush=True)
