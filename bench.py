#!/usr/bin/env python3
"""bench.py -- headline benchmark of the DP alignment hot path (BASELINE.json metric).

One "step" = one complete Needleman-Wunsch alignment (H2D of the two sequences, query
profile, anti-diagonal fill with 2-bit traceback codes, traceback walk, D2H of the op list)
of the synthetic ~90 %-identity DNA pair of SURVEY.md 8d (config 4: 1 Mb x 1 Mb; the timed
window mirrors the reference's own time_ms window, main.cpp:736 -> :820).

  python bench.py [--gpus N] [--steps K] [--warmup W] [--size L] [--kind nw|lcs|sw]

N = 1 runs the whole matrix on one MI355X.  N > 1 column-stripes the same pair over N ranks, one
process per GPU over RCCL (DESIGN.md "Multi-GPU"): `python bench.py --gpus N` starts the N ranks
itself (torch.distributed.run as a CHILD process, before anything here touches HIP); started by
`python -m torch.distributed.run --nproc-per-node N bench.py --gpus N` it is one of the ranks.
The line also carries the config-5 half of BASELINE.json's metric (batched SW pairs/s) as a
nested "batch" object measured in the same run (pairs sharded over the ranks; --no-batch skips it).

  python bench.py --gpus 2 --launch-check     # CPU: two gloo ranks report, no GPU work

Prints ONE JSON line on rank 0.
"""
from __future__ import annotations

import argparse
import json
import os
import sys
import time
from pathlib import Path

ROOT = Path(__file__).resolve().parent
if str(ROOT) not in sys.path:
    sys.path.insert(0, str(ROOT))

# Column stripes run one persistent kernel per GPU while RCCL moves the borders on other streams:
# every stream involved needs a hardware queue of its own (streams that share one are serialised,
# measured: tools/probe_queue_sharing.py), and HIP's default of 4 queues per process is fewer than
# the streams of this program + torch + RCCL.  Must be set before the HIP runtime starts.
if int(os.environ.get("WORLD_SIZE", "1")) > 1:
    os.environ["GPU_MAX_HW_QUEUES"] = "16"

import numpy as np


def launch_ranks(n, argv):
    """`bench.py --gpus N` outside torch.distributed.run: start the N ranks as a child process (never an
    exec: nothing in this process has touched the GPU yet, and nothing will), pass rank 0's JSON line
    through (the child inherits stdout) and return the child's exit code.  The reference's analogue is
    `mpirun -np P aligner` (main.cpp:1720-1723: MPI_Init, rank, size)."""
    import socket
    import subprocess

    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    env = dict(os.environ)
    env["GPU_MAX_HW_QUEUES"] = "16"  # see below: one hardware queue per stream of the overlapped stripes
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    env.setdefault("OMP_NUM_THREADS", "2")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={n}", "--master-addr", "127.0.0.1",
           "--master-port", str(port), str(Path(__file__).resolve())] + list(argv)
    return subprocess.run(cmd, env=env).returncode


# algorithmic work per cell: int32 VALU lane-ops and HBM bytes
OPS_PER_CELL = {"nw": 12, "sw": 14, "lcs": 6}
OPS_CODE = {"nw": 4, "sw": 4, "lcs": 2}  # of which: the direction code (cmp/select/pack), absent from code-free passes
HBM_BYTES_PER_CELL = 0.25
VALU_PEAK_TOPS = 39.3  # 256 CU x 4 SIMD x 16 lanes x 2.4 GHz int32 lane-ops/s: one wave64 instruction per 4 cycles per SIMD
# Measured with inline-asm streams (profiles/r02_probe_valu_asm.txt): v_max_i32 / v_max3_i32 / SDWA / DPP / v_alignbit issue at
# 4.25 cycles (36.9 T) at any occupancy, v_add_u32 at 3.0 (52 T) and v_and_b32 at 2.3 (68 T) once a SIMD holds two waves.  For the
# NW recurrence's 12 ops (5 adds, 7 four-cycle ops) that mix gives 12 / (5 / 52.3 + 7 / 36.9) = 42.1 T: quoted beside `frac`.
VALU_PEAK_MIX_TOPS = {"nw": 42.1, "sw": 41.2, "lcs": 38.8}  # SW: 5 adds + 9 four-cycle ops; LCS: 1 add + 5
HBM_PEAK_GBS = 8000.0


def cpu_baseline(x, y, kind, sample):
    """The oracle (CPU restatement of the reference loops, full-matrix memory behaviour) timed
    on ONE host core on a bounded slice of the same pair.  One core is the reference's real
    parallelism for one pair (globalalign/lcs are replicated, localalign is a
    serial chain)."""
    from oracle import Oracle

    orc = Oracle()
    xs, ys = x[:sample], y[:sample]
    t0 = time.perf_counter()
    if kind == "lcs":
        r = orc.lcs(xs, ys, keep_dp=True)
    else:
        r = getattr(orc, kind)(xs, ys, "dna", keep_dp=True)
    dt = time.perf_counter() - t0
    assert r.rc == 0
    cells = len(xs) * len(ys) * (2 if kind == "sw" else 1)
    return {
        "value": round(len(xs) * len(ys) / dt / 1e9, 4),
        "unit": "GCUPS",
        "cores": 1,
        "kind": "port",
        "sample": f"{kind} fill+traceback of the first {len(xs)}x{len(ys)} residues of the same pair "
                  f"({cells / 1e6:.0f} M cell updates, {dt:.1f} s), oracle/cpu_ref.c -O3, 5 B/cell matrices like the reference",
    }, r


def reference_baseline(x, y, kind, sample):
    """The reference itself: oracle/_ref/aligner_ref is the UNMODIFIED /root/reference/src/main.cpp
    compiled in the build container (oracle/Makefile `ref`); it travels to the GPU box with the tree.
    One process = one core, which is the reference's real parallelism for one pair.  The number is
    its own time_ms (main.cpp:736 -> :820: fill + traceback, no file I/O).  None if it cannot run here."""
    import json as _json
    import shutil
    import tempfile

    from oracle import pyoracle

    if kind == "lcs" or pyoracle.ref_binary() is None:  # lcs writes no stats file
        return None
    tmp = tempfile.mkdtemp(prefix="sa_ref_")
    try:
        q, t = Path(tmp) / "q.fa", Path(tmp) / "t.fa"
        q.write_bytes(b">q\n" + bytes(x[:sample]) + b"\n")
        t.write_bytes(b">t\n" + bytes(y[:sample]) + b"\n")
        rc, _out, _err, files, outdir = pyoracle.run_reference(q, t, 1 if kind == "nw" else 2, "dna")
        shutil.rmtree(outdir, ignore_errors=True)
        name = "global_stats.json" if kind == "nw" else "local_stats.json"
        if rc != 0 or name not in files:
            return None
        st = _json.loads(files[name])
        ms = float(st["time_ms"])
        if ms <= 0:
            return None
        return {"value": round(sample * sample / (ms / 1e3) / 1e9, 4), "unit": "GCUPS", "cores": 1, "kind": "reference",
                "sample": f"the compiled reference (oracle/_ref/aligner_ref, unmodified main.cpp, -O3) on the first {sample}x{sample} "
                          f"residues of the same pair: its own time_ms = {ms:.0f} ms, score {st['score']}"}
    except Exception:  # noqa: BLE001 -- any failure: fall back to the port
        return None
    finally:
        shutil.rmtree(tmp, ignore_errors=True)


def bench_small_pairs(eng, no_cpu):
    """BASELINE.json configs 1 and 2, the latency cases: dna1 x dna2 NW (8 416 x 719, EDNAFULL) and P00533 x P08581 SW
    (1 210 x 1 390, EBLOSUM62).  Per case: one warm C-ABI call (wall and fill-kernel ms, median of 25 after 3 warm-up
    calls); one COLD process of the drop-in CLI, start to result files written (what the reference's caller pays per
    request: alignment_runner.py:176-191 starts one process per alignment) with the CLI's own time_ms; and the compiled
    reference (oracle/_ref/aligner_ref) the same way where its binary is present.  main.cpp:1092-1093, 1268-1271."""
    import re
    import shutil
    import statistics
    import subprocess
    import tempfile

    from sequencealigner_amd import host

    inputs = ROOT / "tests" / "golden" / "inputs"
    cli = ROOT / "sequencealigner_amd" / "aligner"
    out = {}
    for name, q, t, mode, choice, kind, stats_file in (("config1_nw_dna1_dna2", "dna1.fasta", "dna2.fasta", "dna", 1, "nw", "global_stats.json"),
                                                      ("config2_sw_P00533_P08581", "P00533.fasta", "P08581.fasta", "protein", 2, "sw", "local_stats.json")):
        if not (inputs / q).exists():
            continue
        _h1, x = host.read_fasta(inputs / q)
        _h2, y = host.read_fasta(inputs / t)
        fn = getattr(eng, kind)
        walls, kerns, tbs = [], [], []
        for k in range(28):
            t0 = time.perf_counter()
            r = fn(x, y, mode)
            dt = (time.perf_counter() - t0) * 1e3
            if k >= 3:
                walls.append(dt)
                kerns.append(r.kernel_fill_ms)
                tbs.append(r.traceback_ms)
        e = {"shape": f"{len(x)}x{len(y)}", "score": int(r.score), "rows_per_lane": r.rows_per_lane,
             "warm_call_ms": round(statistics.median(walls), 3), "warm_fill_kernel_ms": round(statistics.median(kerns), 3),
             "warm_traceback_plus_d2h_ms": round(statistics.median(tbs), 3), "warm_calls": len(walls)}

        def cold(binary, env=None):
            ws, tms = [], []
            for _ in range(3):
                d = tempfile.mkdtemp(prefix="sa_small_")
                try:
                    t0 = time.perf_counter()
                    p = subprocess.run([str(binary), "--query", str(inputs / q), "--target", str(inputs / t), "--choice", str(choice),
                                        "--mode", mode, "--outdir", d], capture_output=True, env=env, timeout=300)
                    w = (time.perf_counter() - t0) * 1e3
                    sf = Path(d) / stats_file
                    if p.returncode != 0 or not sf.exists():
                        return None
                    m_ = re.search(r'"time_ms":\s*(\d+)', sf.read_text())
                    ws.append(w)
                    tms.append(int(m_.group(1)) if m_ else -1)
                finally:
                    shutil.rmtree(d, ignore_errors=True)
            return {"process_wall_ms": round(min(ws), 1), "time_ms_field": int(statistics.median(tms)), "runs": len(ws)}

        if cli.exists():
            e["cold_cli"] = cold(cli)
        if not no_cpu:
            from oracle import pyoracle

            if pyoracle.ref_binary() is not None:
                env = dict(os.environ)
                env["LD_LIBRARY_PATH"] = "/usr/lib/x86_64-linux-gnu:/opt/conda/lib"
                e["reference_cpu"] = cold(pyoracle.ref_binary(), env)
        out[name] = e
    out["note"] = ("warm_call_ms: one sa_nw / sa_sw call through the C-ABI on a live ctx (host encoding, H2D, fill, traceback, D2H), median; "
                   "cold_cli: one process of sequencealigner_amd/aligner from exec to exit, result files written (HIP start-up and sa_create "
                   "included), best of 3; reference_cpu: oracle/_ref/aligner_ref (the unmodified reference, one core) the same way; "
                   "time_ms_field: what each program writes into its stats JSON (fill + traceback only, main.cpp:736 -> :820)")
    return out


def bench_batch(args, rank, local_rank, world, dist, torch, workload=None, nested=False):
    """Config 5: independent protein pairs (300-1000 aa, BLOSUM62, -5/-1), SW score + end cell
    (--workload batch) or full local alignments (--workload batch-align: + direction codes, per-pair
    traceback, op lists back on the host).  Pairs are sharded contiguously over the ranks; no
    data-path collective.  Returns the result line on rank 0 (None elsewhere)."""
    import sequencealigner_amd as sa
    from sequencealigner_amd.synth import protein_pairs

    workload = workload or args.workload
    per_rank = args.pairs // world
    pool, ox, lx, oy, ly = protein_pairs(per_rank, seed=42 + rank)
    cells = int((lx.astype(np.int64) * ly.astype(np.int64)).sum())
    eng = sa.Engine(local_rank)
    align = workload in ("batch-align", "batch-global")
    glob = workload == "batch-global"
    # The timed region starts with the sequence pool resident in HBM (sa_batch_set_device_pool); the rate with
    # the pool handed over as a host buffer (H2D inside the call) is measured afterwards and reported beside it.
    dev_pool = torch.from_numpy(pool).to(f"cuda:{local_rank}")
    torch.cuda.synchronize()

    def run():
        if align:
            r = (eng.nw_batch_align if glob else eng.sw_batch_align)(pool, ox, lx, oy, ly, "protein")
            return (r.score, r.end_i, r.end_j, r.kernel_ms, r.rc, r)
        return eng.sw_batch(pool, ox, lx, oy, ly, "protein")

    def sync():
        torch.cuda.synchronize()
        if dist is not None:
            dist.barrier()
            torch.cuda.synchronize()

    def timed():
        for _ in range(args.warmup):
            out = run()
        sync()
        t0 = time.perf_counter()
        kms = 0.0
        for _ in range(args.steps):
            out = run()
            kms += out[3]
        sync()
        return out, kms, time.perf_counter() - t0

    eng.set_device_pool(dev_pool.data_ptr(), dev_pool.numel())
    out, kms, dt = timed()
    eng.set_device_pool(None, 0)
    _o, _k, dt_host = timed()
    del _o
    tot = torch.tensor([dt, float(cells), kms, dt_host], dtype=torch.float64, device="cuda")
    if dist is not None:
        mx = tot.clone()
        dist.all_reduce(mx, op=dist.ReduceOp.MAX)
        dist.all_reduce(tot, op=dist.ReduceOp.SUM)
        dt, cells_all, kms, dt_host = float(mx[0]), float(tot[1]), float(mx[2]), float(mx[3])
    else:
        cells_all = float(cells)
    line = None
    if rank == 0:
        ops = (12 if glob else 14) if align else 10
        kname = ("nw_batch_kernel<K>" if glob else "sw_batch_kernel<K,TRACE>") if align else "sw_batch_kernel<K>"
        achieved = cells_all * args.steps * ops / (kms / 1e3) / 1e12 / world
        line = {
            "metric": ("batched NW pairs/s (full global alignments)" if glob else "batched SW pairs/s (full local alignments)") if align else "batched SW pairs/s (score + end cell)", "value": round(per_rank * world * args.steps / dt, 1),
            "unit": "pairs/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": round(dt / args.steps * 1e3, 3), "higher_is_better": True, "scaling": "strong",
            "vs_baseline": None, "dtype": "int32", "data": "synthetic",
            "config": {"workload": f"{per_rank * world} protein pairs, lengths U[300,1000], uniform residues, BLOSUM62 -5/-1, "
                                   f"sequence pool resident in HBM, descriptors in / results out inside the timed region" + (", op lists back on the host" if align else ""), "parallelism": f"pair sharding x{world}"},
            "value_note": "`value`: sequence pool resident in HBM when the timed region starts; this is the rate set against cpu_baseline",
            "value_host_pool": round(per_rank * world * args.steps / dt_host, 1),
            "value_host_pool_note": "same calls with the pool as a pageable host buffer (H2D of the sequences inside every call, PCIe-inclusive)",
            "gcups_end_to_end": round(cells_all * args.steps / dt / 1e9, 1),
            "gcups_kernel_only": round(cells_all * args.steps / (kms / 1e3) / 1e9, 1),
            "roofline": {"bound": "valu", "achieved": round(achieved, 3),
                         "peak": VALU_PEAK_TOPS, "unit": "T int32 lane-ops/s",
                         "frac": round(achieved / VALU_PEAK_TOPS, 4),
                         "traffic": batch_pmc_traffic(workload, per_rank), "traffic_note": "HBM bytes per step (all launches of one call) from the PMC passes in "
                         "profiles/r03_batch_pmc.json (WRITE_SIZE + 2 x FETCH_SIZE); algorithmic: the sequences once (m + n bytes per pair) + 44 B of descriptor / result per pair",
                         "kernel": kname, "ops_per_cell": ops},
        }
        if not args.no_cpu and world == 1:
            from oracle import Oracle

            orc = Oracle()
            ns = min(2000, per_rank)
            threads = min(16, os.cpu_count() or 1)
            t1 = time.perf_counter()
            if glob:  # global alignments: the oracle's full NW per pair, one core
                na = min(300, ns)
                for k in range(na):
                    w = orc.nw(pool[ox[k]:ox[k] + lx[k]], pool[oy[k]:oy[k] + ly[k]], "protein")
                    assert w.score == out[0][k] and np.array_equal(out[5].ops(k), w.ops)
                dta = time.perf_counter() - t1
                line["cpu_baseline"] = {"value": round(na / dta, 1), "unit": "pairs/s", "cores": 1, "kind": "port",
                                        "sample": f"first {na} pairs, full globalalign restatement one after the other on one core, "
                                                  f"{dta:.1f} s; every one equals the GPU's alignment"}
                return line
            rc, sc, ei, ej = orc.sw_score_batch(pool, ox[:ns], lx[:ns], oy[:ns], ly[:ns], "protein", threads=threads)
            dtc = time.perf_counter() - t1
            assert rc == 0 and np.array_equal(sc, out[0][:ns]) and np.array_equal(ei, out[1][:ns]) and np.array_equal(ej, out[2][:ns])
            if align:  # and the alignments of a few pairs against the oracle's full SW
                for k in range(0, ns, max(1, ns // 16)):
                    w = orc.sw(pool[ox[k]:ox[k] + lx[k]], pool[oy[k]:oy[k] + ly[k]], "protein")
                    assert np.array_equal(out[5].ops(k), w.ops) and (out[5].start_i[k], out[5].start_j[k]) == tuple(w.start)
            if align:  # the baseline of this workload is the full local alignment per pair, one core (the oracle's sw is serial)
                na = min(400, ns)
                t1 = time.perf_counter()
                for k in range(na):
                    orc.sw(pool[ox[k]:ox[k] + lx[k]], pool[oy[k]:oy[k] + ly[k]], "protein")
                dta = time.perf_counter() - t1
                line["cpu_baseline"] = {"value": round(na / dta, 1), "unit": "pairs/s", "cores": 1, "kind": "port",
                                        "sample": f"first {na} pairs, full localalign (fill, prefix refill, traceback) one after the other "
                                                  f"on one core, {dta:.1f} s; sampled alignments equal the GPU's"}
                return line
            line["cpu_baseline"] = {"value": round(ns / dtc, 1), "unit": "pairs/s", "cores": threads, "kind": "port",
                                    "sample": f"first {ns} pairs, one pair per thread on {threads} threads, {dtc:.1f} s; results equal the GPU's"}
    eng.close()
    return line


def lcs_pmc_traffic(m, n):
    """Measured HBM bytes per launch of the bit-vector LCS kernel, from the committed PMC summary (1 Mb pair only)."""
    f = ROOT / "profiles" / "r03_lcs1m_pmc.json"
    if not (f.exists() and m == 1_000_000):
        return None
    try:
        return int(json.loads(f.read_text())["derived"]["hbm_traffic_bytes_per_launch"])
    except (KeyError, ValueError):
        return None


def batch_pmc_traffic(workload, per_rank):
    """HBM bytes per call of the batch kernels from the committed PMC passes (profiles/r03_batch_pmc.json), scaled
    by the number of pairs (the traffic is per pair: sequences in, descriptors in, results out); null when there
    is no profile for the workload."""
    f = Path(__file__).resolve().parent / "profiles" / "r03_batch_pmc.json"
    if not f.exists():
        return None
    d = json.loads(f.read_text()).get(workload)
    if not d:
        return None
    return round(d["hbm_traffic_bytes_per_call"] * per_rank / d["pairs"])


def pmc_traffic(args, m, n, world, res):
    """HBM bytes per launch of the dominant kernel from the committed PMC passes (rocprofv3 cannot
    run inside the timed program): WRITE_SIZE + 2 x FETCH_SIZE (gfx950 correction), valid for the
    workload and schedule the profile was taken on (tools/profile_r03.sh), else null."""
    tag = {("nw", 1000000, 999747): "nw1m", ("nw", 100000, 100085): "nw100k", ("sw", 1000000, 999747): "sw1m"}.get((args.kind, m, n))
    f = Path(__file__).resolve().parent / "profiles" / f"r03_{tag}_pmc.json"
    if world != 1 or args.unrelated or tag is None or res.n_launches != 1 or not f.exists():
        return None
    return json.loads(f.read_text())["derived"]["hbm_traffic_bytes_per_launch"]


# what every rank reports about its own stripe in an N > 1 line (milliseconds per step, averaged over the timed steps)
PER_RANK_KEYS = ("rank", "fill_ms", "fill_kernel_ms", "border_wait_ms", "traceback_ms", "rows_per_lane", "overlapped", "stripe_cols")


def launch_check(args, rank, world):
    """`--launch-check`: no GPU work -- every rank joins a gloo group and reports; rank 0 prints which ranks it saw.
    What the CPU test of the launcher runs (tests/test_bench_launcher.py)."""
    import torch
    import torch.distributed as dist

    seen = [rank]
    if world > 1:
        dist.init_process_group("gloo")
        t = [torch.zeros(1, dtype=torch.int64) for _ in range(world)]
        dist.all_gather(t, torch.tensor([rank], dtype=torch.int64))
        seen = sorted(int(v.item()) for v in t)
        ranks = dist.get_world_size()
    else:
        ranks = 1
    if ranks != args.gpus:
        raise SystemExit(f"bench.py: --gpus {args.gpus} but {ranks} ranks joined")
    print(f"LAUNCH_CHECK rank {rank} of {world} pid {os.getpid()} GPU_MAX_HW_QUEUES={os.environ.get('GPU_MAX_HW_QUEUES')}", flush=True)
    if rank == 0:
        # the keys a measured N > 1 line carries per rank (tests/test_bench_launcher.py pins them)
        print(json.dumps({"launch_check": True, "n_gpus": args.gpus, "rccl_ranks": ranks, "ranks_reported": seen,
                          "backend": "gloo (launch check only; measurements use nccl = RCCL)",
                          "per_rank": [dict.fromkeys(PER_RANK_KEYS, 0) | {"rank": r} for r in seen]}), flush=True)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=None, help="ranks, one per GPU (default: WORLD_SIZE if started by torch.distributed.run, else 1)")
    ap.add_argument("--steps", type=int, default=3)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--size", type=int, default=1_000_000, help="query length (target is the mutated copy, ~same length)")
    ap.add_argument("--kind", default="nw", choices=["nw", "lcs", "sw"])
    ap.add_argument("--workload", default="pair", choices=["pair", "batch", "batch-align", "batch-global"],
                    help="pair: one long pair (headline) + a nested config-5 batch object; batch*: config 5 alone")
    ap.add_argument("--pairs", type=int, default=1_000_000)
    ap.add_argument("--cpu-sample", type=int, default=40000)
    ap.add_argument("--no-cpu", action="store_true")
    ap.add_argument("--no-batch", action="store_true", help="pair workload: leave the nested config-5 object out")
    ap.add_argument("--no-small", action="store_true", help="pair workload: leave the nested configs-1-and-2 latency object out")
    ap.add_argument("--launch-check", action="store_true", help="start the ranks, let each report over gloo, no GPU work")
    ap.add_argument("--unrelated", action="store_true",
                    help="SURVEY 8d config 3 variant: independent uniform y (seed 5678) instead of the ~90 %% identity pair")
    args = ap.parse_args()

    in_launcher = "WORLD_SIZE" in os.environ and "RANK" in os.environ
    if args.gpus is None:
        args.gpus = int(os.environ["WORLD_SIZE"]) if in_launcher else 1
    if args.gpus < 1:
        raise SystemExit("bench.py: --gpus must be >= 1")
    if args.gpus > 1 and not in_launcher:
        # nothing above imported torch or touched HIP: the ranks are a child process
        sys.exit(launch_ranks(args.gpus, sys.argv[1:]))

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus:
        raise SystemExit(f"bench.py: --gpus {args.gpus} but torch.distributed.run started {world} ranks")
    if args.launch_check:
        return launch_check(args, rank, world)

    import torch

    if world > 1:
        import torch.distributed as dist

        torch.cuda.set_device(local_rank)
        dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))
        if dist.get_world_size() != args.gpus:
            raise SystemExit(f"bench.py: --gpus {args.gpus} but the process group has {dist.get_world_size()} ranks")
    else:
        dist = None
    rccl_ranks = dist.get_world_size() if dist is not None else 1

    def finish():
        if dist is not None:
            dist.barrier()
            dist.destroy_process_group()

    import sequencealigner_amd as sa
    from sequencealigner_amd.synth import dna_pair

    if args.workload in ("batch", "batch-align", "batch-global"):
        line = bench_batch(args, rank, local_rank, world, dist, torch)
        if rank == 0:
            line["rccl_ranks"] = rccl_ranks
            print(json.dumps(line), flush=True)
        return finish()

    x, y = dna_pair(args.size, seed=5678, related=False) if args.unrelated else dna_pair(args.size, seed=1234)
    m, n = len(x), len(y)
    cells = m * n

    eng = None
    if world > 1:
        from sequencealigner_amd.stripes import StripedAligner

        runner = StripedAligner(local_rank, rank, world, dist)
        run = lambda: runner.align(args.kind, x, y, "dna")  # noqa: E731
    else:
        eng = sa.Engine(local_rank)
        if args.kind == "lcs":
            run = lambda: eng.lcs(x, y)  # noqa: E731
        else:
            run = lambda: getattr(eng, args.kind)(x, y, "dna")  # noqa: E731

    def sync():
        torch.cuda.synchronize()
        if dist is not None:
            dist.barrier()
            torch.cuda.synchronize()

    schedule_note = None
    for w in range(max(args.warmup, 1 if world > 1 else 0)):
        try:
            res = run()
        except RuntimeError as e:
            # Column stripes with the borders crossing under the running persistent kernel need the
            # transfer backend to make progress beside it; if a bounded wait expired instead, every
            # rank gets a StripeError and all fall back to the launch-per-diagonal schedule together.
            if world > 1 and runner.overlap and type(e).__name__ == "StripeError":
                runner.overlap = False
                schedule_note = f"overlapped schedule failed in warmup ({e}); fell back"
                sync()
                res = run()
            else:
                raise
    def timed_steps():
        sync()
        t0 = time.perf_counter()
        acc = {"kernel_ms": 0.0, "tb_ms": 0.0, "dev_ms": 0.0, "fill_ms": 0.0, "wait_ms": 0.0, "launches": 0, "cells_trace": 0, "cells_notrace": 0}
        res = None
        for _ in range(args.steps):
            res = run()
            acc["dev_ms"] += res.fill_ms + res.traceback_ms
            acc["fill_ms"] += res.fill_ms
            acc["kernel_ms"] += res.kernel_fill_ms
            acc["tb_ms"] += res.traceback_ms
            acc["wait_ms"] += res.border_wait_ms
            acc["launches"] += res.n_launches
            acc["cells_trace"] += res.cells_trace
            acc["cells_notrace"] += res.cells_notrace
        sync()
        return res, acc, time.perf_counter() - t0

    try:
        res, acc, dt = timed_steps()
    except RuntimeError as e:
        # The same failure inside the timed loop (every rank raises StripeError together, stripes.py): finish on the
        # in-line schedule and say so, so that the line is always printed and can be read.
        if world > 1 and runner.overlap and type(e).__name__ == "StripeError":
            runner.overlap = False
            schedule_note = f"overlapped schedule failed in a timed step ({e}); all steps re-timed on the in-line schedule"
            res, acc, dt = timed_steps()
        else:
            raise
    kernel_ms, tb_ms, dev_ms, launches = acc["kernel_ms"], acc["tb_ms"], acc["dev_ms"], acc["launches"]
    cells_trace, cells_notrace = acc["cells_trace"], acc["cells_notrace"]
    per_rank = None
    if dist is not None:
        tt = torch.tensor([dt], dtype=torch.float64, device="cuda")
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        dt = float(tt.item())
        # per-rank kernel time and cell counts: the roofline is per GPU, so sum the work and take the longest kernel
        kk = torch.tensor([kernel_ms], dtype=torch.float64, device="cuda")
        dist.all_reduce(kk, op=dist.ReduceOp.MAX)
        kernel_ms = float(kk.item())
        cc = torch.tensor([float(cells_trace), float(cells_notrace)], dtype=torch.float64, device="cuda")
        dist.all_reduce(cc, op=dist.ReduceOp.SUM)
        cells_trace, cells_notrace = int(cc[0].item()), int(cc[1].item())
        # what every rank saw of its own stripe: a bad curve can be read off the record (who waited for whom)
        stripe = res.extra.get("stripe", (0, 0)) if isinstance(res.extra, dict) else (0, 0)
        mine = torch.tensor([float(rank), acc["fill_ms"] / args.steps, acc["kernel_ms"] / args.steps, acc["wait_ms"] / args.steps,
                             acc["tb_ms"] / args.steps, float(res.rows_per_lane), 1.0 if res.extra.get("overlapped") else 0.0,
                             float(stripe[1] - stripe[0])], dtype=torch.float64, device="cuda")
        allr = [torch.zeros_like(mine) for _ in range(world)]
        dist.all_gather(allr, mine)
        per_rank = [{k: (int(v) if k in ("rank", "rows_per_lane", "stripe_cols") else (bool(v) if k == "overlapped" else round(float(v), 3)))
                     for k, v in zip(PER_RANK_KEYS, t.tolist())} for t in allr]

    line = None
    if rank == 0:
        gcups = cells * args.steps / dt / 1e9
        k_s = kernel_ms / 1e3
        ops = OPS_PER_CELL[args.kind]
        ops_nt = ops - OPS_CODE[args.kind]
        # every fill launch is inside kernel_ms (a checkpointed run: the code-free checkpoint pass + the bands
        # recomputed during the traceback) and every cell is counted at the work its launch does
        lane_ops = cells_trace * ops + cells_notrace * ops_nt
        achieved = lane_ops / k_s / 1e12 / world  # per GPU
        hbm_bytes = cells_trace * HBM_BYTES_PER_CELL
        line = {
            "metric": "GCUPS NW fill+traceback (cell updates/s)" if args.kind == "nw" else f"GCUPS {args.kind} fill+traceback",
            "value": round(gcups, 2),
            "unit": "GCUPS",
            "n_gpus": world,
            "rccl_ranks": rccl_ranks,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": round(dt / args.steps * 1e3, 3),
            "higher_is_better": True,
            "scaling": "strong",
            "vs_baseline": None,
            "dtype": "int32",
            "data": "synthetic",
            "config": {
                "workload": f"{args.kind.upper()} {m}x{n} synthetic DNA ({'unrelated, seed 5678' if args.unrelated else '~90% identity, seed 1234'}), EDNAFULL, gaps -5/-1, "
                            f"score + full traceback, bit-exact op list",
                "rows_per_lane": res.rows_per_lane,
                "tile_steps": res.tile_steps,
                "parallelism": "single GPU" if world == 1 else f"column stripes x{world}",
                "schedule": ("persistent strip pipeline" if res.n_launches == 1 else
                             f"persistent strip pipeline, checkpointed: codes recomputed in {res.n_launches - 1} bands during the traceback"
                             if res.n_launches < 100 else "one launch per tile diagonal"),
            },
            "result": {"score": int(res.score), "n_ops": int(len(res.ops))},
            # the C-ABI takes host buffers (2 MB of residues in, 1 MB of moves out): `value` includes them and the
            # host-side encoding; this is the same work timed on the device stream only (HIP events, first H2D
            # enqueued -> last D2H done), i.e. what remains once the inputs are resident
            "gcups_device_timeline": round(cells * args.steps / (dev_ms / 1e3) / 1e9, 2) if world == 1 and dev_ms > 0 else None,
            "breakdown_ms_per_step": {
                "fill_kernels": round(kernel_ms / args.steps, 3),
                "traceback_plus_d2h": round(tb_ms / args.steps, 3),
                "fill_kernel_launches": launches // args.steps,
            },
            "roofline": {
                "bound": "valu",
                "achieved": round(achieved, 3),
                "peak": VALU_PEAK_TOPS,
                "unit": "T int32 lane-ops/s",
                "frac": round(achieved / VALU_PEAK_TOPS, 4),
                "frac_of_mix_weighted_peak": round(achieved / VALU_PEAK_MIX_TOPS[args.kind], 4),
                "peak_note": "peak = one wave64 int32 instruction per 4 cycles per SIMD at 2.4 GHz (SURVEY 8d); the probe in profiles/r02_probe_valu_asm.txt "
                             "measures 36.9 T for max / max3 / SDWA / DPP, 52 T for v_add_u32 with >= 2 waves per SIMD: mix-weighted peak for this recurrence "
                             f"{VALU_PEAK_MIX_TOPS[args.kind]} T",
                "traffic": pmc_traffic(args, m, n, world, res),
                "traffic_note": "HBM bytes per launch, WRITE_SIZE + 2 x FETCH_SIZE of the PMC passes under profiles/; "
                                "algorithmic: 0.25 B/cell trace + 16 B per strip column of bottom-row hand-off",
                "kernel": "fill_persistent" if res.n_launches < 100 else "fill_tile",
                "ops_per_cell": ops,
                "ops_per_cell_code_free_pass": ops_nt,
                "cells_per_step": {"with_codes": cells_trace // args.steps, "code_free": cells_notrace // args.steps},
                "avg_launch_us": round(kernel_ms * 1e3 / max(launches, 1), 2),
                "hbm": {
                    "achieved": round(hbm_bytes / k_s / 1e9 / world, 1),
                    "peak": HBM_PEAK_GBS,
                    "unit": "GB/s",
                    "frac": round(hbm_bytes / k_s / 1e9 / world / HBM_PEAK_GBS, 4),
                    "bytes_per_cell": HBM_BYTES_PER_CELL,
                },
            },
        }
        if args.kind == "lcs" and res.rows_per_lane == 32 and world == 1:
            # bit-vector LCS (sa_lcsbits.hip): 3 integer operations per 32 cells, one bit of traceback record per cell;
            # what it moves is the record, S x steps x 256 bytes per launch (m n / 8 + padding)
            strips, steps_per_strip = (m + 2047) // 2048, ((n + 126 + 63) // 64) * 64
            # (a checkpointed run: the first pass writes no record, the bands write the record of the rows they cover; all
            # launches are inside kernel_ms, so `achieved` is the record written by all of them over the time of all of them)
            rec_step = int(strips * steps_per_strip * 256 * (cells_trace / args.steps / (m * n)))
            rec_bytes = rec_step // max(launches // args.steps, 1)
            gbs = rec_step * args.steps / k_s / 1e9
            line["roofline"] = {
                "bound": "hbm", "achieved": round(gbs, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": round(gbs / HBM_PEAK_GBS, 4),
                "traffic": lcs_pmc_traffic(m, n), "kernel": "lcs_bits_persistent", "bytes_per_cell": 0.125, "bytes_per_launch": rec_bytes,
                "traffic_note": "HBM bytes per launch, WRITE_SIZE + 2 x FETCH_SIZE of the PMC passes in profiles/r03_lcs1m_pmc.json (that workload only)",
                "avg_launch_us": round(kernel_ms * 1e3 / max(launches, 1), 2),
                "ns_per_column_step": round(kernel_ms * 1e6 / max(launches, 1) / steps_per_strip, 1),
                "note": "the only HBM traffic of any size is the write-once traceback record (1 bit per cell).  The kernel does not reach the "
                        "HBM roof: a strip is ONE wave sweeping n columns in sequence, about 11 instructions per column step, and a lone wave "
                        "issues one instruction per 4+ cycles (DESIGN 3.1b); strips beyond the first add only the pipeline lag.",
            }
        if per_rank is not None:
            line["per_rank"] = per_rank
            line["per_rank_note"] = ("ms per step on each rank: fill_ms = H2D + profile + stripe fill, fill_kernel_ms = the stripe's persistent launch "
                                     "(includes the time it waits for its left neighbour), border_wait_ms = of that, what an average wave spent "
                                     "polling for rows of the left border, traceback_ms = this stripe's walk + D2H; overlapped = borders crossed "
                                     "under the running kernel (false: in-line schedule)")
        if schedule_note:
            line["config"]["schedule_note"] = schedule_note
        if not args.no_cpu and world == 1:
            cb, _ = cpu_baseline(x, y, args.kind, min(args.cpu_sample, m, n))
            rb = reference_baseline(x, y, args.kind, min(args.cpu_sample // 2, m, n))
            if rb is not None:  # the reference itself when its binary is here, the port beside it
                rb["port"] = {"value": cb["value"], "sample": cb["sample"]}
                cb = rb
            line["cpu_baseline"] = cb

    if world == 1 and args.workload == "pair" and not args.no_small and rank == 0:
        # configs 1 and 2 of BASELINE.json: the latency cases, warm call / cold CLI process / the reference beside them
        try:
            line["small_pairs"] = bench_small_pairs(eng, args.no_cpu)
        except Exception as e:  # noqa: BLE001 -- never lose the headline line to the side measurement
            line["small_pairs"] = {"error": repr(e)}

    if not args.no_batch:
        # the other half of BASELINE.json's metric, same run: config 5 sharded over the same ranks.  The pair's
        # workspaces (250 GB of codes at 1 Mb x 1 Mb) go first.
        if eng is not None:
            eng.close()
        else:
            runner.engine.eng.release_workspace()
        sync()
        bl = bench_batch(args, rank, local_rank, world, dist, torch, workload="batch", nested=True)
        if rank == 0:
            line["batch"] = {k: bl[k] for k in ("metric", "value", "unit", "n_gpus", "steps", "ms_per_step", "scaling", "config", "value_note",
                                                "value_host_pool", "value_host_pool_note", "gcups_end_to_end", "gcups_kernel_only",
                                                "roofline", "cpu_baseline") if k in bl}
    if rank == 0:
        print(json.dumps(line), flush=True)
    finish()


if __name__ == "__main__":
    main()
