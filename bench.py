#!/usr/bin/env python3
"""bench.py -- headline benchmark of the DP alignment hot path (BASELINE.json metric).

One "step" = one complete Needleman-Wunsch alignment (H2D of the two sequences, query
profile, anti-diagonal fill with 2-bit traceback codes, traceback walk, D2H of the op list)
of the synthetic ~90 %-identity DNA pair of SURVEY.md 8d (config 4: 1 Mb x 1 Mb; the timed
window mirrors the reference's own time_ms window, main.cpp:736 -> :820).

  python bench.py [--gpus N] [--steps K] [--warmup W] [--size L] [--kind nw|lcs|sw]

N = 1 runs the whole matrix on one MI355X.  N > 1 (launched by torch.distributed.run, one
rank per GPU) column-stripes the same pair over the ranks; see DESIGN.md "Multi-GPU".

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

# algorithmic work per cell: int32 VALU lane-ops and HBM bytes
OPS_PER_CELL = {"nw": 12, "sw": 14, "lcs": 6}
HBM_BYTES_PER_CELL = 0.25
VALU_PEAK_TOPS = 39.3  # 256 CU x 64 lanes x 2.4 GHz int32 lane-ops/s (measured: profiles/probe_valu_r01.txt)
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


def bench_batch(args, rank, local_rank, world, dist, torch):
    """Config 5: independent protein pairs (300-1000 aa, BLOSUM62, -5/-1), SW score + end cell
    (--workload batch) or full local alignments (--workload batch-align: + direction codes, per-pair
    traceback, op lists back on the host).  Pairs are sharded contiguously over the ranks; no
    data-path collective."""
    import sequencealigner_amd as sa
    from sequencealigner_amd.synth import protein_pairs

    per_rank = args.pairs // world
    pool, ox, lx, oy, ly = protein_pairs(per_rank, seed=42 + rank)
    cells = int((lx.astype(np.int64) * ly.astype(np.int64)).sum())
    eng = sa.Engine(local_rank)
    align = args.workload in ("batch-align", "batch-global")
    glob = args.workload == "batch-global"
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
    if rank == 0:
        line = {
            "metric": ("batched NW pairs/s (full global alignments)" if glob else "batched SW pairs/s (full local alignments)") if align else "batched SW pairs/s (score + end cell)", "value": round(per_rank * world * args.steps / dt, 1),
            "unit": "pairs/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": round(dt / args.steps * 1e3, 3), "higher_is_better": True, "scaling": "strong",
            "vs_baseline": None, "dtype": "int32", "data": "synthetic",
            "config": {"workload": f"{per_rank * world} protein pairs, lengths U[300,1000], uniform residues, BLOSUM62 -5/-1, "
                                   f"sequence pool resident in HBM, descriptors in / results out inside the timed region" + (", op lists back on the host" if align else ""), "parallelism": f"pair sharding x{world}"},
            "value_host_pool": round(per_rank * world * args.steps / dt_host, 1),
            "value_host_pool_note": "same calls with the pool as a pageable host buffer (H2D of the sequences inside every call)",
            "gcups_end_to_end": round(cells_all * args.steps / dt / 1e9, 1),
            "gcups_kernel_only": round(cells_all * args.steps / (kms / 1e3) / 1e9, 1),
            "roofline": {"bound": "valu", "achieved": round(cells_all * args.steps * ((12 if glob else 14) if align else 10) / (kms / 1e3) / 1e12 / world, 3),
                         "peak": VALU_PEAK_TOPS, "unit": "T int32 lane-ops/s",
                         "frac": round(cells_all * args.steps * ((12 if glob else 14) if align else 10) / (kms / 1e3) / 1e12 / world / VALU_PEAK_TOPS, 4),
                         "traffic": None, "kernel": ("nw_batch_kernel<K>" if glob else "sw_batch_kernel<K,TRACE>") if align else "sw_batch_kernel",
                         "ops_per_cell": (12 if glob else 14) if align else 10},
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
                print(json.dumps(line), flush=True)
                return
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
                print(json.dumps(line), flush=True)
                return
            line["cpu_baseline"] = {"value": round(ns / dtc, 1), "unit": "pairs/s", "cores": threads, "kind": "port",
                                    "sample": f"first {ns} pairs, one pair per thread on {threads} threads, {dtc:.1f} s; results equal the GPU's"}
        print(json.dumps(line), flush=True)
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()


def pmc_traffic(args, m, n, world, res):
    """HBM bytes per launch of the dominant kernel from the committed PMC passes (rocprofv3 cannot
    run inside the timed program): WRITE_SIZE + 2 x FETCH_SIZE (gfx950 correction), valid for the
    workload and schedule the profile was taken on, else null."""
    f = Path(__file__).resolve().parent / "profiles" / "r01_nw1m_pmc_v5_persistent.json"
    if world != 1 or args.kind != "nw" or (m, n) != (1000000, 999747) or res.n_launches != 1 or not f.exists():
        return None
    return json.loads(f.read_text())["derived"]["hbm_traffic_bytes_per_launch"]


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=3)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--size", type=int, default=1_000_000, help="query length (target is the mutated copy, ~same length)")
    ap.add_argument("--kind", default="nw", choices=["nw", "lcs", "sw"])
    ap.add_argument("--workload", default="pair", choices=["pair", "batch", "batch-align", "batch-global"],
                    help="pair: one long pair (headline); batch: config 5, many protein pairs, SW score + end cell")
    ap.add_argument("--pairs", type=int, default=1_000_000)
    ap.add_argument("--cpu-sample", type=int, default=40000)
    ap.add_argument("--no-cpu", action="store_true")
    ap.add_argument("--unrelated", action="store_true",
                    help="SURVEY 8d config 3 variant: independent uniform y (seed 5678) instead of the ~90 %% identity pair")
    args = ap.parse_args()

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))

    import torch

    if world > 1:
        import torch.distributed as dist

        torch.cuda.set_device(local_rank)
        dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))
    else:
        dist = None

    import sequencealigner_amd as sa
    from sequencealigner_amd.synth import dna_pair

    if args.workload in ("batch", "batch-align", "batch-global"):
        return bench_batch(args, rank, local_rank, world, dist, torch)

    x, y = dna_pair(args.size, seed=5678, related=False) if args.unrelated else dna_pair(args.size, seed=1234)
    m, n = len(x), len(y)
    cells = m * n

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
    sync()
    t0 = time.perf_counter()
    kernel_ms = 0.0
    tb_ms = 0.0
    dev_ms = 0.0
    launches = 0
    for _ in range(args.steps):
        res = run()
        dev_ms += res.fill_ms + res.traceback_ms
        kernel_ms += res.kernel_fill_ms
        tb_ms += res.traceback_ms
        launches += res.n_launches
    sync()
    dt = time.perf_counter() - t0
    if dist is not None:
        tt = torch.tensor([dt], dtype=torch.float64, device="cuda")
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        dt = float(tt.item())

    if rank == 0:
        gcups = cells * args.steps / dt / 1e9
        k_s = kernel_ms / 1e3
        ops = OPS_PER_CELL[args.kind]
        achieved = cells * args.steps * ops / k_s / 1e12 / world  # per GPU
        line = {
            "metric": "GCUPS NW fill+traceback (cell updates/s)" if args.kind == "nw" else f"GCUPS {args.kind} fill+traceback",
            "value": round(gcups, 2),
            "unit": "GCUPS",
            "n_gpus": world,
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
                "traffic": pmc_traffic(args, m, n, world, res),
                "traffic_note": "HBM bytes per launch, WRITE_SIZE + 2 x FETCH_SIZE of profiles/r01_nw1m_pmc_v5_persistent.json; "
                                "algorithmic: 0.25 B/cell trace + 16 B per strip column of bottom-row hand-off",
                "kernel": "fill_persistent" if res.n_launches < 100 else "fill_tile",
                "ops_per_cell": ops,
                "avg_launch_us": round(kernel_ms * 1e3 / max(launches, 1), 2),
                "hbm": {
                    "achieved": round(cells * args.steps * HBM_BYTES_PER_CELL / k_s / 1e9 / world, 1),
                    "peak": HBM_PEAK_GBS,
                    "unit": "GB/s",
                    "frac": round(cells * args.steps * HBM_BYTES_PER_CELL / k_s / 1e9 / world / HBM_PEAK_GBS, 4),
                    "bytes_per_cell": HBM_BYTES_PER_CELL,
                },
            },
        }
        if schedule_note:
            line["config"]["schedule_note"] = schedule_note
        if not args.no_cpu and world == 1:
            cb, _ = cpu_baseline(x, y, args.kind, min(args.cpu_sample, m, n))
            rb = reference_baseline(x, y, args.kind, min(args.cpu_sample // 2, m, n))
            if rb is not None:  # the reference itself when its binary is here, the port beside it
                rb["port"] = {"value": cb["value"], "sample": cb["sample"]}
                cb = rb
            line["cpu_baseline"] = cb
        print(json.dumps(line), flush=True)
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
