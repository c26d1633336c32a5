"""torchrun --nproc-per-node W tools/perf_stripes_1gpu.py SIZE: W processes share GPU 0 (gloo
transport through the host): exercises both stripe schedules at size; NOT a scaling measurement."""
import sys
import time
from pathlib import Path

ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))
sys.path.insert(0, str(ROOT / "tests"))
import torch
import torch.distributed as dist

torch.cuda.init()
from helpers.run_stripes_gpu_gloo import GlooForCuda  # noqa: E402
from sequencealigner_amd.stripes import GpuStripeEngine, StripedAligner  # noqa: E402
from sequencealigner_amd.synth import dna_pair  # noqa: E402

dist.init_process_group("gloo")
rank, world = dist.get_rank(), dist.get_world_size()
size = int(sys.argv[1]) if len(sys.argv) > 1 else 200_000
al = StripedAligner(0, rank, world, GlooForCuda(), engine=GpuStripeEngine(0))
x, y = dna_pair(size, seed=5)
for overlap in (True, False, True):
    al.overlap = overlap
    dist.barrier()
    t0 = time.perf_counter()
    r = al.align("nw", x, y, "dna")
    torch.cuda.synchronize()
    dist.barrier()
    dt = time.perf_counter() - t0
    print(f"rank {rank} overlap={overlap} {dt*1e3:.1f} ms score={r.score} launches={r.n_launches} K={r.rows_per_lane} "
          f"fill={r.fill_ms:.1f} tb={r.traceback_ms:.1f}", flush=True)
dist.barrier()
dist.destroy_process_group()
