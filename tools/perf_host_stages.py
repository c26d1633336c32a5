"""Dev tool: where the time of the headline call goes outside the kernels (run with SA_TIMING=1)."""
import sys
import time
from pathlib import Path

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
import sequencealigner_amd as sa
from sequencealigner_amd.synth import dna_pair
eng = sa.Engine(0)
x, y = dna_pair(1_000_000, seed=1234)
for rep in range(3):
    t0 = time.perf_counter()
    r = eng.nw(x, y, "dna")
    t1 = time.perf_counter()
    print(f"python wall {1e3*(t1-t0):.2f} ms; fill_ms {r.fill_ms:.2f} kernel {r.kernel_fill_ms:.2f} tb {r.traceback_ms:.2f}", flush=True)
