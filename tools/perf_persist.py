"""Dev tool: persistent vs launch-based fill, sweep K and waves/SIMD."""
import sys, time
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
import numpy as np
import sequencealigner_amd as sa
from sequencealigner_amd.synth import dna_pair

sizes = [int(a) for a in sys.argv[1].split(",")] if len(sys.argv) > 1 else [100000]
kinds = sys.argv[2].split(",") if len(sys.argv) > 2 else ["nw"]
combos = [tuple(int(v) for v in c.split(":")) for c in sys.argv[3].split(",")] if len(sys.argv) > 3 else [(0, 2), (8, 1), (8, 2), (4, 1), (4, 2), (2, 2)]
eng = sa.Engine(0)
for n in sizes:
    x, y = dna_pair(n)
    cells = len(x) * len(y)
    for kind in kinds:
        eng.set_option("persistent", 0); eng.set_tuning(0, 0)
        ref = eng.lcs(x, y) if kind == "lcs" else getattr(eng, kind)(x, y, "dna")
        print(f"{kind} n={n} launch-path K={ref.rows_per_lane} C={ref.tile_steps} fillk={ref.kernel_fill_ms:.2f}ms tb={ref.traceback_ms:.2f}ms GCUPS(kernel)={cells/ref.kernel_fill_ms/1e6:.1f}", flush=True)
        eng.set_option("persistent", 1)
        for (K, wps) in combos:
            eng.set_tuning(K, 0); eng.set_option("waves_per_simd", wps)
            best = None
            for rep in range(2):
                r = eng.lcs(x, y) if kind == "lcs" else getattr(eng, kind)(x, y, "dna")
                ok = r.score == ref.score and np.array_equal(r.ops, ref.ops)
                if best is None or r.kernel_fill_ms < best.kernel_fill_ms: best = r
            print(f"{kind} n={n} persistent K={best.rows_per_lane} wps={wps} fillk={best.kernel_fill_ms:.2f}ms fill={best.fill_ms:.2f} tb={best.traceback_ms:.2f}ms GCUPS(kernel)={cells/best.kernel_fill_ms/1e6:.1f} GCUPS(total)={cells/(best.fill_ms+best.traceback_ms)/1e6:.1f} same_as_launch_path={ok}", flush=True)
eng.close()
