"""Dev tool: fill_group (sa_group.h) against fill_persistent: same results, fill times.
usage: perf_group.py sizes kinds K:wps,K:wps,...   (K = 0: the library's choice)"""
import sys
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
import numpy as np
import sequencealigner_amd as sa
from sequencealigner_amd.synth import dna_pair

sizes = [int(a) for a in sys.argv[1].split(",")] if len(sys.argv) > 1 else [100000]
kinds = sys.argv[2].split(",") if len(sys.argv) > 2 else ["nw"]
combos = [tuple(int(v) for v in c.split(":")) for c in sys.argv[3].split(",")] if len(sys.argv) > 3 else [(0, 1), (2, 1), (4, 1)]
reps = int(sys.argv[4]) if len(sys.argv) > 4 else 3
eng = sa.Engine(0)
for n in sizes:
    x, y = dna_pair(n)
    cells = len(x) * len(y)
    for kind in kinds:
        for (K, wps) in combos:
            res = {}
            for group in (0, 1):
                if group == 0 and wps > 1:
                    continue
                eng.set_option("group", group); eng.set_tuning(K, 0); eng.set_option("waves_per_simd", wps)
                best = None
                for rep in range(reps):
                    r = getattr(eng, kind)(x, y, "dna")
                    if best is None or r.kernel_fill_ms < best.kernel_fill_ms: best = r
                res[group] = best
                print(f"{kind} n={n} group={group} K={best.rows_per_lane} wps={wps} fillk={best.kernel_fill_ms:.3f}ms tb={best.traceback_ms:.2f}ms "
                      f"GCUPS(kernel)={cells/best.kernel_fill_ms/1e6:.1f} score={best.score}", flush=True)
            if 0 in res and 1 in res:
                a, b = res[0], res[1]
                same = a.score == b.score and a.end == b.end and a.start == b.start and np.array_equal(a.ops, b.ops)
                print(f"   same result: {same}", flush=True)
            elif 1 in res and "ref" in globals():
                pass
eng.close()
