"""Dev tool: sweep (K, C) on synthetic DNA pairs and print GCUPS. Run on the GPU box."""
import sys, time
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
import numpy as np
import sequencealigner_amd as sa
from sequencealigner_amd.synth import dna_pair

def main():
    sizes = [int(a) for a in sys.argv[1].split(",")] if len(sys.argv) > 1 else [20000, 100000]
    kinds = sys.argv[2].split(",") if len(sys.argv) > 2 else ["nw"]
    combos = [(0, 0)] + [(K, C) for K in (1, 2, 4, 8) for C in (64, 128, 256, 512)]
    if len(sys.argv) > 3:
        combos = [tuple(int(v) for v in c.split(":")) for c in sys.argv[3].split(",")]
    eng = sa.Engine(0)
    for n in sizes:
        x, y = dna_pair(n)
        cells = len(x) * len(y)
        for kind in kinds:
            for (K, C) in combos:
                eng.set_tuning(K, C)
                best = None
                for rep in range(2):
                    t0 = time.perf_counter()
                    r = eng.lcs(x, y) if kind == "lcs" else getattr(eng, kind)(x, y, "dna")
                    wall = (time.perf_counter() - t0) * 1e3
                    if best is None or r.kernel_fill_ms < best[0].kernel_fill_ms:
                        best = (r, wall)
                r, wall = best
                print(f"{kind} n={n} K={r.rows_per_lane} C={r.tile_steps} launches={r.n_launches} "
                      f"fillk={r.kernel_fill_ms:.2f}ms fill={r.fill_ms:.2f}ms tb={r.traceback_ms:.2f}ms wall={wall:.1f}ms "
                      f"GCUPS(kernel)={cells / r.kernel_fill_ms / 1e6:.1f} GCUPS(fill+tb)={cells / (r.fill_ms + r.traceback_ms) / 1e6:.1f} "
                      f"score={r.score} nops={len(r.ops)}" + (" [auto]" if K == 0 else ""), flush=True)
    eng.close()

main()
