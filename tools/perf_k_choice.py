"""fill time per rows-per-lane K at several sizes, and what the engine picks by itself"""
import sys
from pathlib import Path

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
import sequencealigner_amd as sa
from sequencealigner_amd.synth import dna_pair

eng = sa.Engine(0)
x, y = dna_pair(1_000_000, seed=1234)
for m, n in ((20_000, 20_000), (100_000, 100_000), (300_000, 300_000), (786_432, 999_000), (600_000, 1_000_000), (1_000_000, 999_747)):
    row = []
    for K in (1, 2, 4, 8):
        eng.set_tuning(K, 0)
        row.append(min(eng.nw(x[:m], y[:n], "dna").kernel_fill_ms for _ in range(2)))
    eng.set_tuning(0, 0)
    r = eng.nw(x[:m], y[:n], "dna")
    print(f"{m}x{n}: K=1 {row[0]:.2f}  K=2 {row[1]:.2f}  K=4 {row[2]:.2f}  K=8 {row[3]:.2f} ms; auto K={r.rows_per_lane} {r.kernel_fill_ms:.2f} ms", flush=True)
