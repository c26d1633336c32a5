"""One GPU: the two ingredients of the column-stripe critical path, per rows-per-lane K.
T_own   = persistent fill of m x n_g (a stripe whose left border is already there)
T_first = persistent fill of R x n_g (how long a rank takes to emit its first border chunk)
finish of the last of G stripes ~ (G-1) * T_first + T_own."""
import sys
from pathlib import Path

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
import sequencealigner_amd as sa
from sequencealigner_amd.synth import dna_pair

m = int(sys.argv[1]) if len(sys.argv) > 1 else 1_000_000
R = 8192
eng = sa.Engine(0)
x, y = dna_pair(m, seed=1234)
for G in (2, 4, 8):
    ng = len(y) // G
    for K in (1, 2, 4, 8):
        eng.set_tuning(K, 0)
        own = min(eng.nw(x, y[:ng], "dna").kernel_fill_ms for _ in range(2))
        first = min(eng.nw(x[:R], y[:ng], "dna").kernel_fill_ms for _ in range(2))
        print(f"G={G} n_g={ng} K={K}: T_own={own:.1f} ms  T_first={first:.2f} ms  -> last stripe done at ~{(G-1)*first+own:.1f} ms", flush=True)
    eng.release_workspace() if hasattr(eng, "release_workspace") else None
