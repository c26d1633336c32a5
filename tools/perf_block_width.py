"""One GPU: does a column block get cheaper when it gets narrower?  (cyclic column blocks)
Persistent fill of m x (n / V) for V = 8, 16, 32, 64 virtual stripes, K = 2 and 1, with fill_persistent and fill_group.
A block is a chain of m / (64 K) strips; while fewer strips are in flight than the GPU has waves the chain, not the
width, sets the time."""
import sys
from pathlib import Path

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
import sequencealigner_amd as sa
from sequencealigner_amd.synth import dna_pair

m = int(sys.argv[1]) if len(sys.argv) > 1 else 1_000_000
eng = sa.Engine(0)
x, y = dna_pair(m, seed=1234)
for K in (2, 1, 4):
    eng.set_tuning(K, 0)
    for group in (0, 1):
        eng.set_option("group", group)
        row = []
        for V in (8, 16, 32, 64):
            w = len(y) // V
            t = min(eng.nw(x, y[:w], "dna", flags=2).kernel_fill_ms for _ in range(2))  # score only: the block's fill
            row.append(f"V={V} (w={w}): {t:.1f} ms")
        print(f"K={K} group={group}  " + "  ".join(row), flush=True)
eng.close()
