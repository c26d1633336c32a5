"""Dev tool: dump the first window of the bit-vector walker (debug build) next to a host emulation."""
import sys
from pathlib import Path

import numpy as np

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
import sequencealigner_amd as sa

eng = sa.Engine(0)
rng = np.random.default_rng(1)
m, n = 64, 300
for (mm, nn) in [(100, 80), (200, 80), (64, 81)]:
    rng.integers(0, 4, mm); rng.integers(0, 4, nn)
x = rng.integers(0, 4, m).astype(np.uint8) + 65
y = rng.integers(0, 4, n).astype(np.uint8) + 65
print("x", bytes(x).decode())
print("y", bytes(y).decode())
g = eng.lcs(x, y)
print(g.ops[:10])
