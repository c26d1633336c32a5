"""Dev tool: bit-vector LCS vs the oracle on a few shapes; prints the first differing move."""
import sys
from pathlib import Path

import numpy as np

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
import sequencealigner_amd as sa
from oracle import Oracle

eng, orc = sa.Engine(0), Oracle()
rng = np.random.default_rng(1)
for (m, n) in [(100, 80), (200, 80), (64, 81), (64, 300), (65, 300), (100, 257), (128, 200), (3000, 80), (5000, 3000)]:
    x = rng.integers(0, 4, m).astype(np.uint8) + 65
    y = rng.integers(0, 4, n).astype(np.uint8) + 65
    g, o = eng.lcs(x, y), orc.lcs(x, y)
    same = len(g.ops) == len(o.ops) and np.array_equal(g.ops, o.ops)
    k = -1
    if not same:
        L = min(len(g.ops), len(o.ops))
        d = np.nonzero(g.ops[:L] != o.ops[:L])[0]
        k = int(d[0]) if len(d) else L
    print(m, n, "K", g.rows_per_lane, "score", g.score, o.score, "ops", len(g.ops), len(o.ops), "same" if same else f"first diff at move {k}: got {g.ops[max(0,k-3):k+5]} want {o.ops[max(0,k-3):k+5]}", flush=True)
