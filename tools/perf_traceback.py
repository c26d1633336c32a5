"""traceback time vs gap density: identical pair (one run per 64-row window) / 90 % / unrelated"""
import sys
from pathlib import Path

import numpy as np

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
import sequencealigner_amd as sa
from sequencealigner_amd.synth import dna_pair

eng = sa.Engine(0)
n = int(sys.argv[1]) if len(sys.argv) > 1 else 400_000
x, y = dna_pair(n, seed=1234)
xu, yu = dna_pair(n, seed=5678, related=False)
for name, a, b in (("identical", x, x), ("90% identity", x, y), ("unrelated", xu, yu)):
    r = min((eng.nw(a, b, "dna") for _ in range(2)), key=lambda r: r.traceback_ms)
    ops = r.ops
    runs = int((np.diff(ops) != 0).sum()) + 1
    print(f"{name}: {len(ops)} moves, {runs} runs, {int((ops != 0).sum())} gap moves, traceback {r.traceback_ms:.2f} ms "
          f"= {r.traceback_ms * 1e3 / (len(ops) / 64):.2f} us per 64 moves, {r.traceback_ms * 1e6 / runs:.0f} ns per run", flush=True)
