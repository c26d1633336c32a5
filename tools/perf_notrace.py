"""fill time with and without code stores (SA_FLAG_NO_TRACEBACK = 2), per K, at 1 Mb and 1.5 Mb rows"""
import sys
from pathlib import Path

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
import sequencealigner_amd as sa
from sequencealigner_amd.synth import dna_pair

eng = sa.Engine(0)
x, y = dna_pair(1_000_000, seed=1234)
for K in (8, 4):
    eng.set_tuning(K, 0)
    a = min(eng.nw(x, y, "dna").kernel_fill_ms for _ in range(2))
    b = min(eng.nw(x, y, "dna", flags=2).kernel_fill_ms for _ in range(2))
    print(f"1M x 1M K={K}: fill with codes {a:.1f} ms, score only {b:.1f} ms", flush=True)
# quantisation: 1024 waves, strips of 512 rows: 1 048 576 rows = exactly 2 rounds of 1024 strips
eng.set_tuning(8, 0)
for m in (1_048_576, 524_288, 786_432):
    b = min(eng.nw(x[:m] if m <= len(x) else x, y, "dna", flags=2).kernel_fill_ms for _ in range(2))
    mm = min(m, len(x))
    print(f"{mm} x {len(y)} K=8 score only: {b:.1f} ms = {mm*len(y)/b/1e6:.0f} GCUPS", flush=True)
