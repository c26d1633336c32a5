"""Dev tool: traceback time at 1 Mb for the related pair (NW, LCS) and the unrelated pair (NW)."""
import sys
from pathlib import Path

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
import sequencealigner_amd as sa
from sequencealigner_amd.synth import dna_pair

eng = sa.Engine(0)
x, y = dna_pair(1_000_000, seed=1234)
xu, yu = dna_pair(1_000_000, seed=5678, related=False)
for name, f in (("nw related", lambda: eng.nw(x, y, "dna")), ("lcs related", lambda: eng.lcs(x, y)), ("nw unrelated", lambda: eng.nw(xu, yu, "dna")),
                ("sw related", lambda: eng.sw(x, y, "dna"))):
    r = min((f() for _ in range(2)), key=lambda r: r.traceback_ms)
    print(f"{name}: fill kernel {r.kernel_fill_ms:.1f} ms, traceback + D2H {r.traceback_ms:.2f} ms, {len(r.ops)} moves", flush=True)
