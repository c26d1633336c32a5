"""Dev tool: strip-chain diagnostics of the persistent pipeline.  Needs a library built with -DSA_DEBUG_HANDOFF
(SA_LIBRARY=sequencealigner_amd/libseqalign_dbg.so); the library prints one [handoff] line per fill to stderr."""
import sys
from pathlib import Path

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
import sequencealigner_amd as sa
from sequencealigner_amd.synth import dna_pair

eng = sa.Engine(0)
for spec in (sys.argv[1:] or ["100000:2", "100000:1", "100000:4", "1000000:8"]):
    n, K = (int(v) for v in spec.split(":"))
    x, y = dna_pair(n, seed=1234)
    eng.set_tuning(K, 0)
    for _ in range(2):
        r = eng.nw(x, y, "dna")
    print(f"n={n} K={K}: fill kernel {r.kernel_fill_ms:.2f} ms", flush=True)
