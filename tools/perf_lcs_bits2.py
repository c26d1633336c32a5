"""Dev tool: bit-vector LCS, with and without the traceback record (is the step bound by its stores?)."""
import sys
from pathlib import Path

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
import sequencealigner_amd as sa
from sequencealigner_amd.engine import SA_FLAG_NO_TRACEBACK
from sequencealigner_amd.synth import dna_pair

eng = sa.Engine(0)
x, y = dna_pair(1_000_000, seed=1234)
for m, n in ((2048, 1_000_000), (100_000, 100_000), (1_000_000, 1_000_000)):
    for fl in (0, SA_FLAG_NO_TRACEBACK):
        r = min((eng.lcs(x[:m], y[:n], flags=fl) for _ in range(2)), key=lambda r: r.kernel_fill_ms)
        print(f"lcs {m} x {n} flags={fl}: fill kernel {r.kernel_fill_ms:.2f} ms = {r.kernel_fill_ms * 1e6 / n:.1f} ns per column", flush=True)
