"""Dev tool: LCS fill / traceback times, bit-vector kernel vs the cell-by-cell kernels."""
import sys
from pathlib import Path

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
import sequencealigner_amd as sa
from sequencealigner_amd.synth import dna_pair

eng = sa.Engine(0)
sizes = [int(a) for a in sys.argv[1:]] or [100_000, 1_000_000]
for n in sizes:
    for related in (True, False):
        x, y = dna_pair(n, seed=1234 if related else 5678, related=related)
        for bits in (1, 0):
            if not bits and n > 400_000 and not related:
                continue
            eng.set_option("lcs_bits", bits)
            r = min((eng.lcs(x, y) for _ in range(2)), key=lambda r: r.kernel_fill_ms + r.traceback_ms)
            print(f"lcs {n} {'related' if related else 'unrelated'} bits={bits}: K={r.rows_per_lane} fill kernel {r.kernel_fill_ms:.2f} ms, traceback + D2H {r.traceback_ms:.2f} ms, "
                  f"score {r.score}, {len(r.ops)} moves, {n * n / r.kernel_fill_ms / 1e6:.0f} GCUPS fill", flush=True)
eng.set_option("lcs_bits", 1)
