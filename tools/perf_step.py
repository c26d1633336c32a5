"""Dev tool: cost of one skewed step per rows-per-lane K: ONE strip (64 K rows) over 1 M columns, so the kernel
time is the strip's sweep and nothing else; then full problems.  SA_LIBRARY selects the build."""
import sys
from pathlib import Path

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
import sequencealigner_amd as sa
from sequencealigner_amd.synth import dna_pair

eng = sa.Engine(0)
eng.set_option("lcs_bits", 0)
x, y = dna_pair(1_000_000, seed=1234)
for kind in ("nw", "sw", "lcs"):
    row = []
    for K in (1, 2, 4, 8) + ((16,) if kind == "lcs" else ()):
        eng.set_tuning(K, 0)
        f = (lambda a, b: eng.lcs(a, b)) if kind == "lcs" else (lambda a, b: getattr(eng, kind)(a, b, "dna"))
        t = min(f(x[:64 * K], y).kernel_fill_ms for _ in range(3))
        row.append(f"K={K} {t * 1e6 / (len(y) + 63):.1f}")
    print(f"{kind} ns/step, one strip:", "  ".join(row), flush=True)
eng.set_tuning(0, 0)
eng.set_option("lcs_bits", 0)
for n in (100_000, 1_000_000):
    for K in (0, 8, 16):
        eng.set_tuning(K, 0)
        r = min((eng.nw(x[:n], y[:n], "dna") for _ in range(2)), key=lambda r: r.kernel_fill_ms)
        print(f"nw {n}: K={r.rows_per_lane} (asked {K}) fill kernel {r.kernel_fill_ms:.2f} ms", flush=True)
for n in (100_000, 1_000_000):
    for K in (0, 8, 16):
        eng.set_tuning(K, 0)
        r = min((eng.lcs(x[:n], y[:n]) for _ in range(2)), key=lambda r: r.kernel_fill_ms)
        print(f"lcs {n}: K={r.rows_per_lane} (asked {K}) fill kernel {r.kernel_fill_ms:.2f} ms", flush=True)
eng.set_tuning(0, 0)
