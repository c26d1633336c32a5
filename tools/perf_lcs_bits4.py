"""Dev tool: bit-vector LCS fill time against the number of strips writing their records at once."""
import sys
from pathlib import Path

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
import sequencealigner_amd as sa
from sequencealigner_amd.synth import dna_pair

eng = sa.Engine(0)
x, y = dna_pair(1_000_000, seed=1234)
for wpb, m in [(w, m) for w in (4, 2, 1) for m in (100_000, 524_288, 1_000_000)] + [(0, m) for m in (2048, 65_536, 131_072, 262_144, 524_288, 786_432, 1_000_000)]:
    eng.set_option("lcs_bits_waves_per_block", wpb)
    r = min((eng.lcs(x[:m], y) for _ in range(2)), key=lambda r: r.kernel_fill_ms)
    S = (m + 2047) // 2048
    print(f"waves per workgroup {wpb or 'auto'}: lcs {m} x {len(y)}: {S} strips, fill kernel {r.kernel_fill_ms:.2f} ms = {r.kernel_fill_ms * 1e6 / len(y):.1f} ns per column, "
          f"{S * 1000192 * 256 / r.kernel_fill_ms / 1e9:.2f} TB/s of records", flush=True)
