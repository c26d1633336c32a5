"""LCS fill time at 1 Mb by rows per lane and waves per SIMD"""
import sys
from pathlib import Path

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
import sequencealigner_amd as sa
from sequencealigner_amd.synth import dna_pair

eng = sa.Engine(0)
x, y = dna_pair(1_000_000, seed=1234)
for wps in (1, 2):
    eng.set_option("waves_per_simd", wps)
    for K in (4, 8):
        eng.set_tuning(K, 0)
        t = min(eng.lcs(x, y).kernel_fill_ms for _ in range(2))
        print(f"waves/SIMD {wps} K {K}: {t:.1f} ms", flush=True)
