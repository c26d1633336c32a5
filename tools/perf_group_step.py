"""Dev tool: ns per step of fill_group against fill_persistent: one strip, one group (4 strips), two groups, over 1 M columns."""
import sys
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
import sequencealigner_amd as sa
from sequencealigner_amd.synth import dna_pair

eng = sa.Engine(0)
x, y = dna_pair(1_000_000, seed=1234)
kinds = sys.argv[1].split(",") if len(sys.argv) > 1 else ["nw"]
for kind in kinds:
    for K in (1, 2, 4, 8):
        eng.set_tuning(K, 0)
        row = []
        for group in (0, 1):
            eng.set_option("group", group)
            for strips in (1, 4, 8):
                t = min(getattr(eng, kind)(x[:64 * K * strips], y, "dna").kernel_fill_ms for _ in range(2))
                row.append(f"g{group} S={strips}: {t * 1e6 / (len(y) + 63):.1f}")
        print(f"{kind} K={K} ns/step  " + "  ".join(row), flush=True)
eng.close()
