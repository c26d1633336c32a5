"""NW fill time per library variant (SA_LIBRARY), one process each: tools/perf_nw_variants.py [--size N] lib1.so lib2.so ..."""
import hashlib
import os
import subprocess
import sys
from pathlib import Path

ROOT = Path(__file__).resolve().parent.parent
if len(sys.argv) > 1 and sys.argv[1] == "--one":
    sys.path.insert(0, str(ROOT))
    from sequencealigner_amd.engine import Engine
    from sequencealigner_amd.synth import dna_pair

    n = int(os.environ.get("N", "1000000"))
    x, y = dna_pair(n, seed=1234)
    eng = Engine(0)
    best = 1e9
    for _ in range(3 if n > 200000 else 8):
        r = eng.nw(x, y, "dna")
        best = min(best, r.kernel_fill_ms)
    print(f"{Path(os.environ.get('SA_LIBRARY', 'default')).name:32s} n={n} fill {best:8.3f} ms  traceback {r.traceback_ms:7.3f} ms  score {r.score}  ops {hashlib.md5(r.ops.tobytes()).hexdigest()[:12]}", flush=True)
    sys.exit(0)
for lib in sys.argv[1:]:
    env = dict(os.environ, SA_LIBRARY=str(Path(lib).resolve()))
    rc = subprocess.run([sys.executable, "-u", __file__, "--one"], env=env, timeout=300).returncode
    if rc != 0:
        sys.exit(rc)
