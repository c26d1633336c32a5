"""Bit-vector LCS fill time per library variant (SA_LIBRARY): tools/perf_lcs_variants.py lib1.so lib2.so ...
Each variant runs in a process of its own; prints fill kernel ms (best of 3), traceback ms, score and a hash of the ops."""
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
    best, tb = 1e9, 0.0
    for _ in range(4):
        r = eng.lcs(x, y)
        if r.kernel_fill_ms < best:
            best, tb = r.kernel_fill_ms, r.traceback_ms
    print(f"{os.environ.get('SA_LIBRARY', 'default'):60s} fill {best:8.3f} ms  traceback {tb:7.3f} ms  score {r.score}  ops {hashlib.md5(r.ops.tobytes()).hexdigest()[:12]}", flush=True)
    sys.exit(0)

for lib in sys.argv[1:]:
    env = dict(os.environ, SA_LIBRARY=str(Path(lib).resolve()))
    rc = subprocess.run([sys.executable, "-u", __file__, "--one"], env=env, timeout=300).returncode
    if rc != 0:
        print(f"{lib}: rc {rc}", flush=True)
        sys.exit(rc)
