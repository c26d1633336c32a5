"""Multi-process test of the column-stripe orchestration (sequencealigner_amd/stripes.py) on CPU:
world_size 2 and 3 over gloo, oracle-backed stripe engine, results equal the whole-matrix oracle."""
import os
import subprocess
import sys
from pathlib import Path

import pytest

ROOT = Path(__file__).resolve().parent.parent


@pytest.mark.parametrize("world", [2, 3])
def test_stripes_over_gloo(world):
    port = 29500 + world + (os.getpid() % 200)
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={world}", "--master-addr", "127.0.0.1",
           "--master-port", str(port), str(ROOT / "tests" / "helpers" / "run_stripes_gloo.py")]
    env = dict(os.environ, OMP_NUM_THREADS="1")
    p = subprocess.run(cmd, capture_output=True, text=True, timeout=600, env=env)
    assert p.returncode == 0, p.stdout[-3000:] + p.stderr[-3000:]
    assert p.stdout.count("STRIPES_OK") == world, p.stdout[-3000:]
