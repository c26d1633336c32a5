"""GPU: 20 s of the randomised parity run (tools/fuzz_gpu.py: random shapes, alphabets, gap settings,
rows per lane, both schedules, checkpoint bands, score-only) against the oracle."""
import subprocess
import sys
from pathlib import Path

import pytest

pytestmark = pytest.mark.gpu


def test_random_parity_20s():
    root = Path(__file__).resolve().parent.parent
    p = subprocess.run([sys.executable, "-u", str(root / "tools" / "fuzz_gpu.py"), "20", "3"], capture_output=True, text=True, timeout=300)
    assert p.returncode == 0 and "FUZZ_OK" in p.stdout, p.stdout[-2000:] + p.stderr[-2000:]


def test_random_batch_parity_15s():
    """tools/fuzz_gpu_batch.py: the five many-pair entry points on random batches (length edges, chunked code budgets)."""
    root = Path(__file__).resolve().parent.parent
    p = subprocess.run([sys.executable, "-u", str(root / "tools" / "fuzz_gpu_batch.py"), "15", "5"], capture_output=True, text=True, timeout=300)
    assert p.returncode == 0 and "FUZZ_BATCH_OK" in p.stdout, p.stdout[-2000:] + p.stderr[-2000:]
