"""Launched by tests/test_gpu_stripes.py::test_lcs_stripes_run_the_bit_vector_kernel with GPU_MAX_HW_QUEUES=16: the ranks of a
striped LCS are replayed one after the other on GPU 0 with real transfer streams (overlapped schedule), so that the
bit-vector kernel and its border conversions run; results against the oracle."""
import sys
from pathlib import Path

ROOT = Path(__file__).resolve().parent.parent.parent
sys.path.insert(0, str(ROOT))
sys.path.insert(0, str(ROOT / "tests"))

import sequencealigner_amd as sa
from oracle import Oracle
from test_gpu_stripes import lcs_bits_stripe_cases

eng = sa.Engine(0)
lcs_bits_stripe_cases(eng, Oracle())
eng.close()
print("LCS_BITS_STRIPES_OK", flush=True)
