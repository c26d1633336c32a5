import os
import sys
from pathlib import Path

import numpy as np
import pytest

ROOT = Path(__file__).resolve().parent.parent
if str(ROOT) not in sys.path:
    sys.path.insert(0, str(ROOT))


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def oracle():
    from oracle import Oracle

    return Oracle()


@pytest.fixture(scope="session")
def _engine_session():
    import sequencealigner_amd as sa

    eng = sa.Engine(0)  # raises loudly when the HIP library or the GPU is missing
    yield eng
    eng.close()


@pytest.fixture()
def engine(_engine_session):
    """The engine in its default configuration (persistent strip pipeline, automatic tuning)."""
    eng = _engine_session
    eng.set_option("persistent", 1)
    eng.set_option("waves_per_simd", 1)
    eng.set_option("lcs_bits", 1)
    eng.set_option("max_workgroups", 0)
    eng.set_option("group", 0)
    eng.set_tuning(0, 0)
    yield eng
    eng.set_option("persistent", 1)
    eng.set_option("waves_per_simd", 1)
    eng.set_option("lcs_bits", 1)
    eng.set_option("max_workgroups", 0)
    eng.set_option("group", 0)
    eng.set_tuning(0, 0)


@pytest.fixture(params=["persistent", "launches"])
def engine_both(engine, request):
    """Both fill schedules: one persistent launch / one launch per tile diagonal."""
    engine.set_option("persistent", 1 if request.param == "persistent" else 0)
    yield engine


DNA = "ACGT"
IUPAC = "ACGTURYSWKMBDHVNX"
PROT20 = "ARNDCQEGHILKMFPSTWYV"
PROT24 = "ARNDCQEGHILKMFPSTWYVBZX*"


def rand_seq(rng, n, alphabet):
    a = np.frombuffer(alphabet.encode(), dtype=np.uint8)
    return a[rng.integers(0, len(a), size=n)].tobytes().decode()


def mutate(rng, s, alphabet, sub=0.06, dele=0.02, ins=0.02):
    """y = s with per-base substitutions / deletions / insertions (SURVEY 8d generator shape)."""
    out = []
    for ch in s:
        r = rng.random()
        if r < dele:
            continue
        if r < dele + sub:
            out.append(alphabet[rng.integers(0, len(alphabet))])
        else:
            out.append(ch)
        if rng.random() < ins:
            out.append(alphabet[rng.integers(0, len(alphabet))])
    return "".join(out)
