"""GPU parity of the persistent strip-pipelined fill (one launch, in-kernel hand-off every 64
steps) against the oracle and against the per-diagonal launch path."""
import numpy as np
import pytest

from conftest import DNA, PROT20, PROT24, mutate, rand_seq
from test_gpu_parity import SHAPES, check

pytestmark = pytest.mark.gpu


@pytest.fixture()
def pengine(engine):
    engine.set_option("persistent", 1)
    yield engine


@pytest.mark.parametrize("kind", ["nw", "sw", "lcs"])
@pytest.mark.parametrize("K", [1, 2, 4, 8])
def test_persistent_random_shapes(pengine, oracle, kind, K):
    rng = np.random.default_rng(200 + K)
    pengine.set_tuning(K, 0)
    for (m, n) in SHAPES + [(5000, 3000), (300, 9000)]:
        g = check(kind, pengine, oracle, rand_seq(rng, m, DNA), rand_seq(rng, n, DNA), "dna")
        assert g.n_launches == 1


@pytest.mark.parametrize("kind", ["nw", "sw", "lcs"])
def test_persistent_tie_storms_and_protein(pengine, oracle, kind):
    rng = np.random.default_rng(77)
    for x, y in [("A" * 3000, "A" * 2570), ("AC" * 2000, "CA" * 1800), ("GATTACA" * 640, "GATACA" * 700)]:
        check(kind, pengine, oracle, x, y, "dna")
    if kind != "lcs":
        x = rand_seq(rng, 2500, PROT24)
        check(kind, pengine, oracle, x, mutate(rng, x, PROT20, 0.2, 0.05, 0.05), "protein")
        check(kind, pengine, oracle, x, rand_seq(rng, 1900, PROT24), "protein", -11.0, -1.0)


@pytest.mark.parametrize("wps", [1, 2, 4])
def test_persistent_many_strips_wave_counts(pengine, oracle, wps):
    """More strips than resident waves: every wave draws several tickets."""
    from sequencealigner_amd.synth import dna_pair

    pengine.set_option("waves_per_simd", wps)
    pengine.set_tuning(1, 0)
    x, y = dna_pair(40000, seed=9)
    o = oracle.nw(x, y, "dna", packed=True)
    g = pengine.nw(x, y, "dna")
    assert g.score == o.score and np.array_equal(g.ops, o.ops)
    o = oracle.lcs(x, y, packed=True)
    g = pengine.lcs(x, y)
    assert g.score == o.score and np.array_equal(g.ops, o.ops)


def test_persistent_equals_launch_path_at_100k(engine):
    from sequencealigner_amd.synth import dna_pair

    x, y = dna_pair(100000, seed=1234)
    engine.set_option("persistent", 0)
    a = engine.nw(x, y, "dna")
    engine.set_option("persistent", 1)
    try:
        for rep in range(3):
            b = engine.nw(x, y, "dna")
            assert b.n_launches == 1
            assert a.score == b.score and np.array_equal(a.ops, b.ops)
        sa_, sb = None, None
        engine.set_option("persistent", 0)
        sa_ = engine.sw(x, y, "dna")
        engine.set_option("persistent", 1)
        sb = engine.sw(x, y, "dna")
        assert (sa_.score, sa_.end, sa_.start) == (sb.score, sb.end, sb.start) and np.array_equal(sa_.ops, sb.ops)
    finally:
        engine.set_option("persistent", 1)
