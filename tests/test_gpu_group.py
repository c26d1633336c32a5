"""GPU parity of fill_group (csrc/sa_group.h: four consecutive strips per workgroup, bottom rows handed down through
LDS rings with progress words; option "group") against the oracle: every K, ragged shapes (partial groups, fewer columns
than a block), tie storms, proteins, fractional penalties, several rounds of group tickets on few workgroups (ring-slot
reuse), two workgroups per CU, checkpointed runs, score-only calls."""
import numpy as np
import pytest

from conftest import DNA, PROT20, PROT24, mutate, rand_seq
from test_gpu_parity import SHAPES, check

pytestmark = pytest.mark.gpu


@pytest.fixture()
def gengine(engine):
    engine.set_option("persistent", 1)
    engine.set_option("group", 1)
    yield engine
    engine.set_option("group", 0)
    engine.set_tuning(0, 0)


@pytest.mark.parametrize("kind", ["nw", "sw"])
@pytest.mark.parametrize("K", [1, 2, 4, 8])
def test_group_random_shapes(gengine, oracle, kind, K):
    rng = np.random.default_rng(300 + K)
    gengine.set_tuning(K, 0)
    for (m, n) in SHAPES + [(5000, 3000), (300, 9000), (64 * K * 4, 500), (64 * K * 4 + 1, 130), (64 * K * 9 - 1, 2000), (64 * K * 5, 63)]:
        g = check(kind, gengine, oracle, rand_seq(rng, m, DNA), rand_seq(rng, n, DNA), "dna")
        assert g.n_launches == 1


@pytest.mark.parametrize("kind", ["nw", "sw"])
def test_group_tie_storms_protein_and_fractional_gaps(gengine, oracle, kind):
    rng = np.random.default_rng(78)
    for x, y in [("A" * 3000, "A" * 2570), ("AC" * 2000, "CA" * 1800), ("GATTACA" * 640, "GATACA" * 700)]:
        check(kind, gengine, oracle, x, y, "dna")
    x = rand_seq(rng, 2500, PROT24)
    check(kind, gengine, oracle, x, mutate(rng, x, PROT20, 0.2, 0.05, 0.05), "protein")
    check(kind, gengine, oracle, x, rand_seq(rng, 1900, PROT24), "protein", -11.0, -1.0)
    for K in (1, 2, 4, 8):
        gengine.set_tuning(K, 0)
        xd = rand_seq(rng, 2100, DNA)
        for go, ge in [(-5.5, -0.5), (-4.25, -1.75), (-0.5, -0.5)]:
            check(kind, gengine, oracle, xd, mutate(rng, xd, DNA), "dna", go, ge)


_ORACLE_MEMO = {}


@pytest.mark.parametrize("max_wg,wps", [(1, 1), (3, 1), (0, 2), (5, 2)])
def test_group_many_rounds_of_tickets(gengine, oracle, max_wg, wps):
    """More groups than workgroups: every workgroup draws several group tickets, the record ring wraps (generations)."""
    from sequencealigner_amd.synth import dna_pair

    gengine.set_option("max_workgroups", max_wg)
    gengine.set_option("waves_per_simd", wps)
    x, y = dna_pair(30000, seed=11)
    for kind in ("nw", "sw"):
        for K in (1, 2):
            gengine.set_tuning(K, 0)
            xs, ys = (x[: 12000 if max_wg else 30000], y) if kind == "nw" else (x[: 6000 if max_wg else 14000], y[:12000])
            key = (kind, len(xs), len(ys))  # the oracle's result does not depend on K or the launch shape: once per shape
            if key not in _ORACLE_MEMO:
                _ORACLE_MEMO[key] = oracle.nw(xs, ys, "dna", packed=True) if kind == "nw" else oracle.sw(xs, ys, "dna")
            o = _ORACLE_MEMO[key]
            g = getattr(gengine, kind)(xs, ys, "dna")
            assert g.score == o.score and g.end == o.end and g.start == o.start and np.array_equal(g.ops, o.ops), (kind, K)


def test_group_checkpointed_and_score_only(gengine, oracle):
    from sequencealigner_amd.engine import SA_FLAG_NO_TRACEBACK

    rng = np.random.default_rng(79)
    try:
        for (m, K, band) in [(700, 1, 1), (3000, 2, 3), (5000, 4, 2), (2600, 8, 1)]:
            x = rand_seq(rng, m, DNA)
            y = mutate(rng, x, DNA)
            gengine.set_tuning(K, 0)
            gengine.set_option("checkpoint_band", band)
            for kind in ("nw", "sw"):
                want, got = getattr(oracle, kind)(x, y, "dna"), getattr(gengine, kind)(x, y, "dna")
                assert got.score == want.score and tuple(got.end) == tuple(want.end) and np.array_equal(got.ops, want.ops), (kind, m, K, band)
                assert got.n_launches >= 2
            gengine.set_option("checkpoint_band", 0)
            got = gengine.nw(x, y, "dna", flags=SA_FLAG_NO_TRACEBACK)
            assert got.score == oracle.nw(x, y, "dna").score
    finally:
        gengine.set_option("checkpoint_band", 0)


def test_group_equals_persistent_at_100k(gengine):
    from sequencealigner_amd.synth import dna_pair

    x, y = dna_pair(100000, seed=1234)
    for kind in ("nw", "sw"):
        gengine.set_option("group", 0)
        a = getattr(gengine, kind)(x, y, "dna")
        gengine.set_option("group", 1)
        b = getattr(gengine, kind)(x, y, "dna")
        assert (a.score, a.end, a.start) == (b.score, b.end, b.start) and np.array_equal(a.ops, b.ops), kind
