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


@pytest.fixture(scope="module")
def many_strips_want(oracle):
    """The oracle's NW and LCS of the 40 kb pair, once for the three wave counts (8 s of host time each)."""
    from sequencealigner_amd.synth import dna_pair

    x, y = dna_pair(40000, seed=9)
    return x, y, oracle.nw(x, y, "dna", packed=True), oracle.lcs(x, y, packed=True)


@pytest.mark.parametrize("wps", [1, 2, 4])
def test_persistent_many_strips_wave_counts(pengine, many_strips_want, wps):
    """More strips than resident waves: every wave draws several tickets."""
    x, y, o_nw, o_lcs = many_strips_want
    pengine.set_option("waves_per_simd", wps)
    pengine.set_tuning(1, 0)
    g = pengine.nw(x, y, "dna")
    assert g.score == o_nw.score and np.array_equal(g.ops, o_nw.ops)
    g = pengine.lcs(x, y)
    assert g.score == o_lcs.score and np.array_equal(g.ops, o_lcs.ops)


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


def test_checkpointed_traceback_equals_oracle(engine, oracle):
    """codes that do not fit are replaced by per-strip checkpoints and the traceback
    recomputes `checkpoint_band` strips at a time; results are identical."""
    import numpy as np

    from conftest import DNA, PROT20, mutate, rand_seq

    rng = np.random.default_rng(77)
    try:
        for (m, K, band) in [(700, 1, 1), (3000, 2, 3), (5000, 4, 2), (2600, 8, 1), (4000, 1, 7), (3000, 2, 1000)]:
            x = rand_seq(rng, m, DNA)
            y = mutate(rng, x, DNA)
            engine.set_tuning(K, 0)
            engine.set_option("checkpoint_band", band)
            for kind in ("nw", "lcs", "sw"):
                want = oracle.lcs(x, y) if kind == "lcs" else getattr(oracle, kind)(x, y, "dna")
                got = engine.lcs(x, y) if kind == "lcs" else getattr(engine, kind)(x, y, "dna")
                assert got.score == want.score and tuple(got.end) == tuple(want.end) and tuple(got.start) == tuple(want.start), (kind, m, K, band)
                assert np.array_equal(got.ops, want.ops), (kind, m, K, band)
                assert got.n_launches >= 2
        # unrelated sequences (gappy path that wanders), protein, a local hit far from the corner, empty local alignment
        engine.set_tuning(2, 0)
        engine.set_option("checkpoint_band", 2)
        x, y = rand_seq(rng, 2500, DNA), rand_seq(rng, 3100, DNA)
        for kind in ("nw", "sw"):
            want, got = getattr(oracle, kind)(x, y, "dna"), getattr(engine, kind)(x, y, "dna")
            assert got.score == want.score and np.array_equal(got.ops, want.ops), kind
        xp = rand_seq(rng, 1500, PROT20)
        yp = mutate(rng, xp, PROT20, 0.2, 0.05, 0.05)
        want, got = oracle.nw(xp, yp, "protein", -11.0, -1.0), engine.nw(xp, yp, "protein", -11.0, -1.0)
        assert got.score == want.score and np.array_equal(got.ops, want.ops)
        core = rand_seq(rng, 300, DNA)
        xl, yl = rand_seq(rng, 900, DNA) + core + rand_seq(rng, 1400, DNA), rand_seq(rng, 200, DNA) + core + rand_seq(rng, 2000, DNA)
        want, got = oracle.sw(xl, yl, "dna"), engine.sw(xl, yl, "dna")
        assert (got.score, tuple(got.end), tuple(got.start)) == (want.score, tuple(want.end), tuple(want.start)) and np.array_equal(got.ops, want.ops)
        got = engine.sw("A" * 700, "C" * 900, "dna")
        assert got.score == 0 and len(got.ops) == 0
    finally:
        engine.set_option("checkpoint_band", 0)
        engine.set_tuning(0, 0)


def test_lcs_large_alphabet_keeps_the_compare_kernel(engine, oracle):
    """LCS takes its match term from an LDS profile when the pair holds at most 31 distinct bytes; beyond
    that (and for every byte value, 0 and 255 included) the byte-compare kernel runs: same results."""
    rng = np.random.default_rng(77)
    for nsym in (2, 31, 32, 200, 256):
        syms = rng.permutation(256)[:nsym].astype(np.uint8)
        if nsym == 256:
            syms = np.arange(256, dtype=np.uint8)
        x = syms[rng.integers(0, nsym, 1500)]
        y = np.concatenate([x[200:900], syms[rng.integers(0, nsym, 400)], x[1000:1400]])
        g, o = engine.lcs(x, y), oracle.lcs(x, y)
        assert g.score == o.score and np.array_equal(g.ops, o.ops), nsym


def test_lcs_sixteen_rows_per_lane(pengine, oracle):
    """The profile-driven LCS has a 16-rows-per-lane instance (1024-row strips: 1 Mb of rows is one round of waves).
    Forced here on ragged shapes, tie storms and a many-strip pair; NW / SW and the byte-compare LCS take 8 instead."""
    from sequencealigner_amd.synth import dna_pair

    rng = np.random.default_rng(1600)
    pengine.set_tuning(16, 0)
    try:
        for (m, n) in SHAPES + [(1023, 1025), (1024, 64), (1025, 3000), (5000, 3000), (300, 9000), (2049, 1)]:
            g = check("lcs", pengine, oracle, rand_seq(rng, m, DNA), rand_seq(rng, n, DNA), "dna")
            assert g.n_launches == 1 and (g.rows_per_lane == 16 or m == 0 or n == 0), (m, n, g.rows_per_lane)
        for x, y in [("A" * 3000, "A" * 2570), ("AC" * 2000, "CA" * 1800), ("GATTACA" * 640, "GATACA" * 700)]:
            check("lcs", pengine, oracle, x, y, "dna")
        x = rand_seq(rng, 2500, PROT24)
        check("lcs", pengine, oracle, x, mutate(rng, x, PROT20, 0.2, 0.05, 0.05), "protein")
        x, y = dna_pair(40000, seed=11)
        o, g = oracle.lcs(x, y, packed=True), pengine.lcs(x, y)
        assert g.rows_per_lane == 16 and g.score == o.score and np.array_equal(g.ops, o.ops)
        # checkpointed traceback over 1024-row strips
        pengine.set_option("checkpoint_band", 2)
        x = rand_seq(rng, 5000, DNA)
        y = mutate(rng, x, DNA)
        o, g = oracle.lcs(x, y), pengine.lcs(x, y)
        assert g.score == o.score and np.array_equal(g.ops, o.ops) and g.n_launches >= 2
        pengine.set_option("checkpoint_band", 0)
        # no 16-row instances for NW / SW / more than 31 distinct bytes: 8 rows per lane then, same results
        g = check("nw", pengine, oracle, rand_seq(rng, 3000, DNA), rand_seq(rng, 2000, DNA), "dna")
        assert g.rows_per_lane == 8
        syms = rng.permutation(256)[:200].astype(np.uint8)
        xb, yb = syms[rng.integers(0, 200, 3000)], syms[rng.integers(0, 200, 2500)]
        g, o = pengine.lcs(xb, yb), oracle.lcs(xb, yb)
        assert g.rows_per_lane == 8 and g.score == o.score and np.array_equal(g.ops, o.ops)
    finally:
        pengine.set_option("checkpoint_band", 0)
        pengine.set_tuning(0, 0)
