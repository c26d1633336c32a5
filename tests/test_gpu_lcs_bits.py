"""GPU parity of the bit-vector LCS (sa_lcsbits.hip: 32 rows per lane, one bit of traceback record per cell) against
the oracle: score and the reference's exact pointer choices (main.cpp:1577-1587, 1602-1634)."""
import numpy as np
import pytest

from conftest import DNA, PROT20, PROT24, mutate, rand_seq
from test_gpu_parity import SHAPES, check

pytestmark = pytest.mark.gpu


def test_bits_random_shapes(engine, oracle):
    rng = np.random.default_rng(3200)
    for (m, n) in SHAPES + [(31, 33), (32, 32), (33, 31), (2047, 2049), (2048, 64), (2049, 1), (1, 2049), (4097, 4100), (5000, 3000),
                            (300, 9000), (9000, 300), (70, 20000), (20000, 70)]:
        for alphabet in (DNA, "AC", PROT24):
            g = check("lcs", engine, oracle, rand_seq(rng, m, alphabet), rand_seq(rng, n, alphabet), "dna")
            assert g.rows_per_lane == 32 and g.n_launches == 1, (m, n, g.rows_per_lane)


def test_bits_tie_storms_and_related_pairs(engine, oracle):
    rng = np.random.default_rng(3201)
    for x, y in [("A" * 3000, "A" * 2570), ("AC" * 2000, "CA" * 1800), ("GATTACA" * 640, "GATACA" * 700), ("A" * 5000, "C" * 4000),
                 ("A" * 2048 + "C" * 2048, "C" * 2048 + "A" * 2048)]:
        g = check("lcs", engine, oracle, x, y, "dna")
        assert g.rows_per_lane == 32
    for m in (700, 2500, 6000, 12000):
        x = rand_seq(rng, m, DNA)
        for sub, ins, dele in ((0.02, 0.0, 0.0), (0.1, 0.03, 0.03), (0.3, 0.15, 0.15)):
            check("lcs", engine, oracle, x, mutate(rng, x, DNA, sub, ins, dele), "dna")
        xp = rand_seq(rng, m, PROT20)
        check("lcs", engine, oracle, xp, mutate(rng, xp, PROT20, 0.2, 0.05, 0.05), "protein")


def test_bits_many_strips_and_wave_counts(engine, oracle):
    """More strips than one workgroup holds, related and unrelated pairs (the walker's window path and its re-centring)."""
    from sequencealigner_amd.synth import dna_pair

    for wps in (1, 2):
        engine.set_option("waves_per_simd", wps)
        for related in (True, False):
            x, y = dna_pair(40000, seed=21 + wps, related=related)
            o, g = oracle.lcs(x, y, packed=True), engine.lcs(x, y)
            assert g.rows_per_lane == 32
            assert g.score == o.score and np.array_equal(g.ops, o.ops), (wps, related)
    x, y = dna_pair(30000, seed=5)
    o, g = oracle.lcs(x[:30000], y[:9000], packed=True), engine.lcs(x[:30000], y[:9000])
    assert g.score == o.score and np.array_equal(g.ops, o.ops)


def test_bits_equals_cell_kernels_at_100k(engine):
    from sequencealigner_amd.synth import dna_pair

    x, y = dna_pair(100000, seed=1234)
    engine.set_option("lcs_bits", 0)
    try:
        a = engine.lcs(x, y)
        assert a.rows_per_lane != 32
    finally:
        engine.set_option("lcs_bits", 1)
    for rep in range(2):
        b = engine.lcs(x, y)
        assert b.rows_per_lane == 32 and b.score == a.score and np.array_equal(a.ops, b.ops)
    xu, yu = dna_pair(100000, seed=99, related=False)
    engine.set_option("lcs_bits", 0)
    try:
        a = engine.lcs(xu, yu)
    finally:
        engine.set_option("lcs_bits", 1)
    b = engine.lcs(xu, yu)
    assert b.score == a.score and np.array_equal(a.ops, b.ops)


def test_bits_score_only_and_fallbacks(engine, oracle):
    import sequencealigner_amd.engine as sa

    rng = np.random.default_rng(3203)
    x, y = rand_seq(rng, 5000, DNA), rand_seq(rng, 4100, DNA)
    o = oracle.lcs(x, y)
    g = engine.lcs(x, y, flags=sa.SA_FLAG_NO_TRACEBACK)
    assert g.rows_per_lane == 32 and g.score == o.score and len(g.ops) == 0
    # every byte value present (no code left for the padding symbol), a forced rows-per-lane, a matrix dump: the
    # cell-by-cell kernels, same results; up to 255 distinct bytes the bit-vector kernel (256 B of LDS per symbol and wave:
    # beyond 127 symbols a workgroup's LDS passes 64 KB, beyond 160 symbols only two waves fit a workgroup)
    syms = rng.permutation(256).astype(np.uint8)
    xb, yb = syms[rng.integers(0, 256, 3000)], syms[rng.integers(0, 256, 2500)]
    xb[:256] = syms  # all 256 present
    g, ob = engine.lcs(xb, yb), oracle.lcs(xb, yb)
    assert g.rows_per_lane != 32 and g.score == ob.score and np.array_equal(g.ops, ob.ops)
    for nsym in (255, 200, 161, 160, 128, 127, 64, 63):
        sy = syms[:nsym]
        xb = sy[rng.integers(0, nsym, 5000)]
        yb = np.concatenate([xb[300:2100], sy[rng.integers(0, nsym, 700)], xb[2500:4800]])
        xb[:nsym] = sy
        for wpb in (0, 4, 1):
            engine.set_option("lcs_bits_waves_per_block", wpb)
            g, ob = engine.lcs(xb, yb), oracle.lcs(xb, yb)
            assert g.rows_per_lane == 32 and g.score == ob.score and np.array_equal(g.ops, ob.ops), (nsym, wpb)
    engine.set_option("lcs_bits_waves_per_block", 0)
    # the small alphabet of a low-complexity byte text next to a wide one in the same call sequence (LDS size changes per call)
    g = check("lcs", engine, oracle, rand_seq(rng, 3000, DNA), rand_seq(rng, 2500, DNA), "dna")
    assert g.rows_per_lane == 32
    engine.set_tuning(4, 0)
    g = engine.lcs(x, y)
    engine.set_tuning(0, 0)
    assert g.rows_per_lane == 4 and g.score == o.score and np.array_equal(g.ops, o.ops)
    g = engine.lcs(x[:300], y[:200], flags=sa.SA_FLAG_KEEP_DP)
    od = oracle.lcs(x[:300], y[:200], keep_dp=True)
    assert g.rows_per_lane != 32 and np.array_equal(g.dp, od.dp)


def test_bits_waves_per_workgroup(engine, oracle):
    """The strips of the bit-vector kernel are independent waves; how many share a workgroup changes nothing."""
    rng = np.random.default_rng(3204)
    x = rand_seq(rng, 9000, DNA)
    y = mutate(rng, x, DNA, 0.1, 0.03, 0.03)
    o = oracle.lcs(x, y)
    try:
        for wpb in (1, 2, 4, 0):
            engine.set_option("lcs_bits_waves_per_block", wpb)
            g = engine.lcs(x, y)
            assert g.rows_per_lane == 32 and g.score == o.score and np.array_equal(g.ops, o.ops), wpb
    finally:
        engine.set_option("lcs_bits_waves_per_block", 0)


def test_bits_checkpointed_bands(engine, oracle):
    """A record that does not fit: the fill keeps every strip's carry records and no record, the traceback runs the
    strips again `checkpoint_band` at a time with their record (forced here; at 2 Mb x 2 Mb the engine does it by itself,
    tests/test_gpu_z_fullsize.py).  Same score and op list; band edges at every strip boundary, paths that cross several
    bands, leave through row 0 or column 0, and start in a ragged last strip."""
    from sequencealigner_amd.synth import dna_pair

    rng = np.random.default_rng(3205)
    try:
        for (m, n, band) in [(700, 900, 1), (2048, 2048, 1), (2049, 2047, 1), (4096, 4100, 1), (4097, 300, 1), (300, 9000, 1), (9000, 300, 2),
                             (6145, 6100, 2), (12000, 11000, 3), (12000, 11000, 1), (5000, 5100, 1000)]:
            engine.set_option("checkpoint_band", band)
            for alphabet in (DNA, "AC"):
                g = check("lcs", engine, oracle, rand_seq(rng, m, alphabet), rand_seq(rng, n, alphabet), "dna")
                assert g.rows_per_lane == 32 and 2 <= g.n_launches <= 1 + -(-((m + 2047) // 2048) // band), (m, n, band, g.n_launches)
            x = rand_seq(rng, m, DNA)
            g = check("lcs", engine, oracle, x, mutate(rng, x, DNA, 0.1, 0.03, 0.03), "dna")
            assert g.rows_per_lane == 32 and g.n_launches >= 2
        engine.set_option("checkpoint_band", 2)
        for x, y in [("A" * 5000, "A" * 4570), ("AC" * 3000, "CA" * 2800), ("A" * 5000, "C" * 4000), ("A" * 2048 + "C" * 2048, "C" * 2048 + "A" * 2048)]:
            g = check("lcs", engine, oracle, x, y, "dna")
            assert g.rows_per_lane == 32 and g.n_launches >= 2
        engine.set_option("checkpoint_band", 3)
        for related in (True, False):
            x, y = dna_pair(40000, seed=31, related=related)
            o, g = oracle.lcs(x, y, packed=True), engine.lcs(x, y)
            assert g.rows_per_lane == 32 and g.n_launches == 1 + 7 and g.score == o.score and np.array_equal(g.ops, o.ops), related
        # the score-only call has nothing to recompute
        import sequencealigner_amd.engine as sa

        g = engine.lcs(x, y, flags=sa.SA_FLAG_NO_TRACEBACK)
        assert g.rows_per_lane == 32 and g.n_launches == 1 and g.score == o.score
    finally:
        engine.set_option("checkpoint_band", 0)
    # and the one-slab run afterwards is the old one
    g = engine.lcs(x, y)
    assert g.n_launches == 1 and g.score == o.score and np.array_equal(g.ops, o.ops)
