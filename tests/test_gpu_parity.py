"""GPU parity: the HIP path (through the C-ABI) against the CPU oracle, bit-exact
(integer DP: score, end/start cell and the complete traceback op list)."""
import numpy as np
import pytest

from conftest import DNA, IUPAC, PROT20, PROT24, mutate, rand_seq

pytestmark = pytest.mark.gpu


def check(kind, eng, orc, x, y, mode="dna", go=-5.0, ge=-1.0):
    if kind == "nw":
        g = eng.nw(x, y, mode, go, ge)
        o = orc.nw(x, y, mode, go, ge)
    elif kind == "sw":
        g = eng.sw(x, y, mode, go, ge)
        o = orc.sw(x, y, mode, go, ge)
    else:
        g = eng.lcs(x, y)
        o = orc.lcs(x, y)
    assert o.rc == 0
    ctx = f"{kind} m={len(x)} n={len(y)} mode={mode} go={go} ge={ge} K={g.rows_per_lane} C={g.tile_steps}"
    assert g.score == o.score, ctx
    assert g.end == o.end, ctx
    assert g.start == o.start, ctx
    assert len(g.ops) == len(o.ops), ctx
    assert np.array_equal(g.ops, o.ops), ctx
    return g


SHAPES = [(1, 1), (1, 7), (7, 1), (2, 3), (63, 64), (64, 63), (65, 65), (100, 257), (257, 100), (300, 1000),
          (513, 129), (1000, 64), (1210, 1390), (2049, 777)]


@pytest.mark.parametrize("kind", ["nw", "sw", "lcs"])
@pytest.mark.parametrize("K", [1, 2, 4, 8])
def test_random_dna_all_tilings(engine_both, oracle, kind, K):
    engine = engine_both
    rng = np.random.default_rng(100 + K)
    for C in (64, 128, 256):
        engine.set_tuning(K, C)
        for (m, n) in SHAPES:
            x = rand_seq(rng, m, DNA)
            y = rand_seq(rng, n, DNA)
            check(kind, engine, oracle, x, y, "dna")
    engine.set_tuning(0, 0)


@pytest.mark.parametrize("kind", ["nw", "sw", "lcs"])
def test_similar_pairs_auto_tuning(engine_both, oracle, kind):
    engine = engine_both
    rng = np.random.default_rng(7)
    engine.set_tuning(0, 0)
    for m in (500, 3000, 6000):
        x = rand_seq(rng, m, DNA)
        y = mutate(rng, x, DNA)
        check(kind, engine, oracle, x, y, "dna")


@pytest.mark.parametrize("kind", ["nw", "sw"])
def test_protein_full_alphabet(engine_both, oracle, kind):
    engine = engine_both
    rng = np.random.default_rng(11)
    for K in (1, 4, 8):
        engine.set_tuning(K, 128)
        for (m, n) in [(110, 116), (443, 400), (1000, 300), (777, 1555)]:
            x = rand_seq(rng, m, PROT24)
            y = rand_seq(rng, n, PROT24)
            check(kind, engine, oracle, x, y, "protein")
            y2 = mutate(rng, x, PROT20, 0.2, 0.05, 0.05)
            check(kind, engine, oracle, x, y2, "protein")
    engine.set_tuning(0, 0)


@pytest.mark.parametrize("kind", ["nw", "sw"])
def test_iupac_dna_index_quirk(engine, oracle, kind):
    """Quirk Q1: NUC.4.4 rows are indexed with char2idx's different order."""
    rng = np.random.default_rng(13)
    for (m, n) in [(19, 19), (200, 333), (1500, 900)]:
        x = rand_seq(rng, m, IUPAC)
        y = rand_seq(rng, n, IUPAC)
        check(kind, engine, oracle, x, y, "dna")


@pytest.mark.parametrize("kind", ["nw", "sw", "lcs"])
def test_tie_storms(engine_both, oracle, kind):
    engine = engine_both
    """Low-complexity inputs where every comparison ties: tie-breaks must survive the
    anti-diagonal evaluation order."""
    cases = [("A" * 300, "A" * 257), ("AC" * 200, "CA" * 180), ("ACGT" * 100, "ACGT" * 90 + "TTTT"),
             ("A" * 500, "T" * 400), ("AAAAAAAAAC" * 40, "AAAAAAAAAG" * 35), ("GATTACA" * 64, "GATACA" * 70)]
    for K in (1, 2, 8):
        engine.set_tuning(K, 64)
        for x, y in cases:
            check(kind, engine, oracle, x, y, "dna")
    engine.set_tuning(0, 0)


@pytest.mark.parametrize("go,ge", [(-10.0, -1.0), (-2.0, -2.0), (-1.0, -3.0), (0.0, 0.0), (-11.0, -1.0), (-3.0, 0.0)])
@pytest.mark.parametrize("kind", ["nw", "sw"])
def test_integral_gap_settings(engine_both, oracle, kind, go, ge):
    engine = engine_both
    rng = np.random.default_rng(17)
    x = rand_seq(rng, 700, DNA)
    y = mutate(rng, x, DNA, 0.1, 0.05, 0.05)
    check(kind, engine, oracle, x, y, "dna", go, ge)
    xp = rand_seq(rng, 333, PROT20)
    yp = mutate(rng, xp, PROT20, 0.3, 0.05, 0.05)
    check(kind, engine, oracle, xp, yp, "protein", go, ge)


@pytest.mark.parametrize("kind", ["nw", "sw", "lcs"])
def test_empty_and_tiny(engine, oracle, kind):
    for x, y in [("", "ACGTACGT"), ("ACGTACGT", ""), ("", ""), ("A", "A"), ("A", "C"), ("AC", "A")]:
        check(kind, engine, oracle, x, y, "dna")


def test_invalid_residue_reports_reference_pair(engine, oracle):
    import sequencealigner_amd as sa

    for x, y in [("acgtACGT", "ACGTACGT"), ("ACGT", "ACzT"), ("ACGJ", "ACGT"), ("AJGT", "AqGT")]:
        o = oracle.nw(x, y, "dna")
        assert o.rc == -2
        with pytest.raises(sa.SeqAlignError) as ei:
            engine.nw(x, y, "dna")
        assert ei.value.code == sa.engine.SA_ERR_INVALID_CODE
        assert (ord(ei.value.bad[0]), ord(ei.value.bad[1])) == o.bad
    # LCS accepts any byte (raw equality, main.cpp:1577)
    g = engine.lcs("acgtNNxx", "acgNNtxx")
    o = oracle.lcs("acgtNNxx", "acgNNtxx")
    assert g.score == o.score and np.array_equal(g.ops, o.ops)


def test_medium_dna_20k(engine_both, oracle):
    engine = engine_both
    rng = np.random.default_rng(23)
    x = rand_seq(rng, 20000, DNA)
    y = mutate(rng, x, DNA)
    engine.set_tuning(0, 0)
    o = oracle.nw(x, y, "dna", packed=True)
    g = engine.nw(x, y, "dna")
    assert g.score == o.score and np.array_equal(g.ops, o.ops)
    ol = oracle.lcs(x, y, packed=True)
    gl = engine.lcs(x, y)
    assert gl.score == ol.score and np.array_equal(gl.ops, ol.ops)


@pytest.mark.parametrize("go,ge", [(-5.5, -0.5), (-4.25, -1.75), (-0.5, -0.5), (-7.9, -0.1), (-5.0, -0.5), (-5.5, -1.0)])
@pytest.mark.parametrize("kind", ["nw", "sw"])
def test_fractional_gap_penalties(engine, oracle, kind, go, ge):
    """Quirk Q3: int + double truncated per operation (main.cpp:603); the GPU emulates the
    truncation in integer arithmetic."""
    rng = np.random.default_rng(29)
    x = rand_seq(rng, 900, DNA)
    y = mutate(rng, x, DNA, 0.1, 0.05, 0.05)
    check(kind, engine, oracle, x, y, "dna", go, ge)
    xp = rand_seq(rng, 333, PROT20)
    yp = mutate(rng, xp, PROT20, 0.3, 0.05, 0.05)
    check(kind, engine, oracle, xp, yp, "protein", go, ge)
    check(kind, engine, oracle, rand_seq(rng, 200, PROT24), rand_seq(rng, 300, PROT24), "protein", go, ge)


def test_fractional_gaps_take_the_persistent_pipeline(engine, oracle):
    """NW with -5.5 / -0.5 is one persistent launch for every rows-per-lane setting (the FLAG_FRAC instances), SW needs
    no special instance at all (E and F are floored at 0, so a fractional penalty acts like its floor)."""
    rng = np.random.default_rng(31)
    x = rand_seq(rng, 3000, DNA)
    y = mutate(rng, x, DNA, 0.1, 0.05, 0.05)
    engine.set_option("persistent", 1)
    for K in (1, 2, 4, 8, 0):
        engine.set_tuning(K, 0)
        for kind in ("nw", "sw"):
            g = check(kind, engine, oracle, x, y, "dna", -5.5, -0.5)
            assert g.n_launches == 1 and (K == 0 or g.rows_per_lane == K)
    engine.set_tuning(0, 0)
    engine.set_option("checkpoint_band", 2)
    try:
        g = check("nw", engine, oracle, x, y, "dna", -4.25, -1.75)
        assert g.n_launches > 1
    finally:
        engine.set_option("checkpoint_band", 0)


def test_fractional_gap_golden_scores(engine):
    import json
    from pathlib import Path

    cases = json.loads((Path(__file__).parent / "golden" / "random_pairs.json").read_text())
    n = 0
    for c in cases:
        if c["go"] is None or float(c["go"]).is_integer() and float(c["ge"]).is_integer():
            continue
        import re
        want = int(re.search(r'"score":\s*(-?\d+)', c["files"]["global_stats.json"]).group(1))
        assert engine.nw(c["x"], c["y"], c["mode"], c["go"], c["ge"]).score == want, c["id"]
        want = int(re.search(r'"score":\s*(-?\d+)', c["files"]["local_stats.json"]).group(1))
        assert engine.sw(c["x"], c["y"], c["mode"], c["go"], c["ge"]).score == want, c["id"]
        n += 1
    assert n >= 8


@pytest.mark.parametrize("kind", ["nw", "sw", "lcs"])
def test_keep_dp_matrix(engine, oracle, kind):
    """SA_FLAG_KEEP_DP: the full S / L matrix the reference dumps with --binary/--txt."""
    rng = np.random.default_rng(31)
    for (m, n, mode, go, ge) in [(1, 1, "dna", -5.0, -1.0), (37, 91, "dna", -5.0, -1.0), (300, 257, "protein", -5.0, -1.0),
                                 (130, 64, "dna", -5.5, -0.5), (0, 5, "dna", -5.0, -1.0), (5, 0, "dna", -5.0, -1.0)]:
        alpha = DNA if mode == "dna" else PROT20
        x, y = rand_seq(rng, m, alpha), rand_seq(rng, n, alpha)
        if kind == "lcs":
            g, o = engine.lcs(x, y, flags=1), oracle.lcs(x, y, keep_dp=True)
        else:
            g = getattr(engine, kind)(x, y, mode, go, ge, flags=1)
            o = getattr(oracle, kind)(x, y, mode, go, ge, keep_dp=True)
        assert g.score == o.score and np.array_equal(g.ops, o.ops)
        assert g.dp is not None and g.dp.shape == (m + 1, n + 1)
        assert np.array_equal(g.dp, o.dp), f"{kind} {m}x{n}"
