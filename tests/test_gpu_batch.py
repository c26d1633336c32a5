"""GPU parity of the batched inter-task SW kernel (score + end cell per pair) vs the oracle."""
import numpy as np
import pytest

from conftest import PROT20, PROT24, rand_seq

pytestmark = pytest.mark.gpu


def make_pool(rng, lens, alphabet=PROT20, related=False):
    pool, ox, lx, oy, ly = [], [], [], [], []
    pos = 0
    for (m, n) in lens:
        x = rand_seq(rng, m, alphabet)
        y = rand_seq(rng, n, alphabet)
        if related and m > 20:
            cut = min(m, n) // 2
            y = y[:n - cut] + x[:cut] if n >= cut else y
            y = y[:n]
        for s, o, l in ((x, ox, lx), (y, oy, ly)):
            o.append(pos)
            l.append(len(s))
            pool.append(s)
            pos += len(s)
    return np.frombuffer("".join(pool).encode(), np.uint8), np.array(ox), np.array(lx, np.int32), np.array(oy), np.array(ly, np.int32)


def compare(engine, oracle, pool, ox, lx, oy, ly, mode="protein", go=-5.0, ge=-1.0):
    sc, ei, ej, ms, rc = engine.sw_batch(pool, ox, lx, oy, ly, mode, go, ge)
    assert rc == 0
    orc, osc, oei, oej = oracle.sw_score_batch(pool, ox, lx, oy, ly, mode, go, ge, threads=8)
    assert orc == 0
    bad = np.nonzero((sc != osc) | (ei != oei) | (ej != oej))[0]
    assert len(bad) == 0, f"first mismatch pair {bad[0]}: m={lx[bad[0]]} n={ly[bad[0]]} gpu=({sc[bad[0]]},{ei[bad[0]]},{ej[bad[0]]}) ref=({osc[bad[0]]},{oei[bad[0]]},{oej[bad[0]]})"
    return ms


def test_batch_all_row_groups(engine, oracle):
    rng = np.random.default_rng(1)
    lens = [(m, int(rng.integers(1, 700))) for m in list(range(1, 70)) + list(range(120, 1030, 37)) + [63, 64, 65, 127, 128, 129, 1023, 1024]]
    compare(engine, oracle, *make_pool(rng, lens))
    compare(engine, oracle, *make_pool(rng, lens, PROT24, related=True))


def test_batch_config5_shape(engine, oracle):
    from sequencealigner_amd.synth import protein_pairs

    pool, ox, lx, oy, ly = protein_pairs(600, seed=42)
    compare(engine, oracle, pool, ox, lx, oy, ly)


def test_batch_gap_settings_and_dna(engine, oracle):
    rng = np.random.default_rng(2)
    lens = [(int(rng.integers(1, 400)), int(rng.integers(1, 400))) for _ in range(60)]
    for go, ge in [(-10.0, -1.0), (-2.0, -2.0), (0.0, 0.0), (-1.0, -3.0), (-3.0, 0.0)]:
        compare(engine, oracle, *make_pool(rng, lens, related=True), go=go, ge=ge)
    compare(engine, oracle, *make_pool(rng, lens, "ACGT", related=True), mode="dna")
    compare(engine, oracle, *make_pool(rng, [(30, 30)] * 20, "A"), mode="dna")  # tie storm: first row-major max


def test_batch_long_pairs_take_the_single_pair_path(engine, oracle):
    rng = np.random.default_rng(3)
    lens = [(1025, 300), (300, 4200), (2000, 2500), (10, 10), (0, 50), (50, 0)]
    compare(engine, oracle, *make_pool(rng, lens))


def test_batch_invalid_residue_flags_only_that_pair(engine, oracle):
    import sequencealigner_amd as sa

    rng = np.random.default_rng(4)
    pool, ox, lx, oy, ly = make_pool(rng, [(50, 60), (70, 80), (90, 100)])
    pool = pool.copy()
    pool[ox[1] + 5] = ord("J")
    sc, ei, ej, ms, rc = engine.sw_batch(pool, ox, lx, oy, ly, "protein")
    assert rc == sa.engine.SA_ERR_INVALID_CODE
    assert sc[1] == np.iinfo(np.int32).min
    for k in (0, 2):
        r = oracle.sw(pool[ox[k]:ox[k] + lx[k]], pool[oy[k]:oy[k] + ly[k]], "protein")
        assert (sc[k], ei[k], ej[k]) == (r.score, r.end[0], r.end[1])
