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
    for go, ge in [(-10.0, -1.0), (-2.0, -2.0), (0.0, 0.0), (-1.0, -3.0), (-3.0, 0.0), (-5.5, -0.5), (-4.25, -1.75), (-0.5, -0.5)]:
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


# ---- sa_sw_batch_align: full local alignments per pair ----
def compare_align(engine, oracle, pool, ox, lx, oy, ly, mode="protein", go=-5.0, ge=-1.0):
    b = engine.sw_batch_align(pool, ox, lx, oy, ly, mode, go, ge)
    assert b.rc == 0
    for k in range(len(ox)):
        x = pool[ox[k]:ox[k] + lx[k]]
        y = pool[oy[k]:oy[k] + ly[k]]
        if lx[k] == 0 or ly[k] == 0:
            assert b.score[k] == 0 and b.n_ops[k] == 0
            continue
        w = oracle.sw(x, y, mode, go, ge)
        got = (int(b.score[k]), (int(b.end_i[k]), int(b.end_j[k])), (int(b.start_i[k]), int(b.start_j[k])))
        assert got == (w.score, tuple(w.end), tuple(w.start)), (k, lx[k], ly[k], got, (w.score, w.end, w.start))
        assert np.array_equal(b.ops(k), w.ops), (k, lx[k], ly[k])
    return b


def test_batch_align_all_row_groups(engine, oracle):
    rng = np.random.default_rng(11)
    lens = [(m, int(rng.integers(1, 500))) for m in list(range(1, 70, 3)) + list(range(120, 1030, 61)) + [63, 64, 65, 127, 128, 129, 1023, 1024]]
    compare_align(engine, oracle, *make_pool(rng, lens))
    compare_align(engine, oracle, *make_pool(rng, lens, PROT24, related=True))


def test_batch_align_config5_shape_and_chunks(engine, oracle):
    from sequencealigner_amd.synth import protein_pairs

    pool, ox, lx, oy, ly = protein_pairs(200, seed=42)
    b1 = compare_align(engine, oracle, pool, ox, lx, oy, ly)
    # a tiny code budget cuts the same batch into many chunks: identical results
    engine.set_option("batch_trace_budget_mb", 1)
    try:
        b2 = engine.sw_batch_align(pool, ox, lx, oy, ly)
    finally:
        engine.set_option("batch_trace_budget_mb", 32768)
    for f in ("score", "end_i", "end_j", "start_i", "start_j", "n_ops", "ops_off"):
        assert np.array_equal(getattr(b1, f), getattr(b2, f)), f
    for k in range(len(ox)):
        assert np.array_equal(b1.ops(k), b2.ops(k))


def test_batch_align_gaps_dna_ties_and_edges(engine, oracle):
    rng = np.random.default_rng(12)
    lens = [(int(rng.integers(1, 300)), int(rng.integers(1, 300))) for _ in range(40)]
    # fractional penalties (the reference's int + double truncation, main.cpp:958-964): for SW they act like their floors
    for go, ge in [(-10.0, -1.0), (-2.0, -2.0), (0.0, 0.0), (-1.0, -3.0), (-5.5, -0.5), (-4.25, -1.75), (-0.5, -0.5), (-7.9, -0.1)]:
        compare_align(engine, oracle, *make_pool(rng, lens, related=True), go=go, ge=ge)
    compare_align(engine, oracle, *make_pool(rng, lens, "ACGT", related=True), mode="dna")
    compare_align(engine, oracle, *make_pool(rng, [(30, 30)] * 10 + [(200, 90)] * 3, "A"), mode="dna")  # tie storms
    # no positive cell; empty sequences; a pair beyond the batch kernel's limits (single-pair path)
    pool = np.frombuffer(b"AAAAAAAA" + b"CCCCCCCC" + b"ACGT" * 300 + b"ACGTT" * 260, np.uint8)
    ox = np.array([0, 0, 0, 16])
    lx = np.array([8, 0, 8, 1200], np.int32)
    oy = np.array([8, 8, 8, 16 + 1200])
    ly = np.array([8, 8, 0, 1300], np.int32)
    compare_align(engine, oracle, pool, ox, lx, oy, ly, mode="dna")


def test_batch_align_invalid_residue(engine, oracle):
    pool = np.frombuffer(b"ACGTACGT" + b"ACGTxCGT" + b"ACGTACGA", np.uint8)
    b = engine.sw_batch_align(pool, np.array([0, 0]), np.array([8, 8], np.int32), np.array([8, 16]), np.array([8, 8], np.int32), "dna")
    assert b.rc == -2
    assert b.score[0] == np.iinfo(np.int32).min and b.n_ops[0] == 0
    w = oracle.sw(pool[0:8], pool[16:24], "dna")
    assert b.score[1] == w.score and np.array_equal(b.ops(1), w.ops)


# ---- sa_nw_batch_align: global alignments per pair ----
def compare_nw_align(engine, oracle, pool, ox, lx, oy, ly, mode="protein", go=-5.0, ge=-1.0):
    b = engine.nw_batch_align(pool, ox, lx, oy, ly, mode, go, ge)
    assert b.rc == 0
    for k in range(len(ox)):
        x = pool[ox[k]:ox[k] + lx[k]]
        y = pool[oy[k]:oy[k] + ly[k]]
        w = oracle.nw(x, y, mode, go, ge)
        assert int(b.score[k]) == w.score, (k, lx[k], ly[k], int(b.score[k]), w.score)
        assert np.array_equal(b.ops(k), w.ops), (k, lx[k], ly[k])
    return b


def test_nw_batch_align_all_row_groups(engine, oracle):
    rng = np.random.default_rng(21)
    lens = [(m, int(rng.integers(1, 500))) for m in list(range(1, 70, 3)) + list(range(120, 1030, 61)) + [63, 64, 65, 127, 128, 129, 1023, 1024]]
    compare_nw_align(engine, oracle, *make_pool(rng, lens))
    compare_nw_align(engine, oracle, *make_pool(rng, lens, PROT24, related=True))
    # short targets (every sub-block masked), targets just around the 64-step boundaries
    lens = [(int(rng.integers(1, 400)), n) for n in (1, 2, 15, 16, 17, 47, 48, 49, 63, 64, 65, 79, 80, 81, 127, 128, 129, 143, 144, 145)]
    compare_nw_align(engine, oracle, *make_pool(rng, lens, related=True))


def test_nw_batch_align_gaps_dna_ties_and_edges(engine, oracle):
    rng = np.random.default_rng(22)
    lens = [(int(rng.integers(1, 300)), int(rng.integers(1, 300))) for _ in range(40)]
    # fractional penalties: the FRAC instances of the kernel (sign-dependent truncation, main.cpp:603-604, 625-634)
    for go, ge in [(-10.0, -1.0), (-2.0, -2.0), (0.0, 0.0), (-1.0, -3.0), (-3.0, 0.0), (-5.5, -0.5), (-4.25, -1.75), (-0.5, -0.5), (-7.9, -0.1),
                   (-5.0, -0.5), (-5.5, -1.0)]:
        compare_nw_align(engine, oracle, *make_pool(rng, lens, related=True), go=go, ge=ge)
    lens2 = [(m, int(rng.integers(1, 500))) for m in (1, 63, 64, 65, 300, 640, 1000, 1024)]
    compare_nw_align(engine, oracle, *make_pool(rng, lens2, related=True), go=-5.5, ge=-0.5)
    compare_nw_align(engine, oracle, *make_pool(rng, lens, "ACGT", related=True), mode="dna")
    compare_nw_align(engine, oracle, *make_pool(rng, [(30, 30)] * 10 + [(200, 90)] * 3, "A"), mode="dna")  # tie storms
    # empty sides, both empty, and a pair beyond the batch kernel's limits (single-pair path)
    pool = np.frombuffer(b"ACGTACGT" + b"ACGT" * 300 + b"ACGTT" * 260, np.uint8)
    ox = np.array([0, 0, 0, 8])
    lx = np.array([8, 0, 0, 1200], np.int32)
    oy = np.array([0, 0, 0, 8 + 1200])
    ly = np.array([0, 5, 0, 1300], np.int32)
    compare_nw_align(engine, oracle, pool, ox, lx, oy, ly, mode="dna")


def test_nw_batch_align_seed_segments(engine):
    """the seed_aligner's anchored segments (tests/golden/seed_segments.json, from the compiled
    reference's align_segment_globally) in ONE call per score table and gap setting"""
    from seed_common import load, replay, scoring_arrays

    gold = load()
    modes = ["dna", "protein"]
    for mode in (0, 1):
        sub, idx = scoring_arrays(gold["tables"][modes[mode]])
        for go, ge in sorted({(c["go"], c["ge"]) for c in gold["nw"]}):
            cases = [c for c in gold["nw"] if c["mode"] == mode and (c["go"], c["ge"]) == (go, ge) and (c["s1"] or c["s2"])]
            if not cases:
                continue
            pool, ox, lx, oy, ly, pos = [], [], [], [], [], 0
            for c in cases:
                for s, o, l in ((c["s1"], ox, lx), (c["s2"], oy, ly)):
                    o.append(pos)
                    l.append(len(s))
                    pool.append(s)
                    pos += len(s)
            buf = np.frombuffer(("".join(pool) or "A").encode("latin-1"), np.uint8)
            b = engine.nw_batch_align(buf, np.array(ox), np.array(lx, np.int32), np.array(oy), np.array(ly, np.int32), modes[mode], go, ge,
                                      sub=sub, idx256=idx)
            assert b.rc == 0
            for k, c in enumerate(cases):
                if c["s1"] and c["s2"]:
                    assert int(b.score[k]) == c["score"], c
                assert replay(c["s1"].encode("latin-1"), c["s2"].encode("latin-1"), b.ops(k)) == (c["a1"], c["a2"]), c


# ---- sa_lcs_batch_align ----
def test_lcs_batch_align(engine, oracle):
    rng = np.random.default_rng(31)
    lens = [(m, int(rng.integers(1, 500))) for m in list(range(1, 70, 3)) + list(range(120, 1030, 61)) + [63, 64, 65, 127, 128, 129, 1023, 1024]]
    lens += [(int(rng.integers(1, 400)), n) for n in (1, 2, 15, 16, 17, 63, 64, 65, 127, 128, 129)]
    for alphabet, related in (("ACGT", True), (PROT20, True), ("AB", False), ("A", False)):
        pool, ox, lx, oy, ly = make_pool(rng, lens, alphabet, related=related)
        b = engine.lcs_batch_align(pool, ox, lx, oy, ly)
        for k in range(len(ox)):
            w = oracle.lcs(pool[ox[k]:ox[k] + lx[k]], pool[oy[k]:oy[k] + ly[k]])
            assert int(b.score[k]) == w.score, (alphabet, k, lx[k], ly[k])
            assert np.array_equal(b.ops(k), w.ops), (alphabet, k, lx[k], ly[k])
    # empty sides and a pair beyond the batch kernel's limits
    pool = np.frombuffer(b"ACGTACGT" + b"ACGT" * 300 + b"ACGTT" * 260, np.uint8)
    ox, lx = np.array([0, 0, 0, 8]), np.array([8, 0, 0, 1200], np.int32)
    oy, ly = np.array([0, 0, 0, 8 + 1200]), np.array([0, 5, 0, 1300], np.int32)
    b = engine.lcs_batch_align(pool, ox, lx, oy, ly)
    for k in range(4):
        w = oracle.lcs(pool[ox[k]:ox[k] + lx[k]], pool[oy[k]:oy[k] + ly[k]])
        assert int(b.score[k]) == w.score and np.array_equal(b.ops(k), w.ops), k


def test_lcs_batch_align_seed_segments(engine):
    from seed_common import load, replay

    cases = [c for c in load()["lcs"] if c["s1"] or c["s2"]]
    pool, ox, lx, oy, ly, pos = [], [], [], [], [], 0
    for c in cases:
        for s, o, l in ((c["s1"], ox, lx), (c["s2"], oy, ly)):
            o.append(pos)
            l.append(len(s))
            pool.append(s)
            pos += len(s)
    buf = np.frombuffer("".join(pool).encode("latin-1"), np.uint8)
    b = engine.lcs_batch_align(buf, np.array(ox), np.array(lx, np.int32), np.array(oy), np.array(ly, np.int32))
    for k, c in enumerate(cases):
        assert int(b.score[k]) == c["len"], c
        assert replay(c["s1"].encode("latin-1"), c["s2"].encode("latin-1"), b.ops(k)) == (c["g1"], c["g2"]), c


# ---- sa_batch_set_device_pool: sequences already on the GPU ----
def test_device_resident_pool_gives_identical_results(engine):
    import ctypes as C

    import sequencealigner_amd as sa

    rng = np.random.default_rng(31)
    lens = [(int(rng.integers(1, 900)), int(rng.integers(1, 900))) for _ in range(80)] + [(1100, 500), (0, 40), (40, 0), (64, 64)]
    pool, ox, lx, oy, ly = make_pool(rng, lens, related=True)
    host = {
        "sw": engine.sw_batch(pool, ox, lx, oy, ly, "protein"),
        "swa": engine.sw_batch_align(pool, ox, lx, oy, ly, "protein"),
        "nwa": engine.nw_batch_align(pool, ox, lx, oy, ly, "protein"),
        "lcs": engine.lcs_batch_align(pool, ox, lx, oy, ly),
    }
    # device memory through the HIP runtime the library itself is linked to (torch must not be imported AFTER the
    # library has initialised HIP: INTEGRATION.md, "Other hosts")
    hip = C.CDLL("libamdhip64.so.7")
    dev = C.c_void_p()
    assert hip.hipMalloc(C.byref(dev), C.c_size_t(len(pool))) == 0
    assert hip.hipMemcpy(dev, C.c_void_p(pool.ctypes.data), C.c_size_t(len(pool)), 1) == 0  # hipMemcpyHostToDevice
    engine.set_device_pool(dev.value, len(pool))
    try:
        sc, ei, ej, _ms, rc = engine.sw_batch(None, ox, lx, oy, ly, "protein")
        assert rc == 0 and all(np.array_equal(a, b) for a, b in zip((sc, ei, ej), host["sw"][:3]))
        for key, call in (("swa", lambda: engine.sw_batch_align(None, ox, lx, oy, ly, "protein")),
                          ("nwa", lambda: engine.nw_batch_align(None, ox, lx, oy, ly, "protein")),
                          ("lcs", lambda: engine.lcs_batch_align(None, ox, lx, oy, ly))):
            b, h = call(), host[key]
            for f in ("score", "end_i", "end_j", "start_i", "start_j", "n_ops", "ops_off"):
                assert np.array_equal(getattr(b, f), getattr(h, f)), (key, f)
            for k in range(len(ox)):
                assert np.array_equal(b.ops(k), h.ops(k)), (key, k)
        # a pair that ends beyond the pool is refused
        with pytest.raises(sa.engine.SeqAlignError):
            engine.sw_batch(None, ox + 10 ** 9, lx, oy, ly, "protein")
    finally:
        engine.set_device_pool(None, 0)
        hip.hipFree(dev)
    sc2 = engine.sw_batch(pool, ox, lx, oy, ly, "protein")[0]
    assert np.array_equal(sc2, host["sw"][0])
    with pytest.raises(sa.engine.SeqAlignError):  # a pool needs a size
        engine.set_device_pool(1234, 0)


def test_batch_penalty_limit_is_the_documented_one(engine):
    """include/seqalign_hip.h (sa_sw_batch): penalties beyond 5121 x (|penalty| + 1) <= 1.2e8 are refused with
    SA_ERR_ARG by the many-pair entry points -- -23 431 passes, -23 432 does not -- while sa_sw takes such a pair."""
    from sequencealigner_amd.engine import SA_ERR_ARG, SeqAlignError
    from sequencealigner_amd.synth import protein_pairs

    pool, ox, lx, oy, ly = protein_pairs(8, seed=3, lo=50, hi=200)
    ok = engine.sw_batch(pool, ox, lx, oy, ly, "protein", gap_open=-23431.0, gap_extend=-1.0)
    assert ok[4] == 0
    with pytest.raises(SeqAlignError) as e:
        engine.sw_batch(pool, ox, lx, oy, ly, "protein", gap_open=-23432.0, gap_extend=-1.0)
    assert e.value.code == SA_ERR_ARG and "overflow" in str(e.value)
    with pytest.raises(SeqAlignError):
        engine.nw_batch_align(pool, ox, lx, oy, ly, "protein", gap_open=-25000.0, gap_extend=-1.0)
    one = engine.sw(pool[ox[0]:ox[0] + lx[0]], pool[oy[0]:oy[0] + ly[0]], "protein", -25000.0, -1.0)
    assert one.score == ok[0][0]  # a penalty that large or larger: both are the best ungapped local alignment
