"""GPU, SURVEY 8f-3 (partial): the seed_aligner's segment kernels through the C-ABI with the
seed_aligner's own score tables passed as a custom sa_scoring -- against the outputs of the
reference's functions (tests/golden/seed_segments.json, made from the compiled reference):
anchored-segment NW and segment LCS with their complete gapped strings, windowed SW with score, end cell
and the reference's alignment (one sa_sw_batch / sa_sw_window_batch call per table for all windows)."""
import numpy as np
import pytest

from seed_common import load, replay, scoring_arrays

pytestmark = pytest.mark.gpu
MODES = ["dna", "protein"]


@pytest.fixture(scope="module")
def gold():
    return load()


def test_segment_nw(engine, gold):
    tabs = [scoring_arrays(gold["tables"][m]) for m in MODES]
    for c in gold["nw"]:
        x, y = c["s1"].encode("latin-1"), c["s2"].encode("latin-1")
        if not x and not y:
            continue
        sub, idx = tabs[c["mode"]]
        r = engine.nw(x, y, MODES[c["mode"]], c["go"], c["ge"], sub=sub, idx256=idx)
        if x and y:
            assert r.score == c["score"], c
        assert replay(x, y, r.ops) == (c["a1"], c["a2"]), c


def test_segment_lcs(engine, gold):
    for c in gold["lcs"]:
        x, y = c["s1"].encode("latin-1"), c["s2"].encode("latin-1")
        if not x and not y:
            continue
        r = engine.lcs(x, y)
        assert r.score == c["len"]
        assert replay(x, y, r.ops) == (c["g1"], c["g2"]), c


def test_window_sw_batched(engine, gold):
    for mode in (0, 1):
        sub, idx = scoring_arrays(gold["tables"][MODES[mode]])
        for go, ge in sorted({(c["go"], c["ge"]) for c in gold["sw"]}):
            cases = [c for c in gold["sw"] if c["mode"] == mode and (c["go"], c["ge"]) == (go, ge) and c["s1"] and c["s2"]]
            if not cases:
                continue
            pool, ox, lx, oy, ly, pos = [], [], [], [], [], 0
            for c in cases:
                for s, o, l in ((c["s1"], ox, lx), (c["s2"], oy, ly)):
                    o.append(pos)
                    l.append(len(s))
                    pool.append(s)
                    pos += len(s)
            buf = np.frombuffer("".join(pool).encode("latin-1"), np.uint8)
            sc, ei, ej, _ms, rc = engine.sw_batch(buf, np.array(ox), np.array(lx, np.int32), np.array(oy), np.array(ly, np.int32),
                                                  MODES[mode], go, ge, sub=sub, idx256=idx)
            assert rc == 0
            for k, c in enumerate(cases):
                assert sc[k] == c["score"], c
                if c["score"] > 0:
                    assert (c["q_off"] + ei[k] - 1, c["t_off"] + ej[k] - 1) == (c["q_end"], c["t_end"]), c


def test_window_sw_alignment(engine, gold):
    """perform_sw_in_window's complete result per window: score, end, start and the two aligned strings."""
    seen = 0
    for mode in (0, 1):
        sub, idx = scoring_arrays(gold["tables"][MODES[mode]])
        for go, ge in sorted({(c["go"], c["ge"]) for c in gold["sw"]}):
            cases = [c for c in gold["sw"] if c["mode"] == mode and (c["go"], c["ge"]) == (go, ge)]
            if not cases:
                continue
            pool, ox, lx, oy, ly, pos = [], [], [], [], [], 0
            for c in cases:
                for s, o, l in ((c["s1"], ox, lx), (c["s2"], oy, ly)):
                    o.append(pos)
                    l.append(len(s))
                    pool.append(s)
                    pos += len(s)
            buf = np.frombuffer(("".join(pool) + "A").encode("latin-1"), np.uint8)
            sc, ei, ej, si, sj, ln, _ms, rc = engine.sw_window_batch(buf, np.array(ox), np.array(lx, np.int32), np.array(oy),
                                                                     np.array(ly, np.int32), MODES[mode], go, ge, sub=sub, idx256=idx)
            assert rc == 0
            for k, c in enumerate(cases):
                assert sc[k] == c["score"], c
                if c["score"] <= 0:
                    assert (ei[k], ej[k], si[k], sj[k], ln[k]) == (0, 0, 0, 0, 0), c
                    continue
                assert ei[k] - si[k] == ln[k] == ej[k] - sj[k]
                assert (c["q_off"] + ei[k] - 1, c["t_off"] + ej[k] - 1) == (c["q_end"], c["t_end"]), c
                assert (c["q_off"] + si[k], c["t_off"] + sj[k]) == (c["q_start"], c["t_start"]), c
                assert (c["s1"][si[k]:ei[k]], c["s2"][sj[k]:ej[k]]) == (c["a1"], c["a2"]), c
                seen += 1
    assert seen > 100


def test_window_sw_rejects_zero_gap_open(engine):
    from sequencealigner_amd.engine import SeqAlignError
    buf = np.frombuffer(b"ACGTACGT", np.uint8)
    with pytest.raises(SeqAlignError):
        engine.sw_window_batch(buf, [0], [4], [4], [4], "dna", 0.0, -1.0)
