"""the seed_aligner's segment kernels.  align_segment_globally
(main_fmindex.cpp:1049-1143) and compute_lcs_for_segment (:1145-1207) are the recurrences and
single-pointer tracebacks of globalalign / lcs with another score table; perform_sw_in_window
(:938-1047) has the fill, floors and first-strict-maximum rule of localalign; its state-machine
traceback, for negative gap-open penalties, is the ungapped diagonal through the best cell while
S > 0.  Here: the oracle, fed with the table the reference's
own score functions return, against the outputs of those functions (fixture made by
tests/golden/make_golden_seed.py from the compiled reference)."""
import numpy as np
import pytest

from seed_common import load, replay

MODES = ["dna", "protein"]


@pytest.fixture(scope="module")
def gold():
    return load()


def test_segment_nw_equals_reference_function(oracle, gold):
    for c in gold["nw"]:
        oracle.set_custom_table(gold["tables"][MODES[c["mode"]]]["score"])
        x, y = c["s1"].encode("latin-1"), c["s2"].encode("latin-1")
        r = oracle.nw(x, y, "custom", c["go"], c["ge"])
        if len(x) == 0 and len(y) == 0:
            assert c["score"] == 0 and c["a1"] == ""
            continue
        if len(x) and len(y):  # (with an empty side the reference returns go + (k-1) ge as a double-to-int sum)
            assert r.score == c["score"], c
        assert replay(x, y, r.ops) == (c["a1"], c["a2"]), c


def test_segment_lcs_equals_reference_function(oracle, gold):
    for c in gold["lcs"]:
        x, y = c["s1"].encode("latin-1"), c["s2"].encode("latin-1")
        r = oracle.lcs(x, y)
        assert r.score == c["len"]
        if len(x) == 0 and len(y) == 0:
            continue
        g1, g2 = replay(x, y, r.ops)
        assert (g1, g2) == (c["g1"], c["g2"]), c
        diag = np.asarray(r.ops) == 0
        assert int(diag.sum()) == c["len"] == len(c["lcs"])


def test_window_sw_score_and_end_cell_equal_reference_function(oracle, gold):
    for c in gold["sw"]:
        oracle.set_custom_table(gold["tables"][MODES[c["mode"]]]["score"])
        x, y = c["s1"].encode("latin-1"), c["s2"].encode("latin-1")
        if not x or not y:
            assert c["score"] == 0
            continue
        r = oracle.sw(x, y, "custom", c["go"], c["ge"])
        assert r.score == c["score"], c
        if c["score"] > 0:
            assert (c["q_off"] + r.end[0] - 1, c["t_off"] + r.end[1] - 1) == (c["q_end"], c["t_end"]), c


def test_window_sw_alignment_is_the_ungapped_diagonal(oracle, gold):
    """What sa_sw_window_batch returns, restated on the oracle's matrix: from the best cell, diagonal
    steps while S > 0 -- equal to the reference function's strings and start coordinates."""
    seen = 0
    for c in gold["sw"]:
        if c["score"] <= 0:
            continue
        assert c["go"] < 0
        oracle.set_custom_table(gold["tables"][MODES[c["mode"]]]["score"])
        x, y = c["s1"].encode("latin-1"), c["s2"].encode("latin-1")
        r = oracle.sw(x, y, "custom", c["go"], c["ge"], keep_dp=True)
        i, j = r.end
        n = 0
        while i - n > 0 and j - n > 0 and r.dp[i - n, j - n] > 0:
            n += 1
        assert (c["s1"][i - n:i], c["s2"][j - n:j]) == (c["a1"], c["a2"]), c
        assert (c["q_off"] + i - n, c["t_off"] + j - n) == (c["q_start"], c["t_start"]), c
        seen += 1
    assert seen > 100
