"""GPU: Smith-Waterman at size against the reference ITSELF (tests/golden/big_sw.json: the unmodified reference ran
--choice 2 on 20 000 / 30 000 pairs and on two 70 000-row matrices in the build container).  What the small SW tests
cannot reach: launches with more strips than resident waves (second round of tickets, ring-slot reuse and record
generations with the per-row best reduction on top), forced here three ways -- one row per lane on the 70 000-row
matrices (1 094 strips for 1 024 waves), a workgroup cap (option "max_workgroups") that makes every pair take many
rounds, and the checkpointed schedule.  Files byte for byte (main.cpp:1099-1128, 1182-1265)."""
import json
import subprocess
import tempfile
from pathlib import Path

import numpy as np
import pytest

from sequencealigner_amd import host
from test_oracle_golden import GOLD, LOCAL_FILES, check_files, sw_big_pair

pytestmark = pytest.mark.gpu
PKG = Path(__file__).resolve().parent.parent / "sequencealigner_amd"
CASES = ["synth_20000_1234_c2", "synth_30000_77_c2", "unrel_20000_5678_c2", "tall_70000x5000_99_c2", "tall_70000x4000_98_c2"]


def _case(cid):
    return {c["id"]: c for c in json.loads((GOLD / "big_sw.json").read_text())}[cid]


def _check(c, x, y, r):
    with tempfile.TemporaryDirectory() as d:
        host.write_result("local", d, "seqA synth", "seqB synth", "dna", x, y, r.score, r.end, r.ops)
        check_files(d, c["files"], LOCAL_FILES)


@pytest.mark.parametrize("cid", CASES)
def test_sw_big_equals_reference_all_schedules(engine, cid):
    c = _case(cid)
    x, y = sw_big_pair(c)
    try:
        a = engine.sw(x, y, "dna")  # automatic rows per lane, one persistent launch
        assert a.n_launches == 1
        _check(c, x, y, a)
        for K, wgs in ((1, 0), (1, 16), (2, 5), (8, 3)):  # K = 1: most strips; few workgroups: many rounds per wave
            engine.set_tuning(K, 0)
            engine.set_option("max_workgroups", wgs)
            b = engine.sw(x, y, "dna")
            assert b.rows_per_lane == K and b.n_launches == 1
            strips = -(-len(x) // (64 * K))
            if cid.startswith("tall") and K == 1 and wgs == 0:
                assert strips > 1024  # more strips than the launch has waves
            assert (b.score, b.end, b.start) == (a.score, a.end, a.start), (K, wgs)
            assert np.array_equal(a.ops, b.ops), (K, wgs)
        engine.set_tuning(0, 0)
        engine.set_option("max_workgroups", 0)
        engine.set_option("checkpoint_band", 7)  # codes recomputed seven strips at a time
        b = engine.sw(x, y, "dna")
        assert b.n_launches > 1 and (b.score, b.end, b.start) == (a.score, a.end, a.start) and np.array_equal(a.ops, b.ops)
        engine.set_option("checkpoint_band", 0)
        engine.set_option("persistent", 0)  # one launch per tile diagonal
        b = engine.sw(x, y, "dna")
        assert b.n_launches > 1 and (b.score, b.end, b.start) == (a.score, a.end, a.start) and np.array_equal(a.ops, b.ops)
    finally:
        engine.set_option("checkpoint_band", 0)
        engine.set_option("max_workgroups", 0)


@pytest.mark.parametrize("cid", ["synth_30000_77_c2", "tall_70000x5000_99_c2"])
def test_cli_sw_big_equals_reference(tmp_path, cid):
    from make_golden_shared import write_fasta

    c = _case(cid)
    x, y = sw_big_pair(c)
    write_fasta(tmp_path / "q.fasta", "seqA synth", x.tobytes().decode())
    write_fasta(tmp_path / "t.fasta", "seqB synth", y.tobytes().decode())
    out = tmp_path / "o"
    r = subprocess.run([str(PKG / "aligner"), "--query", str(tmp_path / "q.fasta"), "--target", str(tmp_path / "t.fasta"),
                        "--choice", "2", "--mode", "dna", "--outdir", str(out)], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0 and r.stderr == "", r.stderr
    check_files(out, c["files"], LOCAL_FILES)


def test_nw_lcs_many_rounds_equal_oracle(engine, oracle):
    """The same workgroup cap on NW and LCS (cell kernels and the bit-vector kernel): a 9 000 x 7 000 pair swept by
    2 or 3 workgroups takes a dozen rounds of tickets per wave."""
    from sequencealigner_amd.synth import dna_pair

    x, y = dna_pair(9000, seed=31)
    y = y[:7000]
    want_nw, want_lcs = oracle.nw(x, y, "dna"), oracle.lcs(x, y)
    try:
        for K, wgs in ((1, 2), (2, 3), (4, 1)):
            engine.set_tuning(K, 0)
            engine.set_option("max_workgroups", wgs)
            a = engine.nw(x, y, "dna")
            assert a.score == want_nw.score and np.array_equal(a.ops, want_nw.ops), (K, wgs)
            b = engine.lcs(x, y)
            assert b.score == want_lcs.score and np.array_equal(b.ops, want_lcs.ops), (K, wgs)
        engine.set_tuning(0, 0)
        engine.set_option("max_workgroups", 1)  # bit-vector LCS: 5 strips of 2048 rows on 2 waves
        b = engine.lcs(x, y)
        assert b.rows_per_lane == 32 and b.score == want_lcs.score and np.array_equal(b.ops, want_lcs.ops)
    finally:
        engine.set_option("max_workgroups", 0)
