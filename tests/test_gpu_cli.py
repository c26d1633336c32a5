"""GPU: the `aligner` CLI end to end against the reference's own output files (golden)."""
import json
import subprocess
from pathlib import Path

import pytest

from test_oracle_golden import GLOBAL_FILES, GOLD, LCS_FILES, LOCAL_FILES, check_files

pytestmark = pytest.mark.gpu
PKG = Path(__file__).resolve().parent.parent / "sequencealigner_amd"


def test_cli_bundled_pairs_byte_exact(tmp_path):
    for c in json.loads((GOLD / "bundled.json").read_text()):
        out = tmp_path / c["id"]
        r = subprocess.run([str(PKG / "aligner"), "--query", str(GOLD / "inputs" / c["query"]), "--target",
                            str(GOLD / "inputs" / c["target"]), "--choice", "4", "--mode", c["mode"], "--outdir", str(out)],
                           capture_output=True, text=True, timeout=120)
        assert r.returncode == 0 and r.stderr == "", r.stderr
        check_files(out, c["files"], GLOBAL_FILES + LOCAL_FILES + LCS_FILES)


def test_cli_invalid_residue_like_reference(tmp_path):
    q, t = tmp_path / "q.fa", tmp_path / "t.fa"
    q.write_text(">q\nacgtACGT\n")
    t.write_text(">t\nACGTACGT\n")
    r = subprocess.run([str(PKG / "aligner"), "--query", str(q), "--target", str(t), "--choice", "4", "--outdir",
                        str(tmp_path / "o")], capture_output=True, text=True, timeout=60)
    assert r.returncode == 0 and r.stderr == "Error: Invalid DNA code: a,A\n"
    assert not list((tmp_path / "o").glob("*"))


def test_cli_big_dna3_dna4(tmp_path):
    c = {x["id"]: x for x in json.loads((GOLD / "big.json").read_text())}["dna3_dna4"]
    r = subprocess.run([str(PKG / "aligner"), "--query", str(GOLD / "inputs" / c["query"]), "--target",
                        str(GOLD / "inputs" / c["target"]), "--choice", "1", "--mode", "dna", "--outdir", str(tmp_path)],
                       capture_output=True, text=True, timeout=300)
    assert r.returncode == 0 and r.stderr == "", r.stderr
    check_files(tmp_path, c["files"], GLOBAL_FILES)


def test_cli_dp_dumps_match_reference_layout(tmp_path):
    """--txt / --binary dumps (main.cpp:422-528): compared against files built from the oracle's
    matrices through the same writers is circular, so the format is pinned by hand here and
    the values by the oracle."""
    import numpy as np

    from oracle import Oracle

    q, t = tmp_path / "q.fa", tmp_path / "t.fa"
    x, y = "ACGTTGCAAGCT", "ACGTGCATAGT"
    q.write_text(">q (GQ)\n" + x + "\n")
    t.write_text(">t\n" + y + "\n")
    out = tmp_path / "o"
    r = subprocess.run([str(PKG / "aligner"), "--query", str(q), "--target", str(t), "--choice", "4", "--outdir", str(out),
                        "--txt"], capture_output=True, text=True, timeout=60)
    assert r.returncode == 0 and r.stderr == ""
    orc = Oracle()
    o = orc.nw(x, y, "dna", keep_dp=True)
    want = "".join(" ".join(f"{v:5d}" for v in row) + "\n" for row in o.dp)
    assert (out / "global_dp_matrix.txt").read_text() == want
    o = orc.sw(x, y, "dna", keep_dp=True)
    want = "".join(" ".join(f"{v:5d}" for v in row) + "\n" for row in o.dp)
    assert (out / "local_dp_matrix.txt").read_text() == want
    o = orc.lcs(x, y, keep_dp=True)
    want = "".join(" ".join(f"{v:5d}" for v in row) + "\n" for row in o.dp)
    assert (out / "lcs_dp_lengths.txt").read_text() == want
    want = "".join(" ".join(ch.decode() for ch in row) + "\n" for row in o.ptr)
    assert (out / "lcs_traceback_pointers.txt").read_text() == want
    out2 = tmp_path / "o2"
    r = subprocess.run([str(PKG / "aligner"), "--query", str(q), "--target", str(t), "--choice", "1", "--outdir", str(out2),
                        "--txt", "--binary"], capture_output=True, text=True, timeout=60)
    assert r.returncode == 0 and not (out2 / "global_dp_matrix.txt").exists()
    raw = (out2 / "global_dp_matrix.bin").read_bytes()
    rows, cols = np.frombuffer(raw[:8], np.int32)
    assert (rows, cols) == (len(x) + 1, len(y) + 1)
    assert np.array_equal(np.frombuffer(raw[8:], np.int32).reshape(rows, cols), orc.nw(x, y, "dna", keep_dp=True).dp)
