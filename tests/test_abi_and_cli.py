"""CPU tests of the boundary: every symbol the headers declare is exported, and the CLI's
argument handling / exit codes match the reference (main.cpp:1731-1791, 1828-1838)."""
import ctypes as C
import re
import subprocess
from pathlib import Path

import pytest

ROOT = Path(__file__).resolve().parent.parent
PKG = ROOT / "sequencealigner_amd"


def declared(header, prefix):
    text = (ROOT / "include" / header).read_text()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(" + prefix + r"[a-z0-9_]+)\s*\(", text)))


def test_hip_library_exports_every_declared_symbol():
    names = declared("seqalign_hip.h", "sa_")
    assert {"sa_create", "sa_nw", "sa_sw", "sa_lcs", "sa_sw_batch", "sa_free_result"} <= set(names)
    lib = C.CDLL(str(PKG / "libseqalign_hip.so"))
    for n in names:
        assert hasattr(lib, n), n


def test_host_library_exports_every_declared_symbol():
    names = declared("seqalign_host.h", "sah_")
    assert len(names) >= 7
    lib = C.CDLL(str(PKG / "libsa_host.so"))
    for n in names:
        assert hasattr(lib, n), n


def test_product_never_references_the_oracle():
    for p in list(PKG.rglob("*.py")) + list((PKG / "csrc").glob("*")) + list((ROOT / "include").glob("*")):
        if p.is_file() and p.suffix in (".py", ".h", ".hip", ".cpp", ""):
            try:
                t = p.read_text()
            except UnicodeDecodeError:
                continue
            assert "cpu_ref" not in t and "import oracle" not in t and "from oracle" not in t, p


def run_cli(*args):
    return subprocess.run([str(PKG / "aligner"), *args], capture_output=True, text=True, timeout=60)


def test_cli_exit_codes_and_messages(tmp_path):
    r = run_cli("--help")
    assert r.returncode == 0 and "--choice: 1=global, 2=local, 3=LCS, 4=all" in r.stdout
    r = run_cli("--bogus")
    assert r.returncode == 1 and r.stderr == "Unknown option: --bogus\n"
    r = run_cli("--query", "a.fa")
    assert r.returncode == 1 and r.stderr == "Missing required arguments: --query, --target, --choice\n"
    r = run_cli("--mode", "rna")
    assert r.returncode == 1 and r.stderr == "Unknown mode: rna\n"
    # a missing input file is an exception in the reference: message on stderr, exit code 0
    r = run_cli("--query", str(tmp_path / "nope.fa"), "--target", str(tmp_path / "nope2.fa"), "--choice", "1",
                "--outdir", str(tmp_path / "o"))
    assert r.returncode == 0 and r.stderr.startswith("Error: Error: Unable to open ")
    q = tmp_path / "q.fa"
    q.write_text(">q\nACGT\n")
    r = run_cli("--query", str(q), "--target", str(q), "--choice", "7", "--outdir", str(tmp_path / "o"))
    assert r.returncode == 0 and r.stderr == "Invalid method. Use --choice 1/2/3/4.\n"


def test_fasta_reader_quirks(tmp_path):
    """Quirk Q6: first header only, all records concatenated, blank lines skipped, CR kept."""
    from sequencealigner_amd import host

    p = tmp_path / "multi.fa"
    p.write_bytes(b">first rec (GENE1)\nACGT\n\nAC\r\n>second\nGG\n")
    h, s = host.read_fasta(p)
    assert h == "first rec (GENE1)" and s == b"ACGTAC\rGG"
    assert host.accession(h, "dna") == "first" and host.gene_symbol(h, "dna") == "GENE1"
    ph = "sp|P00533|EGFR_HUMAN Epidermal growth factor receptor OS=Homo sapiens GN=EGFR"
    assert host.accession(ph, "protein") == "P00533" and host.gene_symbol(ph, "protein") == "EGFR"
    assert host.accession("no pipes here", "protein") == "no pipes here"
    assert host.gene_symbol("x GN=ABC1;foo", "protein") == "ABC1"


def test_multi_fasta_records(tmp_path):
    from sequencealigner_amd.host import read_fasta_records

    f = tmp_path / "m.fa"
    f.write_text(">a one\nACGT\n\nAC\n>b\n>c|x|y\nTTT\r\nGG\n")
    recs = read_fasta_records(f)
    assert recs == [("a one", b"ACGTAC"), ("b", b""), ("c|x|y", b"TTT\rGG")]
    g = tmp_path / "nohdr.fa"
    g.write_text("ACGT\nAC\n>z\nG\n")
    assert read_fasta_records(g) == [("", b"ACGTAC"), ("z", b"G")]
    with pytest.raises(FileNotFoundError):
        read_fasta_records(tmp_path / "missing.fa")


def test_cli_under_mpirun_ranks_above_zero_do_nothing(tmp_path):
    """existing `mpirun -np P ./aligner ...` scripts keep working -- only rank 0 runs the GPU job."""
    import os

    q = tmp_path / "q.fa"
    q.write_text(">q\nACGT\n")
    out = tmp_path / "o"
    for var in ("OMPI_COMM_WORLD_RANK", "PMI_RANK"):
        env = dict(os.environ, **{var: "3"})
        r = subprocess.run([str(PKG / "aligner"), "--query", str(q), "--target", str(q), "--choice", "1", "--outdir", str(out)],
                           capture_output=True, text=True, timeout=60, env=env)
        assert r.returncode == 0 and r.stdout == "" and r.stderr == "" and not out.exists()
    env = dict(os.environ, OMPI_COMM_WORLD_RANK="1")  # argument errors stay per-rank, as in the reference
    r = subprocess.run([str(PKG / "aligner"), "--bogus"], capture_output=True, text=True, timeout=60, env=env)
    assert r.returncode == 1
