"""Helpers shared by the golden generators (tests/golden/make_golden*.py) and the tests that rebuild their inputs."""


def write_fasta(path, header, seq, width=70):
    with open(path, "w") as f:
        f.write(">" + header + "\n")
        for i in range(0, len(seq), width):
            f.write(seq[i:i + width] + "\n")
