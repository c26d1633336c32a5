"""Helpers shared by the golden generators (tests/golden/make_golden*.py) and the tests that rebuild their inputs."""


def write_fasta(path, header, seq, width=70):
    with open(path, "w") as f:
        f.write(">" + header + "\n")
        for i in range(0, len(seq), width):
            f.write(seq[i:i + width] + "\n")


def tall_pair(m, n, seed, related):
    """A tall matrix (many rows, few columns) for SW: x = m uniform ACGT bases; y = either a ~90 %-identity copy of
    the n-base stretch in the middle of x (related: the local alignment sits in the middle rows) or n independent
    bases.  m = 70 000 rows at one row per lane is 1 094 strips, more than the 1 024 waves a persistent launch holds."""
    import numpy as np

    from sequencealigner_amd.synth import dna_pair

    acgt = np.frombuffer(b"ACGT", dtype=np.uint8)
    rng = np.random.default_rng(seed)
    x = acgt[rng.integers(0, 4, size=m)].copy()
    if not related:
        return x, acgt[rng.integers(0, 4, size=n)].copy()
    xs, ys = dna_pair(n, seed=seed + 1)
    lo = (m - n) // 2
    x[lo:lo + n] = xs
    return x, ys
