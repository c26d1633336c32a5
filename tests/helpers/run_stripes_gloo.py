"""Launched by tests/test_stripes_gloo.py under torch.distributed.run (gloo, CPU): every rank
runs StripedAligner with the oracle-backed engine; rank 0 checks the stitched result against the
single-matrix oracle."""
import os
import sys
from pathlib import Path

ROOT = Path(__file__).resolve().parent.parent.parent
sys.path.insert(0, str(ROOT))
sys.path.insert(0, str(ROOT / "tests"))
sys.path.insert(0, str(ROOT / "tests" / "helpers"))

import numpy as np
import torch.distributed as dist

from oracle import Oracle
from oracle_stripe_engine import OracleStripeEngine
from sequencealigner_amd.stripes import StripedAligner
from sequencealigner_amd.synth import dna_pair


def main():
    dist.init_process_group("gloo")
    rank, world = dist.get_rank(), dist.get_world_size()
    al = StripedAligner(0, rank, world, dist, engine=OracleStripeEngine(), chunk_rows=97, min_stripe_cols=8)
    orc = Oracle()
    cases = []
    for n, seed, rel in [(700, 1, True), (1500, 2, True), (900, 3, False), (40, 4, True), (333, 5, True)]:
        x, y = dna_pair(n, seed=seed, related=rel)
        cases.append((x, y))
    cases.append((np.frombuffer(b"A" * 300, np.uint8), np.frombuffer(b"A" * 257, np.uint8)))
    cases.append((np.frombuffer(b"ACGT" * 100, np.uint8), np.frombuffer(b"TTTT" * 50 + b"ACGT" * 60, np.uint8)))
    rng = np.random.default_rng(7)
    noise = lambda k: rng.choice(np.frombuffer(b"ACGT", np.uint8), k)  # noqa: E731
    core = noise(400)
    cases.append((np.concatenate([noise(150), core, noise(90)]), np.concatenate([noise(300), core, noise(500)])))  # spans stripes
    cases.append((np.concatenate([noise(50), core[:60], noise(50)]), np.concatenate([noise(700), core[:60], noise(40)])))  # inside the last
    cases.append((np.frombuffer(b"A" * 200, np.uint8), np.frombuffer(b"C" * 300, np.uint8)))  # score 0
    ok = True
    for x, y in cases:
        for kind in ("nw", "lcs", "sw"):
            for go, ge in ((-5.0, -1.0), (-2.0, -2.0), (-5.5, -0.5)):
                r = al.align(kind, x, y, "dna", go, ge)
                want = orc.lcs(x, y) if kind == "lcs" else getattr(orc, kind)(x, y, "dna", go, ge)
                good = r.score == want.score and (rank != 0 or np.array_equal(r.ops, want.ops))
                if kind == "sw":  # every rank learns the end cell and the 'X' cell
                    good = good and tuple(r.end) == tuple(want.end) and tuple(r.start) == tuple(want.start)
                if not good:
                    print(f"rank {rank}: MISMATCH {kind} m={len(x)} n={len(y)} go={go}: {r.score} vs {want.score}", flush=True)
                ok = ok and good
    # An invalid residue in the LAST stripe's columns only: every rank must raise the reference's error (same pair)
    # before any transfer is posted -- nobody hangs -- and the next alignment works.
    from sequencealigner_amd.engine import SeqAlignError
    from sequencealigner_amd.stripes import StripeError

    x, y = dna_pair(600, seed=11)
    yb = y.copy()
    yb[len(yb) - 3] = ord("z")
    try:
        al.align("nw", x, yb, "dna")
        ok = False
        print(f"rank {rank}: no error for an invalid residue", flush=True)
    except SeqAlignError as e:
        good = e.code == -2 and e.bad == (chr(x[0]), "z")
        if not good:
            print(f"rank {rank}: wrong error {e} {e.bad}", flush=True)
        ok = ok and good
    # One rank's stripe fill fails before it posts anything (as an allocation failure would): its neighbours have
    # posted every chunk; the failed rank posts the matching transfers, all ranks vote and all raise StripeError.
    for failing in range(world):
        al.engine.fail_next = rank == failing
        try:
            al.align("nw", x, y, "dna")
            ok = False
            print(f"rank {rank}: no StripeError when rank {failing} fails", flush=True)
        except StripeError:
            pass
    r = al.align("nw", x, y, "dna")
    ok = ok and r.score == orc.nw(x, y, "dna").score
    dist.barrier()
    dist.destroy_process_group()
    print(f"rank {rank}: {'STRIPES_OK' if ok else 'STRIPES_FAIL'}", flush=True)
    sys.exit(0 if ok else 1)


if __name__ == "__main__":
    main()
