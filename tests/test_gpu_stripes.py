"""GPU: the stripe kernels (border in / border out / stop-at-left-edge traceback) on one device.
The ranks of a striped run are replayed one after the other in a single process, so the border
of stripe g is complete before stripe g+1 starts; the exchange protocol itself is covered by
tests/test_stripes_gloo.py.  The stitched result must equal the single-GPU result bit for bit."""
import numpy as np
import pytest

from conftest import DNA, PROT20, mutate, rand_seq

pytestmark = pytest.mark.gpu


class DevBuf:
    """Plain hipMalloc'd border buffer (no torch in this test: the HIP runtime the engine loaded
    is used directly)."""
    hip = None

    def __init__(self, nbytes):
        import ctypes as C

        if DevBuf.hip is None:
            DevBuf.hip = C.CDLL("libamdhip64.so")
            DevBuf.hip.hipMalloc.argtypes = [C.POINTER(C.c_void_p), C.c_size_t]
            DevBuf.hip.hipMemset.argtypes = [C.c_void_p, C.c_int, C.c_size_t]
            DevBuf.hip.hipFree.argtypes = [C.c_void_p]
        self.p = C.c_void_p()
        assert DevBuf.hip.hipMalloc(C.byref(self.p), nbytes) == 0
        assert DevBuf.hip.hipMemset(self.p, 0, nbytes) == 0

    def data_ptr(self):
        return self.p.value

    def free(self):
        DevBuf.hip.hipFree(self.p)


class DevStreams:
    """two HIP streams for the overlapped stripe schedule (stream_in / stream_out), synchronised between the replayed ranks"""

    def __init__(self):
        import ctypes as C

        DevBuf(8).free()  # loads libamdhip64
        self.hip = DevBuf.hip
        self.hip.hipStreamCreate.argtypes = [C.POINTER(C.c_void_p)]
        self.hip.hipStreamSynchronize.argtypes = [C.c_void_p]
        self.hip.hipStreamDestroy.argtypes = [C.c_void_p]
        self.s = [C.c_void_p(), C.c_void_p()]
        for q in self.s:
            assert self.hip.hipStreamCreate(C.byref(q)) == 0

    def sync(self):
        for q in self.s:
            assert self.hip.hipStreamSynchronize(q) == 0

    def close(self):
        for q in self.s:
            self.hip.hipStreamDestroy(q)


def striped(engine, kind, x, y, G, mode, go, ge, chunk_rows, streams=None, options=None):
    xb = np.frombuffer(x.encode(), np.uint8)
    yb = np.frombuffer(y.encode(), np.uint8)
    m, n = len(xb), len(yb)
    bounds = [n * k // G for k in range(G + 1)]
    events = []
    borders = [None] + [DevBuf((m + 1) * 8) for _ in range(G - 1)] + [None]
    score = None
    # the trace of stripe g must survive until its traceback: one engine per stripe
    import sequencealigner_amd as sa

    engs = [engine] + [sa.Engine(0) for _ in range(G - 1)]
    for e in engs:
        for name, value in (options or {}).items():
            e.set_option(name, value)
    for g in range(G):
        left, right = borders[g], borders[g + 1]
        r = engs[g].stripe_fill(kind, xb, yb[bounds[g]:bounds[g + 1]], n, bounds[g], g == 0, g == G - 1,
                                None if left is None else left.data_ptr(), None if right is None else right.data_ptr(),
                                chunk_rows, lambda ev, k, g=g: events.append((g, ev, k)), mode, go, ge,
                                **({"stream_in": streams.s[0].value, "stream_out": streams.s[1].value} if streams else {}))
        if streams:
            streams.sync()  # the border conversions / signals of this rank are done before the next rank is replayed
        striped.last_rows_per_lane = r.rows_per_lane
        if g == G - 1:
            score = r.score
    ops = []
    si = m
    for g in range(G - 1, -1, -1):
        tb = engs[g].stripe_traceback(si, bounds[g + 1] - bounds[g])
        ops.append(tb.ops)
        si = tb.start[0]
        assert tb.start[1] == 0
    assert si == 0
    for e in engs[1:]:
        e.close()
    for b in borders:
        if b is not None:
            b.free()
    return score, np.concatenate(ops), events


@pytest.mark.parametrize("kind", ["nw", "lcs"])
def test_stripes_equal_single_gpu(engine, oracle, kind):
    rng = np.random.default_rng(41)
    for (m, G, K, C, R) in [(700, 2, 1, 64, 100), (3000, 3, 2, 128, 512), (5000, 4, 4, 64, 1000), (2500, 8, 8, 256, 4096)]:
        x = rand_seq(rng, m, DNA)
        y = mutate(rng, x, DNA)
        for e in (engine,):
            e.set_tuning(K, C)
        want = oracle.lcs(x, y) if kind == "lcs" else oracle.nw(x, y, "dna")
        import sequencealigner_amd as sa

        score, ops, events = striped(engine, kind, x, y, G, "dna", -5.0, -1.0, R)
        assert score == want.score, (kind, m, G)
        assert np.array_equal(ops, want.ops), (kind, m, G)
        n_chunks = (m + R - 1) // R
        for g in range(G):
            need = [k for (gg, ev, k) in events if gg == g and ev == 0]
            ready = [k for (gg, ev, k) in events if gg == g and ev == 1]
            assert need == (list(range(n_chunks)) if g > 0 else [])
            assert ready == (list(range(n_chunks)) if g < G - 1 else [])
    engine.set_tuning(0, 0)


def lcs_bits_stripe_cases(engine, oracle):
    """Overlapped schedule (transfer streams given) with border chunks of whole 2048-row strips: LCS stripes run
    lcs_bits_persistent with the border column converted between (H, E) rows and vertical-difference bits on the
    transfer streams; same score and op list as the oracle.  Ragged row counts, 2..5 stripes, related and unrelated."""
    rng = np.random.default_rng(43)
    st = DevStreams()
    try:
        for (m, n, G, R, rel) in [(5000, 5200, 2, 2048, True), (9001, 7000, 3, 4096, True), (4097, 9000, 5, 2048, False),
                                  (20000, 20000, 4, 8192, True), (2048, 3000, 2, 2048, False), (100, 700, 2, 2048, True)]:
            if rel:
                x = rand_seq(rng, m, DNA)
                y = mutate(rng, x, DNA)[:n]
            else:
                x, y = rand_seq(rng, m, DNA), rand_seq(rng, n, DNA)
            want = oracle.lcs(x, y, packed=True)
            score, ops, _ = striped(engine, "lcs", x, y, G, "dna", -5.0, -1.0, R, streams=st)
            assert striped.last_rows_per_lane == 32, "the bit-vector kernel did not run"
            assert score == want.score, (m, n, G, R)
            assert np.array_equal(ops, want.ops), (m, n, G, R)
        # chunks that are not whole strips, or the option off: the cell-by-cell kernels, same answer
        x = rand_seq(rng, 6000, DNA)
        y = mutate(rng, x, DNA)
        want = oracle.lcs(x, y, packed=True)
        score, ops, _ = striped(engine, "lcs", x, y, 3, "dna", -5.0, -1.0, 1024, streams=st)
        assert striped.last_rows_per_lane != 32, striped.last_rows_per_lane
        assert score == want.score, (score, want.score)
        assert np.array_equal(ops, want.ops)
        score, ops, _ = striped(engine, "lcs", x, y, 3, "dna", -5.0, -1.0, 2048, streams=st, options={"lcs_bits": 0})
        assert striped.last_rows_per_lane != 32, striped.last_rows_per_lane
        assert score == want.score, (score, want.score)
        assert np.array_equal(ops, want.ops)
    finally:
        engine.set_option("lcs_bits", 1)
        st.close()


def test_lcs_stripes_run_the_bit_vector_kernel():
    """lcs_bits_stripe_cases in a process of its own with 16 hardware queues (what bench.py sets for N > 1): the
    transfer streams must not share a hardware queue with the stream of the persistent kernel that waits for them."""
    import os
    import subprocess
    import sys
    from pathlib import Path

    root = Path(__file__).resolve().parent.parent
    env = dict(os.environ, GPU_MAX_HW_QUEUES="16")
    p = subprocess.run([sys.executable, str(root / "tests" / "helpers" / "run_lcs_bits_stripes.py")], capture_output=True, text=True, timeout=600, env=env)
    assert p.returncode == 0 and "LCS_BITS_STRIPES_OK" in p.stdout, p.stdout[-3000:] + p.stderr[-3000:]


def striped_sw(engine, x, y, G, mode, go, ge, chunk_rows):
    """SW over G stripes replayed on one GPU: per-stripe best -> winner by (score desc, row asc,
    column asc) -> traceback from the winner leftwards until the 'X' cell."""
    import sequencealigner_amd as sa

    xb = np.frombuffer(x.encode(), np.uint8)
    yb = np.frombuffer(y.encode(), np.uint8)
    m, n = len(xb), len(yb)
    bounds = [n * k // G for k in range(G + 1)]
    borders = [None] + [DevBuf((m + 1) * 8) for _ in range(G - 1)] + [None]
    engs = [engine] + [sa.Engine(0) for _ in range(G - 1)]
    cands = []
    for g in range(G):
        left, right = borders[g], borders[g + 1]
        r = engs[g].stripe_fill("sw", xb, yb[bounds[g]:bounds[g + 1]], n, bounds[g], g == 0, g == G - 1,
                                None if left is None else left.data_ptr(), None if right is None else right.data_ptr(),
                                chunk_rows, lambda ev, k: None, mode, go, ge)
        cands.append((-r.score, r.end[0], bounds[g] + r.end[1], g))
    neg, bi, bj, owner = min(cands)
    score, ops, start = -neg, [], (0, 0)
    if score > 0:
        row, col = bi, bj - bounds[owner]
        for g in range(owner, -1, -1):
            tb = engs[g].stripe_traceback(row, col)
            ops.append(tb.ops)
            start = (tb.start[0], bounds[g] + tb.start[1])
            if not (g > 0 and tb.start[1] == 0 and tb.start[0] > 0):
                break
            row, col = tb.start[0], bounds[g] - bounds[g - 1]
    else:
        bi = bj = 0
    for e in engs[1:]:
        e.close()
    for b in borders:
        if b is not None:
            b.free()
    return score, (bi, bj), start, (np.concatenate(ops) if ops else np.zeros(0, np.uint8))


def test_sw_stripes_equal_oracle(engine, oracle):
    rng = np.random.default_rng(47)
    core = rand_seq(rng, 900, DNA)
    cases = []
    for (m, G, K, C, R) in [(700, 2, 1, 64, 128), (3000, 3, 2, 128, 512), (5000, 4, 4, 64, 1024), (2500, 8, 8, 256, 512)]:
        x = rand_seq(rng, m, DNA)
        cases.append((x, mutate(rng, x, DNA), G, K, C, R))
    # a local hit that spans stripe boundaries inside unrelated flanks; one inside a single stripe; none
    cases.append((rand_seq(rng, 400, DNA) + core + rand_seq(rng, 300, DNA), rand_seq(rng, 1500, DNA) + mutate(rng, core, DNA) + rand_seq(rng, 1200, DNA), 4, 2, 64, 512))
    cases.append((rand_seq(rng, 300, DNA) + core[:80] + rand_seq(rng, 100, DNA), rand_seq(rng, 2500, DNA) + core[:80] + rand_seq(rng, 60, DNA), 3, 1, 64, 128))
    cases.append(("A" * 600, "C" * 900, 3, 2, 64, 128))
    for persistent in (1, 0):
        engine.set_option("persistent", persistent)
        for (x, y, G, K, C, R) in cases:
            engine.set_tuning(K, C)
            want = oracle.sw(x, y, "dna")
            score, end, start, ops = striped_sw(engine, x, y, G, "dna", -5.0, -1.0, R)
            assert score == want.score, (len(x), G, persistent)
            assert tuple(end) == tuple(want.end) and tuple(start) == tuple(want.start), (len(x), G, end, want.end, start, want.start)
            assert np.array_equal(ops, want.ops), (len(x), G, persistent)
    engine.set_tuning(0, 0)
    engine.set_option("persistent", 1)


def test_stripes_protein_and_gaps(engine, oracle):
    rng = np.random.default_rng(43)
    x = rand_seq(rng, 1800, PROT20)
    y = mutate(rng, x, PROT20, 0.2, 0.05, 0.05)
    for go, ge in ((-5.0, -1.0), (-11.0, -1.0), (-2.0, -2.0), (-5.5, -0.5), (-4.25, -1.75)):
        want = oracle.nw(x, y, "protein", go, ge)
        score, ops, _ = striped(engine, "nw", x, y, 3, "protein", go, ge, 256)
        assert score == want.score and np.array_equal(ops, want.ops)


@pytest.mark.parametrize("world,queues", [(2, 16), (3, 16), (2, 1)])
def test_striped_aligner_processes_share_one_gpu(world, queues):
    """The real stripe engine (HIP kernels, torch stream and border tensors, event callbacks,
    traceback chain, op gather) in `world` processes on GPU 0; gloo carries the chunks."""
    import os
    import subprocess
    import sys
    from pathlib import Path

    root = Path(__file__).resolve().parent.parent
    port = 29700 + world + 7 * queues + (os.getpid() % 200)
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={world}", "--master-addr", "127.0.0.1",
           "--master-port", str(port), str(root / "tests" / "helpers" / "run_stripes_gpu_gloo.py")]
    # queues=16: every stream gets a hardware queue of its own (what bench.py sets for N > 1);
    # queues=1: the transfers are serialised behind the persistent kernel, whose bounded waits must
    # then expire on every rank and the aligner must fall back to the in-line schedule
    env = dict(os.environ, OMP_NUM_THREADS="2", GPU_MAX_HW_QUEUES=str(queues), EXPECT_FALLBACK="1" if queues == 1 else "0")
    p = subprocess.run(cmd, capture_output=True, text=True, timeout=600, env=env)
    assert p.returncode == 0, p.stdout[-3000:] + p.stderr[-3000:]
    assert p.stdout.count("GPU_STRIPES_OK") == world, p.stdout[-3000:]
    if queues == 1:
        assert "falling back" in p.stdout
