"""GPU: sa_sw_batch_noalloc (the one entry point that allocates nothing: its device scratch is a file-scope
static __device__ array) against sa_sw_batch through the C-ABI, bit-exact (integer scores and cells), on two
calls in a row, the second with a smaller input -- the array keeps the first call's contents."""
import numpy as np
import pytest

from sequencealigner_amd.synth import protein_pairs

pytestmark = pytest.mark.gpu


def _both(engine, pool, ox, lx, oy, ly, **kw):
    a = engine.sw_batch(pool, ox, lx, oy, ly, "protein", **kw)
    b = engine.sw_batch_noalloc(pool, ox, lx, oy, ly, "protein", **kw)
    assert a[4] == b[4]
    for k in range(3):
        assert np.array_equal(a[k], b[k]), k
    return b


def test_noalloc_equals_sw_batch_twice_second_smaller(engine, oracle):
    pool, ox, lx, oy, ly = protein_pairs(3000, seed=7, lo=1, hi=1000)
    lx[5] = 0  # an empty query: score 0, cell (0, 0), no device slot written
    first = _both(engine, pool, ox, lx, oy, ly)
    assert first[0].max() > 0
    # second call, smaller input, different pairs: nothing of the first call may show through
    pool2, ox2, lx2, oy2, ly2 = protein_pairs(257, seed=8, lo=30, hi=400)
    second = _both(engine, pool2, ox2, lx2, oy2, ly2)
    rc, sc, ei, ej = oracle.sw_score_batch(pool2, ox2, lx2, oy2, ly2, "protein", threads=8)
    assert rc == 0 and np.array_equal(sc, second[0]) and np.array_equal(ei, second[1]) and np.array_equal(ej, second[2])
    # and a third one with an invalid residue: same error protocol as sa_sw_batch, flag cleared afterwards
    pool3 = pool2.copy()
    pool3[ox2[3] + 1] = ord("J")
    third = _both(engine, pool3, ox2, lx2, oy2, ly2)
    assert third[4] == -2 and third[0][3] == np.iinfo(np.int32).min
    again = _both(engine, pool2, ox2, lx2, oy2, ly2)
    assert again[4] == 0 and np.array_equal(again[0], second[0])


def test_noalloc_allocates_no_device_memory(engine):
    """The property the entry point exists for: the free device memory is the same before and after its calls.  The
    ctx's cached workspaces are released first and no allocating entry point runs in between, so a hipMalloc / ensure()
    that crept back into run_batch_noalloc would show (hipMemGetInfo through the runtime the library is linked against)."""
    import ctypes as C

    hip = C.CDLL("libamdhip64.so")

    def free_bytes():
        f, t = C.c_size_t(), C.c_size_t()
        assert hip.hipMemGetInfo(C.byref(f), C.byref(t)) == 0
        return f.value

    pool, ox, lx, oy, ly = protein_pairs(2000, seed=17, lo=20, hi=900)
    want = engine.sw_batch(pool, ox, lx, oy, ly, "protein")
    engine.release_workspace()
    first = engine.sw_batch_noalloc(pool, ox, lx, oy, ly, "protein")  # anything lazily set up by the runtime happens here
    before = free_bytes()
    second = engine.sw_batch_noalloc(pool, ox, lx, oy, ly, "protein")
    third = engine.sw_batch_noalloc(pool[: ox[500]], ox[:500], lx[:500], oy[:500], ly[:500], "protein")
    assert free_bytes() == before
    for k in range(3):
        assert np.array_equal(want[k], first[k]) and np.array_equal(want[k], second[k]) and np.array_equal(want[k][:500], third[k])


def test_noalloc_overlapping_calls_are_refused_not_interleaved(engine):
    """All calls of the process on one GPU share that GPU's scratch array: a call that arrives while another one holds
    it gets SA_ERR_ARG ("scratch in use") instead of overwriting its descriptors.  Two ctxs on GPU 0, two threads
    (ctypes releases the GIL during the call): every call either returns the right results or that refusal."""
    import threading

    import sequencealigner_amd as sa
    from sequencealigner_amd.engine import SA_ERR_ARG, SeqAlignError

    pool, ox, lx, oy, ly = protein_pairs(6000, seed=21, lo=300, hi=1000)
    want = engine.sw_batch(pool, ox, lx, oy, ly, "protein")
    other = sa.Engine(0)
    stats = {"ok": 0, "refused": 0, "bad": []}
    lock = threading.Lock()

    def worker(eng):
        for _ in range(12):
            try:
                got = eng.sw_batch_noalloc(pool, ox, lx, oy, ly, "protein")
                good = all(np.array_equal(want[k], got[k]) for k in range(3))
                with lock:
                    stats["ok"] += 1
                    if not good:
                        stats["bad"].append("results differ")
            except SeqAlignError as e:
                with lock:
                    if e.code == SA_ERR_ARG and "scratch in use" in str(e):
                        stats["refused"] += 1
                    else:
                        stats["bad"].append(str(e))

    try:
        ts = [threading.Thread(target=worker, args=(e,)) for e in (engine, other)]
        for t in ts:
            t.start()
        for t in ts:
            t.join()
    finally:
        other.close()
    assert not stats["bad"], stats
    assert stats["ok"] >= 2 and stats["ok"] + stats["refused"] == 24, stats
    # and the flag is free again afterwards
    got = engine.sw_batch_noalloc(pool, ox, lx, oy, ly, "protein")
    assert all(np.array_equal(want[k], got[k]) for k in range(3))


def test_noalloc_reads_a_device_pool_in_place(engine):
    import ctypes as C

    pool, ox, lx, oy, ly = protein_pairs(500, seed=9, lo=100, hi=700)
    want = engine.sw_batch(pool, ox, lx, oy, ly, "protein")
    # the TEST allocates the device pool (through the HIP runtime the library itself is linked against: already
    # loaded, so this is the same instance); the entry point under test reads it in place
    hip = C.CDLL("libamdhip64.so")
    dev = C.c_void_p()
    assert hip.hipMalloc(C.byref(dev), C.c_size_t(len(pool))) == 0
    try:
        assert hip.hipMemcpy(dev, C.c_void_p(pool.ctypes.data), C.c_size_t(len(pool)), 1) == 0  # hipMemcpyHostToDevice
        engine.set_device_pool(dev.value, len(pool))
        try:
            got = engine.sw_batch_noalloc(None, ox, lx, oy, ly, "protein")
        finally:
            engine.set_device_pool(None, 0)
    finally:
        hip.hipFree(dev)
    for k in range(3):
        assert np.array_equal(want[k], got[k])


def test_noalloc_rejects_what_does_not_fit(engine):
    from sequencealigner_amd.engine import SA_ERR_ARG, SA_ERR_UNSUPPORTED, SeqAlignError

    n = 16384 + 1
    z = np.zeros(n, np.int64)
    one = np.ones(n, np.int32)
    with pytest.raises(SeqAlignError) as e:
        engine.sw_batch_noalloc(np.frombuffer(b"AR", np.uint8), z, one, z + 1, one, "protein")
    assert e.value.code == SA_ERR_ARG and "SA_NOALLOC_MAX_PAIRS" in str(e.value)
    big = np.full((32 << 20) + 2000, ord("A"), np.uint8)
    with pytest.raises(SeqAlignError) as e:
        engine.sw_batch_noalloc(big, [0], [500], [len(big) - 600], [600], "protein")
    assert e.value.code == SA_ERR_ARG and "SA_NOALLOC_MAX_SEQ_BYTES" in str(e.value)
    with pytest.raises(SeqAlignError) as e:
        engine.sw_batch_noalloc(big, [0], [1025], [2000], [600], "protein")
    assert e.value.code == SA_ERR_UNSUPPORTED
