"""One long pair over several GPUs: column stripes, one process per GPU.

Rank g owns the columns [n*g/G, n*(g+1)/G) of the same DP matrix and runs the ordinary
strip/tile fill on them (sa_stripe_fill).  The only exchange is the stripe's border column,
(H, E) per matrix row = 8 bytes per row, sent rank g -> g+1 in chunks of `chunk_rows` rows with
torch.distributed send/recv (backend "nccl" = RCCL over xGMI on the GPU box, "gloo" in the CPU
tests): nearest-neighbour point-to-point only, no collective on the data path.  The traceback then
runs right to left, each rank walking its own codes and handing the exit row to its left
neighbour; rank 0 concatenates the op lists.

On the GPU the stripe is ONE persistent launch and the border crosses while it runs: receives are
enqueued on a receive stream, sends on a send stream, and the library brackets them with one-wave
signal / wait kernels (include/seqalign_hip.h, sa_stripe_desc.stream_in / stream_out).  The two links
of a rank (to its left and to its right neighbour) use two different process groups, so that the
transfers of one never queue behind the other inside the backend.

The engine is duck-typed (stripe_fill / stripe_traceback / alloc_border) so that the same
orchestration is exercised on CPU by the tests with an oracle-backed engine.
"""
from __future__ import annotations

import contextlib
import os

import numpy as np

from .engine import MODES, SA_ERR_INVALID_CODE, SA_EV_NEED_LEFT, AlignResult, SeqAlignError, load_library


def first_invalid_pair(x: np.ndarray, y: np.ndarray, mode: str):
    """The residue pair the reference's lazy check would throw on first, scanning the cells in its row-major
    fill order (main.cpp:640 inside :622, :738) -- the host-side rule of the library (seqalign_hip.hip,
    first_invalid_pair), evaluated here on the WHOLE x and y: every rank holds both, so every rank reaches the
    same verdict before any border transfer is posted.  None when every cell is valid."""
    sc = load_library().sa_builtin_scoring(MODES[mode]).contents
    idx = np.ctypeslib.as_array((np.ctypeslib.ctypes.c_uint8 * 256).from_address(sc.idx256))
    if len(x) == 0 or len(y) == 0:
        return None
    bad_y = np.flatnonzero(idx[y] == 255)
    bad_x = np.flatnonzero(idx[x] == 255)
    if len(bad_y):  # row 1 already meets it: at column 1 if x[0] itself is invalid, else at the first bad column
        return (chr(x[0]), chr(y[0])) if len(bad_x) and bad_x[0] == 0 else (chr(x[0]), chr(y[bad_y[0]]))
    if len(bad_x):
        return chr(x[bad_x[0]]), chr(y[0])
    return None


class GpuStripeEngine:
    """sa.Engine + torch CUDA tensors for the borders; kernels run on torch's current stream so
    that they are ordered with the RCCL send/recv torch enqueues."""

    def __init__(self, device_index: int):
        import torch

        from .engine import Engine

        self.torch = torch
        self.device = torch.device("cuda", device_index)
        torch.cuda.set_device(self.device)
        self.eng = Engine(device_index)
        # An explicit stream: torch's default stream has handle 0, which the C-ABI reads as "use
        # the ctx's own stream" -- and then the kernels would not be ordered with RCCL's send/recv.
        self.stream = torch.cuda.Stream(device=self.device)
        self.eng.set_stream(self.stream.cuda_stream)
        # border transfers overlap the running fill: receives on one stream, sends on another
        self.stream_in = torch.cuda.Stream(device=self.device)
        self.stream_out = torch.cuda.Stream(device=self.device)

    def transfer_context(self, incoming: bool):
        return self.torch.cuda.stream(self.stream_in if incoming else self.stream_out)

    def transfer_streams(self):
        return self.stream_in.cuda_stream, self.stream_out.cuda_stream

    def sync_transfers(self):
        self.stream_in.synchronize()
        self.stream_out.synchronize()

    def stream_context(self):
        return self.torch.cuda.stream(self.stream)

    def alloc_border(self, m):
        return self.torch.zeros((m + 1, 2), dtype=self.torch.int32, device=self.device)

    def small(self, values, dtype=None):
        return self.torch.tensor(values, dtype=dtype or self.torch.int64, device=self.device)

    def bytes_tensor(self, arr: np.ndarray):
        return self.torch.from_numpy(np.ascontiguousarray(arr)).to(self.device)

    def empty_bytes(self, n):
        return self.torch.empty(n, dtype=self.torch.uint8, device=self.device)

    def stripe_fill(self, *a, **kw):
        return self.eng.stripe_fill(*a, **kw)

    def stripe_traceback(self, i, j):
        return self.eng.stripe_traceback(i, j)

    def whole(self, kind, x, y, mode, go, ge):
        return self.eng.lcs(x, y) if kind == "lcs" else getattr(self.eng, kind)(x, y, mode, go, ge)


class StripeError(RuntimeError):
    """some rank's stripe fill failed (raised on every rank)"""


class StripedAligner:
    def __init__(self, local_rank, rank, world, dist, engine=None, chunk_rows=8192, min_stripe_cols=256, overlap=True):
        self.rank, self.world, self.dist = rank, world, dist
        self.engine = engine if engine is not None else GpuStripeEngine(local_rank)
        self.chunk_rows = int(chunk_rows)
        self.min_stripe_cols = int(min_stripe_cols)
        # overlap: persistent fill with the borders crossing on two transfer streams while it runs;
        # False: the launch-per-diagonal schedule with the transfers in line on the compute stream
        self.overlap = bool(overlap) and os.environ.get("SA_STRIPES_OVERLAP", "1") != "0"
        # link (g-1, g) uses group g & 1: a rank's two links never share a backend queue
        self.link_groups = [None, None]
        if world > 2 and hasattr(dist, "new_group"):
            self.link_groups = [dist.new_group(list(range(world))), dist.new_group(list(range(world)))]
        self._warm = False

    def _link(self, right_rank):
        """process group of the link whose right end is `right_rank`"""
        return self.link_groups[right_rank & 1]

    def _warm_links(self):
        """First use of a point-to-point link sets the connection up (for RCCL that can take
        seconds and synchronise the device): do it before a persistent kernel is waiting on it."""
        if self._warm:
            return
        self._warm = True
        E, dist, g, G = self.engine, self.dist, self.rank, self.world
        if not hasattr(E, "transfer_context"):
            return
        for incoming in (True, False):
            with E.transfer_context(incoming):
                t = E.small([g])
                grouped = self.link_groups[0] is not None
                if incoming and g > 0:
                    dist.recv(t, src=g - 1, **({"group": self._link(g)} if grouped else {}))
                if not incoming and g < G - 1:
                    dist.send(t, dst=g + 1, **({"group": self._link(g + 1)} if grouped else {}))
        E.sync_transfers()

    def align(self, kind, x, y, mode="dna", gap_open=-5.0, gap_extend=-1.0) -> AlignResult:
        """Every rank passes the same x and y; every rank returns the score, rank 0 also the
        complete op list (identical to the single-GPU sa_nw / sa_lcs result)."""
        if kind not in ("nw", "lcs", "sw"):
            raise ValueError(f"unknown kind {kind!r}")
        E, dist, g, G = self.engine, self.dist, self.rank, self.world
        x = np.frombuffer(x.encode("latin-1") if isinstance(x, str) else bytes(x), np.uint8) if not isinstance(x, np.ndarray) else x
        y = np.frombuffer(y.encode("latin-1") if isinstance(y, str) else bytes(y), np.uint8) if not isinstance(y, np.ndarray) else y
        m, n = len(x), len(y)
        if G == 1 or m == 0 or n < self.min_stripe_cols * G:
            return self._on_rank0(kind, x, y, mode, gap_open, gap_extend)
        # Go / no-go BEFORE any transfer is posted, identical on every rank: a stripe fill that refused its input on
        # one rank only (the bad residue sits in that rank's columns) would leave its neighbours' transfers unmatched.
        if kind != "lcs":
            bad = first_invalid_pair(x, y, mode)
            if bad is not None:
                raise SeqAlignError(SA_ERR_INVALID_CODE, f"invalid residue pair {bad[0]},{bad[1]}", bad)

        with (E.stream_context() if hasattr(E, "stream_context") else contextlib.nullcontext()):
            return self._align_striped(kind, x, y, m, n, mode, gap_open, gap_extend)

    def _align_striped(self, kind, x, y, m, n, mode, gap_open, gap_extend) -> AlignResult:
        E, dist, g, G = self.engine, self.dist, self.rank, self.world
        bounds = [n * k // G for k in range(G + 1)]
        c0, c1 = bounds[g], bounds[g + 1]
        ng = c1 - c0
        first, last = g == 0, g == G - 1
        left = None if first else E.alloc_border(m)
        right = None if last else E.alloc_border(m)
        R = self.chunk_rows
        sends = []
        overlapped = self.overlap and hasattr(E, "transfer_context")
        kw_in = {"group": self._link(g)} if self.link_groups[0] is not None else {}
        kw_out = {"group": self._link(g + 1)} if self.link_groups[0] is not None else {}

        n_chunks = (m + R - 1) // R
        posted = {True: 0, False: 0}  # chunks whose receive / send has been posted (both in ascending order)

        def on_event(event, k):
            lo, hi = k * R + 1, min(m, (k + 1) * R) + 1
            incoming = event == SA_EV_NEED_LEFT
            with (E.transfer_context(incoming) if overlapped else contextlib.nullcontext()):
                if incoming:
                    dist.recv(left[lo:hi], src=g - 1, **kw_in)
                else:
                    sends.append(dist.isend(right[lo:hi], dst=g + 1, **kw_out))
            posted[incoming] = k + 1

        extra = {}
        if overlapped:
            self._warm_links()
            E.sync_transfers()  # the border tensors were allocated on the compute stream
            E.stream.synchronize()
            extra["stream_in"], extra["stream_out"] = E.transfer_streams()
        failure = None
        try:
            res = E.stripe_fill(kind, x, y[c0:c1], n, c0, first, last,
                                None if first else left.data_ptr(), None if last else right.data_ptr(), R, on_event,
                                mode, gap_open, gap_extend, **extra)
        except RuntimeError as e:  # e.g. a bounded wait inside the persistent kernel expired, or no memory
            failure = e
            # This rank may have stopped before it posted (all of) its transfers, but its neighbours post every chunk:
            # post what is missing, so that every send meets its receive and everybody reaches the vote below.  What
            # travels is garbage; the vote discards the whole alignment.
            if not first:
                for k in range(posted[True], n_chunks):
                    on_event(SA_EV_NEED_LEFT, k)
            if not last:
                for k in range(posted[False], n_chunks):
                    on_event(SA_EV_NEED_LEFT + 1, k)
        with (E.transfer_context(False) if overlapped else contextlib.nullcontext()):
            for w in sends:
                w.wait()
        if overlapped:
            E.sync_transfers()
        # everybody learns whether every stripe is good before the traceback chain starts
        ok = E.small([0 if failure is not None else 1])
        dist.all_reduce(ok, op=dist.ReduceOp.MIN)
        if int(ok.item()) == 0:
            raise StripeError(f"rank {g}: {failure}" if failure is not None else f"rank {g}: another rank's stripe fill failed")

        if kind == "sw":
            return self._finish_sw(res, m, n, c0, ng, first, overlapped)

        # traceback, right to left: (exit row) travels g+1 -> g
        if last:
            si = m
        else:
            t = E.small([0])
            dist.recv(t, src=g + 1)
            si = int(t.item())
        tb = E.stripe_traceback(si, ng)
        if not first:
            dist.send(E.small([tb.start[0]]), dst=g - 1)

        # op lists to rank 0 (right-most stripe first), score from the last rank
        score = E.small([res.score if last else 0])
        dist.broadcast(score, src=G - 1)
        if g == 0:
            parts = {0: tb.ops}
            for src in range(1, G):
                ln = E.small([0])
                dist.recv(ln, src=src)
                buf = E.empty_bytes(int(ln.item()))
                if int(ln.item()):
                    dist.recv(buf, src=src)
                parts[src] = buf.cpu().numpy()
            ops = np.concatenate([parts[k] for k in range(G - 1, -1, -1)]) if parts else np.zeros(0, np.uint8)
        else:
            dist.send(E.small([len(tb.ops)]), dst=0)
            if len(tb.ops):
                dist.send(E.bytes_tensor(tb.ops), dst=0)
            ops = np.zeros(0, np.uint8)
        out = AlignResult(int(score.item()), (m, n), (0, 0), ops)
        out.fill_ms, out.kernel_fill_ms, out.traceback_ms = res.fill_ms, res.kernel_fill_ms, tb.traceback_ms
        out.n_launches, out.rows_per_lane, out.tile_steps = res.n_launches, res.rows_per_lane, res.tile_steps
        out.cells_trace, out.cells_notrace = getattr(res, "cells_trace", 0), getattr(res, "cells_notrace", 0)
        out.border_wait_ms = getattr(res, "border_wait_ms", 0.0)
        out.extra = {"stripe": (c0, c1), "stripe_ops": len(tb.ops), "overlapped": bool(overlapped)}
        return out

    def _gather_ops(self, my_ops):
        """op lists to rank 0, right-most stripe first"""
        E, dist, g, G = self.engine, self.dist, self.rank, self.world
        if g == 0:
            parts = {0: my_ops}
            for src in range(1, G):
                ln = E.small([0])
                dist.recv(ln, src=src)
                buf = E.empty_bytes(int(ln.item()))
                if int(ln.item()):
                    dist.recv(buf, src=src)
                parts[src] = buf.cpu().numpy()
            return np.concatenate([parts[k] for k in range(G - 1, -1, -1)])
        dist.send(E.small([len(my_ops)]), dst=0)
        if len(my_ops):
            dist.send(E.bytes_tensor(my_ops), dst=0)
        return np.zeros(0, np.uint8)

    def _finish_sw(self, res, m, n, c0, ng, first, overlapped=False) -> AlignResult:
        """SW over stripes: the reference's cross-rank arg-max (main.cpp:1151-1177) with its
        first-strict-row-major-maximum rule, then the traceback from the winning cell leftwards
        until the 'X' cell (main.cpp:1233)."""
        E, dist, g, G = self.engine, self.dist, self.rank, self.world
        INF = (1 << 62)
        best = E.small([int(res.score)])
        dist.all_reduce(best, op=dist.ReduceOp.MAX)
        score = int(best.item())
        if score <= 0:  # no positive cell anywhere: empty local alignment
            return AlignResult(0, (0, 0), (0, 0), np.zeros(0, np.uint8))
        # among the cells attaining it: smallest row, then smallest (global) column
        key = E.small([(int(res.end[0]) << 32) | (c0 + int(res.end[1])) if int(res.score) == score else INF])
        dist.all_reduce(key, op=dist.ReduceOp.MIN)
        bi, bj = int(key.item()) >> 32, int(key.item()) & 0xFFFFFFFF
        bounds = [n * k // G for k in range(G + 1)]
        owner = next(k for k in range(G) if bounds[k] < bj <= bounds[k + 1])
        ops = np.zeros(0, np.uint8)
        stop = [0, 0]  # the 'X' cell in global coordinates, contributed by the rank where the walk ended
        tb_ms = 0.0
        if g <= owner:
            if g == owner:
                row, col, done = bi, bj - c0, 0
            else:
                t = E.small([0, 0])
                dist.recv(t, src=g + 1)
                row, done = (int(v) for v in t.tolist())
                col = ng
            if not done:
                tb = E.stripe_traceback(row, col)
                ops, tb_ms = tb.ops, tb.traceback_ms
                goes_on = (not first) and tb.start[1] == 0 and tb.start[0] > 0
                if not goes_on:
                    stop = [int(tb.start[0]), c0 + int(tb.start[1])]
                row, done = int(tb.start[0]), 0 if goes_on else 1
            if not first:
                dist.send(E.small([row, done]), dst=g - 1)
        st = E.small(stop)
        dist.all_reduce(st, op=dist.ReduceOp.SUM)
        all_ops = self._gather_ops(ops)
        out = AlignResult(score, (bi, bj), tuple(int(v) for v in st.tolist()), all_ops)
        out.fill_ms, out.kernel_fill_ms, out.traceback_ms = res.fill_ms, res.kernel_fill_ms, tb_ms
        out.n_launches, out.rows_per_lane, out.tile_steps = res.n_launches, res.rows_per_lane, res.tile_steps
        out.cells_trace, out.cells_notrace = getattr(res, "cells_trace", 0), getattr(res, "cells_notrace", 0)
        out.border_wait_ms = getattr(res, "border_wait_ms", 0.0)
        out.extra = {"stripe": (c0, c0 + ng), "stripe_ops": len(ops), "owner": owner, "overlapped": bool(overlapped)}
        return out

    def _on_rank0(self, kind, x, y, mode, go, ge) -> AlignResult:
        """Inputs too small to stripe: rank 0 aligns, everybody learns the score."""
        E, dist = self.engine, self.dist
        ctx = E.stream_context() if hasattr(E, "stream_context") else contextlib.nullcontext()
        with ctx:
            return self._on_rank0_inner(kind, x, y, mode, go, ge)

    def _on_rank0_inner(self, kind, x, y, mode, go, ge) -> AlignResult:
        E, dist = self.engine, self.dist
        if self.rank == 0:
            r = E.whole(kind, x, y, mode, go, ge)
            vals = [r.score]
        else:
            r = AlignResult(0, (len(x), len(y)), (0, 0), np.zeros(0, np.uint8))
            vals = [0]
        if self.world > 1:
            t = E.small(vals)
            dist.broadcast(t, src=0)
            r.score = int(t.item())
        return r
