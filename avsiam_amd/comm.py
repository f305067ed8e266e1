"""Collectives of the data-parallel path, injected into the model.

The reference gets its collectives from two places: ``GatherLayer`` (/root/reference/src/models/gather_layer.py:21-37:
all_gather forward, all_reduce backward) and DistributedDataParallel's bucketed gradient all-reduce
(/root/reference/src/traintest_cavmae_base.py:58-59).  Here they are two calls on a small object:

    all_gather(out, inp)            out[W * n] <- every rank's inp[n]          (c2: ONE packed [2,B,D] message per rank)
    all_reduce_async(t) -> handle   SUM over ranks, in place; handle.wait() orders the caller's stream behind it   (c1)

``TorchDistComm`` is the product implementation: torch.distributed with backend "nccl", which on ROCm is RCCL over xGMI.
Tests inject their own implementation (gloo with host staging: two ranks cannot share one GPU under RCCL), so no test-only
branch lives in the product code.  ``GradReducer`` is the DDP-bucket equivalent on the flat gradient arena: contiguous
chunks are all-reduced as soon as the backward schedule declares them final, overlapping the rest of the backward.
"""
import os

import torch


class LocalComm:
    """World size 1: nothing to exchange."""
    world, rank, active = 1, 0, False

    def all_gather(self, out, inp):
        out.view(-1)[:inp.numel()].copy_(inp.reshape(-1))

    def all_reduce_async(self, t):
        return _Done()

    def all_reduce(self, t):
        pass


class _Done:
    def wait(self):
        pass


class TorchDistComm:
    """torch.distributed ("nccl" == RCCL on ROCm).  Asynchronous all-reduces run on the backend's own stream: the call
    orders them behind the work already queued on the current stream, handle.wait() orders the current stream behind them."""

    def __init__(self, group=None, always=False):
        """always=True: issue the collectives at world size 1 too (a one-rank RCCL group is the most a single-GPU box can
        run of this path; tests/test_boundary_gpu.py)."""
        import torch.distributed as dist
        self.dist, self.group = dist, group
        self.world, self.rank = dist.get_world_size(group), dist.get_rank(group)
        self.active = self.world > 1 or always

    def all_gather(self, out, inp):
        self.dist.all_gather_into_tensor(out, inp, group=self.group)

    def all_reduce_async(self, t):
        return self.dist.all_reduce(t, group=self.group, async_op=True)

    def all_reduce(self, t):
        self.dist.all_reduce(t, group=self.group)


class RcclComm:
    """The same collectives through the C ABI's own RCCL wrappers (avs_comm_* / avs_allreduce / avs_allgather of
    include/avsiam_hip.h): one communicator with a stream of its own and event hand-off - no torch.distributed call on the
    data path.  torch.distributed (any backend, or none at world size 1) is used once, to hand rank 0's 128-byte
    rendezvous id to the other ranks.  Select with ``set_distributed(..., comm=RcclComm(...))`` or AVSIAM_COMM=rccl."""

    def __init__(self, rank=None, world=None, group=None, always=False):
        import ctypes
        from . import _lib
        self._lib, self._ct = _lib, ctypes
        import torch.distributed as dist
        have = dist.is_available() and dist.is_initialized()
        self.rank = rank if rank is not None else (dist.get_rank(group) if have else 0)
        self.world = world if world is not None else (dist.get_world_size(group) if have else 1)
        self.active = self.world > 1 or always
        uid = ctypes.create_string_buffer(128)
        if self.rank == 0:
            _lib.call("avs_comm_unique_id", ctypes.cast(uid, ctypes.c_void_p))
        if self.world > 1:
            if not have:
                raise _lib.AvsiamHipError("RcclComm: world size > 1 needs an initialised torch.distributed group to exchange the rendezvous id")
            box = [bytes(uid.raw)]
            dist.broadcast_object_list(box, src=dist.get_global_rank(group, 0) if group is not None else 0, group=group)
            uid = ctypes.create_string_buffer(box[0], 128)
        handle = ctypes.c_void_p()
        _lib.call("avs_comm_init", ctypes.cast(uid, ctypes.c_void_p), self.rank, self.world, ctypes.byref(handle))
        self._h = handle
        # destroyed at interpreter exit while HIP and RCCL are still loaded (a __del__ during teardown may run after they are gone)
        import atexit
        import weakref
        ref = weakref.ref(self)
        atexit.register(lambda: ref() is not None and ref().close())

    def _dtype(self, t):
        if t.dtype == torch.float32:
            return 0
        if t.dtype == torch.bfloat16:
            return 1
        raise self._lib.AvsiamHipError(f"RcclComm: unsupported dtype {t.dtype}")

    def all_gather(self, out, inp):
        assert out.is_contiguous() and inp.is_contiguous() and out.numel() == self.world * inp.numel()
        st = self._lib.current_stream()
        self._lib.call("avs_allgather", self._h, inp, out, inp.numel(), self._dtype(inp), st)
        self._lib.call("avs_comm_wait", self._h, st)

    def all_reduce_async(self, t):
        assert t.is_contiguous()
        self._lib.call("avs_allreduce", self._h, t, t.numel(), self._dtype(t), self._lib.current_stream())
        return self

    def wait(self):
        self._lib.call("avs_comm_wait", self._h, self._lib.current_stream())

    def all_reduce(self, t):
        self.all_reduce_async(t).wait()

    def close(self):
        if self._h is not None:
            self._lib.call("avs_comm_destroy", self._h)
            self._h = None

    def __del__(self):
        import sys
        if sys is None or sys.is_finalizing():      # interpreter teardown: HIP / RCCL may already be unloaded - atexit has closed us
            return
        try:
            self.close()
        except Exception:
            pass


class HostStagedComm:
    """The same collectives on a gloo group, device tensors staged through the host.  NOT a production path (every message is a
    device -> host -> gloo -> host -> device round trip, synchronous): it exists to rehearse the multi-rank code path where RCCL
    cannot run - several ranks sharing one GPU (RCCL refuses two ranks on one device), i.e. the tests and `bench.py` under
    AVSIAM_BENCH_SHARE_GPU=1.  `messages` records the element count of every all-reduce."""

    def __init__(self, group=None):
        import torch.distributed as dist
        self.dist, self.group = dist, group
        self.world, self.rank = dist.get_world_size(group), dist.get_rank(group)
        self.active = self.world > 1
        self.messages = []

    def all_gather(self, out, inp):
        host = [torch.empty(inp.shape, dtype=inp.dtype) for _ in range(self.world)]
        self.dist.all_gather(host, inp.detach().cpu(), group=self.group)
        out.view(self.world, -1).copy_(torch.stack([h.reshape(-1) for h in host]))

    def all_reduce_async(self, t):
        self.all_reduce(t)
        return _Done()

    def all_reduce(self, t):
        self.messages.append(t.numel())
        h = t.detach().cpu()
        self.dist.all_reduce(h, group=self.group)
        t.copy_(h)


def default_comm(world):
    """AVSIAM_COMM=rccl: the C ABI's own RCCL communicator instead of torch.distributed's."""
    if world <= 1:
        return LocalComm()
    if os.environ.get("AVSIAM_COMM", "torch") == "rccl":
        return RcclComm()
    return TorchDistComm()


class GradReducer:
    """All-reduce(SUM) of one pass's live gradient range [lo, hi) of the flat arena, in chunks.

    The backward schedule calls ``ready(a, b)`` when the gradients in [a, b) are final (all kernels writing them are
    queued on the current stream, side streams joined); the chunk's all-reduce starts right away and overlaps the rest
    of the backward.  ``finish()`` reduces whatever part of [lo, hi) was never declared and waits for everything.
    Ranges may arrive in any order and may touch; they must not overlap.  Small ready-ranges are coalesced until
    ``min_elems`` are pending, so that xGMI sees few, large messages (ring collectives are per-link bound).

    mode (AVSIAM_DP_OVERLAP): "1" (default) chunks as described; "0" one blocking message in finish() - what round 1 did.
    wire (AVSIAM_DP_WIRE): "fp32" (default) reduces the arena in place; "bf16" halves the bytes on xGMI: each chunk is rounded
    to bf16 into a staging buffer, summed over the ranks in bf16, and written back over the fp32 gradients when its handle
    is waited for (the local gradients and everything downstream - Adam moments, master weights - stay fp32).
    """

    def __init__(self, comm, g, lo, hi, min_elems=16 << 20, overlap=None, wire=None, staging=None, boundary=None, staging_lo=None):
        """boundary: an offset inside (lo, hi) no message may straddle - finish(defer_from=boundary) can then leave the messages
        above it in flight (the MAE-only parameters, whose all-reduce need not end before the next contrastive pass).
        staging / staging_lo: the bf16 wire buffer and the arena offset its element 0 stands for.  A buffer shared between reducers
        (the model keeps one across steps) must be indexed by ABSOLUTE arena offset (staging_lo = 0, at least `hi` elements): messages
        one reducer leaves in flight (deferral) then never alias the range another reducer stages."""
        self.comm, self.g, self.lo, self.hi = comm, g, lo, hi
        self.boundary = boundary if boundary is not None and lo < boundary < hi else None
        self.deferred = []
        self.min_elems = min_elems
        if overlap is None:
            overlap = os.environ.get("AVSIAM_DP_OVERLAP", "1") != "0"
        self.active = getattr(comm, "active", comm.world > 1)
        self.overlap = overlap and self.active
        self.wire = wire or os.environ.get("AVSIAM_DP_WIRE", "fp32")
        assert self.wire in ("fp32", "bf16"), self.wire
        # staging: a bf16 buffer of at least hi - lo elements the caller keeps across steps (allocated here if not given)
        self.staging, self.staging_lo = staging, (lo if staging_lo is None else staging_lo)
        if self.wire == "bf16" and self.active and (staging is None or self.staging_lo > lo or self.staging_lo + staging.numel() < hi):
            self.staging, self.staging_lo = torch.empty(hi - lo, dtype=torch.bfloat16, device=g.device), lo
        self.sent = []            # [a, b) ranges already handed to the collective
        self.pending = []         # declared final, not yet sent
        self.handles = []
        self.messages = 0

    def _send(self, a, b):
        if self.boundary is not None and a < self.boundary < b:
            self._send(a, self.boundary)
            self._send(self.boundary, b)
            return
        if self.wire == "bf16":
            st = self.staging[a - self.staging_lo:b - self.staging_lo]
            st.copy_(self.g[a:b])                                    # round to nearest even, on the stream the gradients were written on
            self.handles.append((self.comm.all_reduce_async(st), a, b))
        else:
            self.handles.append((self.comm.all_reduce_async(self.g[a:b]), a, b))
        self.messages += 1

    def ready(self, a, b):
        if not self.overlap or b <= a:
            return
        a, b = max(a, self.lo), min(b, self.hi)
        if b <= a:
            return
        self.pending.append((a, b))
        if sum(y - x for x, y in self.pending) >= self.min_elems:
            self._flush()

    def _merge(self, ranges):
        out = []
        for a, b in sorted(ranges):
            if out and a <= out[-1][1]:
                assert a == out[-1][1], "GradReducer: overlapping ready ranges"
                out[-1] = (out[-1][0], b)
            else:
                out.append((a, b))
        return out

    def _flush(self):
        for a, b in self._merge(self.pending):
            self._send(a, b)
            self.sent.append((a, b))
        self.pending = []

    def _wait(self, handles):
        for h, a, b in handles:
            h.wait()
            if self.wire == "bf16":
                self.g[a:b].copy_(self.staging[a - self.staging_lo:b - self.staging_lo])

    def finish(self, defer_from=None):
        """Send what was never declared, then wait.  defer_from (= the reducer's boundary): messages at or above it stay in flight;
        wait_deferred() orders the caller's stream behind them later."""
        if not self.active:
            return
        assert defer_from is None or defer_from == self.boundary, "defer_from must be the boundary the messages were split at"
        self._flush()
        cur = self.lo
        for a, b in self._merge(self.sent) + [(self.hi, self.hi)]:
            if a > cur:
                self._send(cur, a)
            cur = max(cur, b)
        now = [x for x in self.handles if defer_from is None or x[1] < defer_from]
        self.deferred = [x for x in self.handles if not (defer_from is None or x[1] < defer_from)]
        self._wait(now)
        self.handles, self.sent = [], []

    def wait_deferred(self):
        self._wait(self.deferred)
        self.deferred = []
