#!/usr/bin/env python3
"""The collective calls of the data-parallel path on the real backend ("nccl" = RCCL), as far as ONE GPU allows: a process
group of world size 1 launched the way the driver launches bench.py for N > 1, then the same calls, argument shapes and
dtypes the engine / bench use (all_gather_into_tensor of the packed embeddings, all_reduce(SUM) over a live-gradient-sized
fp32 range, all_reduce(MAX) of the fp64 timing scalar, barrier).

    python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29533 tools/rccl_selfcheck.py
"""
import os
import sys

import torch
import torch.distributed as dist

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main():
    rank, world, local = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"]), int(os.environ["LOCAL_RANK"])
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)
    dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)
    B, D = 64, 768
    reps = torch.randn(2 * B, D, device=dev)
    all_reps = torch.zeros(world, 2 * B, D, device=dev)
    dist.all_gather_into_tensor(all_reps.view(world * 2 * B, D), reps)
    assert torch.equal(all_reps[rank], reps)
    g = torch.ones(212_123_392, device=dev)                       # the MAE pass's live gradient range
    dist.all_reduce(g[38_400:])                                   # a slice of the flat buffer, like arena.live_slice
    assert float(g.sum()) == g.numel() * 1.0 * 1 or world > 1
    t = torch.tensor([1.25], device=dev, dtype=torch.float64)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    torch.cuda.synchronize()
    dist.barrier()
    torch.cuda.synchronize()
    print(f"rank {rank}/{world}: rccl collectives ok (backend {dist.get_backend()})", flush=True)
    native_comm(rank, world, dev)
    if "--engine" in sys.argv:
        engine_path(rank, world, dev)
    dist.destroy_process_group()


def native_comm(rank, world, dev):
    """The C ABI's own communicator (avs_comm_* / avs_allreduce / avs_allgather / avs_reducescatter): fp32 and bf16, ordering against
    the caller's stream in both directions."""
    from avsiam_amd import _lib
    from avsiam_amd.comm import RcclComm
    c = RcclComm(always=True)
    assert (c.rank, c.world) == (rank, world)
    assert _lib.load().avs_comm_rank(c._h) == rank and _lib.load().avs_comm_world(c._h) == world
    st = _lib.current_stream()
    for dt in (torch.float32, torch.bfloat16):
        x = torch.arange(1 << 20, device=dev, dtype=torch.float32).remainder(251).to(dt)
        y = x * 2                                               # queued on the caller's stream: the collective must see it
        h = c.all_reduce_async(y)
        h.wait()
        assert torch.equal(y, x * 2 * world)
        out = torch.zeros(world * x.numel(), device=dev, dtype=dt)
        c.all_gather(out, x)
        assert torch.equal(out.view(world, -1)[rank], x)
        rs = torch.zeros(x.numel() // world, device=dev, dtype=dt)
        _lib.call("avs_reducescatter", c._h, x, rs, rs.numel(), 0 if dt == torch.float32 else 1, st)
        _lib.call("avs_comm_wait", c._h, st)
        n = rs.numel()
        assert torch.equal(rs, x[rank * n:(rank + 1) * n] * world)
    torch.cuda.synchronize()
    c.close()
    print(f"rank {rank}/{world}: native communicator ok", flush=True)


def engine_path(rank, world, dev):
    """The data-parallel code path of the model itself on this process group: with comm.TorchDistComm(always=True) the embedding
    all-gather (engine.ContrastivePass.forward) and the chunked asynchronous gradient all-reduce (comm.GradReducer, issued from
    engine.Stack.backward) run on RCCL even at world size 1, where they must change nothing."""
    import random
    from avsiam_amd.comm import RcclComm, TorchDistComm
    from avsiam_amd.config import AVSiamConfig
    from avsiam_amd.maskplan import make_contrastive_plan, make_mae_plan
    from avsiam_amd.models import CAVMAE_BASE
    from avsiam_amd.param_spec import P1, P2
    from avsiam_amd.weights import synth_inputs
    cfg = AVSiamConfig(audio_tokens=128)
    B = 4
    a, v = synth_inputs(cfg, B, 87 + rank)
    a, v = a.to(dev), v.to(dev)
    gen = torch.Generator().manual_seed(1)
    pm, pc = make_mae_plan(cfg, B, gen), make_contrastive_plan(cfg, B, gen, random.Random(1))
    res = []
    # no collectives / torch.distributed / the C ABI's communicator / the latter with bf16 on the wire
    for comm, wire in ((None, "fp32"), (TorchDistComm(always=True), "fp32"), (RcclComm(always=True), "fp32"), (RcclComm(always=True), "bf16")):
        os.environ["AVSIAM_DP_WIRE"] = wire
        m = CAVMAE_BASE(cfg=cfg, init_seed=5, init_mode="random", verbose=False).to(dev)
        m.set_distributed(world, rank, comm)
        m.publish_grads = False
        outs = []
        for mae, plan, which in ((False, pc, P1), (True, pm, P2)):
            out = m(a, v, mae_loss_weight=1 if mae else 0, contrast_loss_weight=0 if mae else 1, mask_plan=plan)
            out[0].backward()
            lo, hi = m.arena.range[which]
            outs.append((out[0].item(), m.arena.g[lo:hi].double().norm().item(), m.last_reduce_messages))
        res.append(outs)
    torch.cuda.synchronize()
    os.environ.pop("AVSIAM_DP_WIRE")
    for k, other in enumerate(res[1:]):
        tol = 1e-5 if k < 2 else 2e-3                       # the bf16 wire rounds every gradient element once (2^-9 relative)
        for (l0, g0, _), (l1, g1, msgs) in zip(res[0], other):
            assert l0 == l1 and abs(g0 - g1) <= tol * g0, (k, l0, l1, g0, g1)
            assert world > 1 or msgs >= 2, msgs             # the chunked reducer really issued several all-reduces
    assert res[3][0][1] != res[0][0][1]                     # and it really went through the bf16 staging buffer
    print(f"rank {rank}/{world}: engine path on rccl ok (messages per pass: {[r[2] for r in res[1]]}; native communicator: {[r[2] for r in res[2]]})", flush=True)


if __name__ == "__main__":
    main()
