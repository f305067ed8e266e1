#!/usr/bin/env python3
"""The collective calls of the data-parallel path on the real backend ("nccl" = RCCL), as far as ONE GPU allows: a process
group of world size 1 launched the way the driver launches bench.py for N > 1, then the same calls, argument shapes and
dtypes the engine / bench use (all_gather_into_tensor of the packed embeddings, all_reduce(SUM) over a live-gradient-sized
fp32 range, all_reduce(MAX) of the fp64 timing scalar, barrier).

    python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29533 tools/rccl_selfcheck.py
"""
import os

import torch
import torch.distributed as dist


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
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
