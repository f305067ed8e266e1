#!/usr/bin/env python3
"""Attention throughput vs sequence length (random data, one process).  usage: bench_attn.py"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from avsiam_amd import ops  # noqa: E402


def timeit(fn, iters=5):
    fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters * 1e-3


def main():
    dev = "cuda"
    for H, hd, L, rows_target in [(12, 64, 39, 40000), (12, 64, 49, 40000), (12, 64, 78, 40000), (12, 64, 117, 40000), (12, 64, 128, 40000),
                                  (12, 64, 156, 40000), (12, 64, 196, 40000), (12, 64, 307, 40000), (12, 64, 512, 40000), (12, 64, 618, 40000),
                                  (16, 32, 708, 80000), (16, 32, 2472, 160000)]:
        D = H * hd
        nseq = max(1, rows_target // L)
        lens = [L] * nseq
        rows = nseq * L
        rp = ops.pad_rows(rows)
        qkv = torch.zeros(rp, 3 * D, device=dev, dtype=torch.bfloat16)
        qkv[:rows] = torch.randn(rows, 3 * D, device=dev).bfloat16()
        out = torch.zeros(rp, D, device=dev, dtype=torch.bfloat16)
        lse = torch.zeros(H, rp, device=dev)
        dout = torch.zeros(rp, D, device=dev, dtype=torch.bfloat16)
        dout[:rows] = torch.randn(rows, D, device=dev).bfloat16()
        dqkv = torch.zeros_like(qkv)
        delta = torch.zeros_like(lse)
        fl = 4.0 * nseq * L * L * D
        msg = f"H={H} hd={hd} L={L:5d} nseq={nseq:5d}:"
        for tr in (128, 64):
            tiles = ops.AttnTiles(lens, dev, tile_rows=tr)
            tf = timeit(lambda: ops.attn_fwd(qkv, tiles, H, out, lse))
            tb = timeit(lambda: ops.attn_bwd(qkv, tiles, H, out, dout, lse, delta, dqkv))
            msg += f"  [tile {tr}] fwd {tf*1e6:7.1f} us {fl/tf/1e12:6.1f} TF/s bwd {tb*1e6:7.1f} us {2.5*fl/tb/1e12:6.1f} TF/s"
        print(msg, flush=True)


if __name__ == "__main__":
    main()
