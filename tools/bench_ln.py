#!/usr/bin/env python3
"""LayerNorm forward / backward at the step's shapes: us, algorithmic GB/s (ops.layernorm_* byte counts)."""
import argparse
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from avsiam_amd import ops  # noqa: E402

BF16, F32 = torch.bfloat16, torch.float32


def timeit(fn, iters):
    fn(); fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters * 1e-3


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--iters", type=int, default=20)
    ap.add_argument("--tag", default="")
    ap.add_argument("--stream", choices=("fp32", "bf16"), default="bf16", help="residual-gradient stream of the backward (engine.GRAD_STREAM)")
    ap.add_argument("--shapes", default="", help="name:rows:D:calls,... instead of the ViT-B step (ViT-H/14: p1:51456:1280:64,tow:51456:1280:62,dec:205888:512:16)")
    ap.add_argument("--dma", type=int, default=-1, help="knob ln_dma for this run: 0 register kernel | 1 LDS-DMA, one row ahead | 2 two rows ahead")
    args = ap.parse_args()
    if args.dma >= 0:
        from avsiam_amd import _lib
        _lib.tuning_set("ln_dma", args.dma)
    dev = "cuda"
    tot = 0.0
    shapes = (("p1", 95630, 768, 24), ("video", 31360, 768, 22), ("audio", 8192, 768, 22), ("mm", 39552, 768, 4), ("dec", 158208, 512, 16))
    if args.shapes:
        shapes = tuple((n, int(r), int(d), int(c)) for n, r, d, c in (t.split(":") for t in args.shapes.split(",")))
    for name, rows, D, calls in shapes:
        rp = ops.pad_rows(rows, 128)
        x, dres, dx = (torch.randn(rp, D, device=dev) for _ in range(3))
        dy = torch.randn(rp, D, device=dev).to(BF16)
        y = torch.zeros(rp, D, device=dev, dtype=BF16)
        dxb = torch.zeros(rp, D, device=dev, dtype=BF16)
        g0, b0, g1, b1 = (torch.randn(D, device=dev) for _ in range(4))
        dg0, db0, dg1, db1, dcol = (torch.zeros(D, device=dev) for _ in range(5))
        mean, rstd = torch.zeros(rp, device=dev), torch.zeros(rp, device=dev)
        mod = (torch.arange(rows, device=dev) >= rows // 4).to(torch.uint8)
        ws = torch.zeros(ops.layernorm_ws(rows, D), device=dev)
        tf = timeit(lambda: ops.layernorm_fwd(x, g0, b0, y, mean, rstd, rows, 1e-5, g1, b1, mod), args.iters)
        if args.stream == "bf16":           # the step's form: bf16 residual-gradient stream in, bf16 dx out only (10 B per element)
            dres_b, dxb2 = dres.to(BF16), torch.zeros(rp, D, device=dev, dtype=BF16)
            tb = timeit(lambda: ops.layernorm_bwd(dy, x, mean, rstd, g0, None, dg0, db0, ws, rows, g1, dg1, db1, mod, None, dres_b, dxb2, dcol), args.iters)
            bb = rows * D * 10.0
        else:
            tb = timeit(lambda: ops.layernorm_bwd(dy, x, mean, rstd, g0, dx, dg0, db0, ws, rows, g1, dg1, db1, mod, None, dres, dxb, dcol), args.iters)
            bb = rows * D * 16.0
        bf = rows * D * 6.0
        tot += calls * (tf + tb)
        print(f"{args.tag:8s} {name:6s} rows={rows:6d} D={D}: fwd {tf * 1e6:7.1f} us {bf / tf / 1e9:7.0f} GB/s   bwd {tb * 1e6:7.1f} us {bb / tb / 1e9:7.0f} GB/s", flush=True)
    print(f"{args.tag:8s} per step (fwd+bwd, block LayerNorms): {tot * 1e3:.2f} ms", flush=True)


if __name__ == "__main__":
    main()
