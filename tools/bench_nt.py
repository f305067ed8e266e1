#!/usr/bin/env python3
"""Forward/dgrad GEMM shapes of the B=64, T=10 step, one line per shape (us, TFLOP/s): for A/B runs of kernel builds on ONE box
    for v in old new old new; do AVSIAM_HIP_LIB=$PWD/avsiam_amd/csrc/ab_$v.so python tools/bench_nt.py --tag $v; done
"""
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
    ap.add_argument("--stacks", default="p1,dec")
    args = ap.parse_args()
    dev = "cuda"
    stacks = {"p1": (95630, 768, 3072), "tow": (39552, 768, 3072), "dec": (158208, 512, 2048)}
    tot = 0.0
    for sname in args.stacks.split(","):
        M, D, Hd = stacks[sname]
        Mp = ops.pad_rows(M, 256)
        rnd = lambda n, dt=BF16: (torch.randn(Mp, n, device=dev) * 0.5).to(dt)  # noqa: E731
        xD, xH, x3 = rnd(D), rnd(Hd), rnd(3 * D)
        oD, oH, o3, oH2 = rnd(D), rnd(Hd), rnd(3 * D), rnd(Hd)
        fD, rD = rnd(D, F32), rnd(D, F32)
        w = lambda n, k: (torch.randn(n, k, device=dev) * 0.03).to(BF16)  # noqa: E731
        Wqkv, Wproj, Wfc1, Wfc2 = w(3 * D, D), w(D, D), w(Hd, D), w(D, Hd)
        Wqkv_t, Wfc1_t, Wfc2_t = w(D, 3 * D), w(D, Hd), w(Hd, D)
        b3, bD, bH = torch.randn(3 * D, device=dev), torch.randn(D, device=dev), torch.randn(Hd, device=dev)
        cs = torch.zeros(Hd, device=dev)
        kinds = [
            ("qkv_fwd", lambda: ops.gemm_nt(xD, Wqkv, o3, M, bias=b3, scale_cols=D, col_scale=0.18), 3 * D, D),
            ("proj_fwd_res", lambda: ops.gemm_nt(xD, Wproj, fD, M, bias=bD, res=rD), D, D),
            ("fc1_fwd_gelu", lambda: ops.gemm_nt(xD, Wfc1, oH, M, bias=bH, out2=oH2, act=1), Hd, D),
            ("fc2_fwd_res", lambda: ops.gemm_nt(xH, Wfc2, fD, M, bias=bD, res=rD), D, Hd),
            ("fc2_dgrad_gelu'", lambda: ops.gemm_nt(xD, Wfc2_t, oH, M, aux=oH2, act=2, colsum=cs), Hd, D),
            ("fc1_dgrad", lambda: ops.gemm_nt(xH, Wfc1_t, oD, M), D, Hd),
            ("proj_dgrad", lambda: ops.gemm_nt(xD, Wproj, oD, M), D, D),
            ("qkv_dgrad", lambda: ops.gemm_nt(x3, Wqkv_t, oD, M), D, 3 * D),
        ]
        for kname, fn, N, K in kinds:
            t = timeit(fn, args.iters)
            tot += t
            print(f"{args.tag:10s} {sname:4s} {kname:16s} M={M} N={N:5d} K={K:5d} {t * 1e6:8.1f} us {2.0 * M * N * K / t / 1e12:7.1f} TF/s", flush=True)
        del xD, xH, x3, oD, oH, o3, oH2, fD, rD
    print(f"{args.tag:10s} total {tot * 1e6:.1f} us", flush=True)


if __name__ == "__main__":
    main()
