#!/usr/bin/env python3
"""Forward GEMMs between the small-problem kernels and the persistent 256^2 kernel (the reference's one-frame shapes: 3 136 ... 45 312 rows):
rate by tile choice.   python tools/bench_nt_midsize.py"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from avsiam_amd import _lib, ops  # noqa: E402


def timeit(fn, iters=20):
    fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters * 1e-3


def main():
    dev = "cuda"
    shapes = []
    for M in (2832, 3136, 8192, 11328, 16384, 22656, 45312):
        D, H = (512, 2048) if M == 45312 else (768, 3072)
        shapes += [(M, D, 3 * D, 0), (M, D, D, 0), (M, D, H, 1), (M, H, D, 0)]
    for M, K, N, act in shapes:
        A = (torch.randn(M, K, device=dev) * 0.5).bfloat16()
        W = (torch.randn(N, K, device=dev) * 0.03).bfloat16()
        bias = torch.zeros(N, device=dev)
        out = torch.zeros(M, N, device=dev, dtype=torch.bfloat16)
        out2 = torch.zeros(M, N, device=dev, dtype=torch.bfloat16) if act else None
        msg = f"M={M:6d} K={K:4d} N={N:4d} act {act} ({-(-M // 256) * (N // 256):4d} tiles of 256^2):"
        for big_min in (1 << 20, 224, 0, 1):      # 128 x 128 kernels | rounds 1 - 4 | the default (half the CUs) | the persistent 256^2 kernel whatever the tile count
            _lib.tuning_set("nt_big_min", big_min)
            t = timeit(lambda: ops.gemm_nt(A, W, out, M, bias=bias, out2=out2, act=act))
            msg += f"  big_min {big_min:7d}: {t * 1e6:7.1f} us {2.0 * M * N * K / t * 1e-12:6.0f} TF/s"
        _lib.tuning_set("nt_big_min", 0)
        print(msg, flush=True)


if __name__ == "__main__":
    main()
