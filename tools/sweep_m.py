#!/usr/bin/env python3
"""nt GEMM rate vs row count (qkv-forward shape): separates tile-quantisation / cache-footprint / sustained-clock effects."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from avsiam_amd import ops  # noqa: E402


def timeit(fn, iters):
    fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters * 1e-3


def main():
    dev = "cuda"
    K, N = 768, 2304
    W = (torch.randn(N, K, device=dev) * 0.03).bfloat16()
    for M in (256 * 28, 256 * 57, 256 * 114, 256 * 171, 256 * 228, 256 * 285, 256 * 370, 95630, 256 * 512, 256 * 626):
        A = (torch.randn(M, K, device=dev) * 0.5).bfloat16()
        out = torch.zeros(M, N, device=dev, dtype=torch.bfloat16)
        msg = f"M={M:7d} ({M * 9 / 256 / 256:6.2f} rounds)"
        for iters in (3, 10, 40):
            t = timeit(lambda: ops.gemm_nt(A, W, out, M), iters)
            msg += f"  x{iters}: {t * 1e6:7.1f} us {2.0 * M * N * K / t / 1e12:6.0f} TF/s"
        print(msg, flush=True)


if __name__ == "__main__":
    main()
