#!/usr/bin/env python3
"""fp8 (e4m3) vs bf16 forward GEMM on the step's shapes and on ViT-H's: us, TFLOP/s (operands quantised beforehand).
    python tools/bench_fp8.py"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from avsiam_amd import ops  # noqa: E402


def timeit(fn, iters=10):
    fn(); fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters * 1e-3


def main():
    dev = "cuda"
    for name, M, N, K, f32 in (("qkv ViT-B", 95630, 2304, 768, False), ("fc1-like ViT-B", 95630, 3072, 768, False), ("fc2 ViT-B", 95630, 768, 3072, True),
                               ("qkv ViT-H", 47815, 3840, 1280, False), ("fc1-like ViT-H", 47815, 5120, 1280, False), ("fc2 ViT-H", 47815, 1280, 5120, True)):
        Mp = ops.pad_rows(M, 256)
        A = torch.zeros(Mp, K, device=dev, dtype=torch.bfloat16); A[:M] = torch.randn(M, K, device=dev).bfloat16()
        W = (torch.randn(N, K, device=dev) * 0.05).bfloat16()
        b = torch.randn(N, device=dev)
        res = torch.randn(Mp, N, device=dev) if f32 else None
        out = torch.zeros(Mp, N, device=dev, dtype=torch.float32 if f32 else torch.bfloat16)
        sa, sw = ops.FP8_MAX / ops.absmax(A), ops.FP8_MAX / ops.absmax(W)
        A8, W8 = ops.quantize_fp8(A, sa), ops.quantize_fp8(W, sw)
        tq = timeit(lambda: ops.quantize_fp8(A, sa, out=A8))
        t16 = timeit(lambda: ops.gemm_nt(A, W, out, M, bias=b, res=res))
        t8 = timeit(lambda: ops.gemm_nt_fp8(A8, W8, out, M, 1.0 / (sa * sw), bias=b, res=res))
        fl = 2.0 * M * N * K
        print(f"{name:16s} M={M} N={N} K={K}: bf16 {t16 * 1e6:7.1f} us {fl / t16 / 1e12:6.0f} TF/s   fp8 {t8 * 1e6:7.1f} us {fl / t8 / 1e12:6.0f} TF/s   "
              f"({t16 / t8:.2f}x; quantising the activations: {tq * 1e6:.1f} us)", flush=True)


if __name__ == "__main__":
    main()
