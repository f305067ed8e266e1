#!/usr/bin/env python3
"""Where the hand-written nt GEMM stands against the vendor library at the step's plain shapes (no epilogue on either side:
out = A @ W^T in bf16, fp32 accumulate).  Diagnostic only - the product path never calls the library.

    python tools/bench_vs_blaslt.py
"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from avsiam_amd import ops  # noqa: E402


def timeit(fn, iters=10):
    fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters * 1e-3


def main():
    dev = "cuda"
    # rows of the batch-64 step: decoder 158208, contrastive stack 95630, MAE towers + joint 39552
    cases = [(158208, 512, 1536), (158208, 512, 512), (158208, 512, 2048), (158208, 2048, 512), (158208, 1536, 512),
             (95630, 768, 2304), (95630, 768, 768), (95630, 768, 3072), (95630, 3072, 768), (95630, 2304, 768),
             (39552, 768, 2304), (39552, 768, 3072), (39552, 3072, 768)]
    for M, K, N in cases:
        Mp = ops.pad_rows(M, 256)
        A = torch.zeros(Mp, K, device=dev, dtype=torch.bfloat16); A[:M] = torch.randn(M, K, device=dev).bfloat16()
        W = (torch.randn(N, K, device=dev) * 0.05).bfloat16()
        out = torch.zeros(Mp, N, device=dev, dtype=torch.bfloat16)
        ref = torch.empty(Mp, N, device=dev, dtype=torch.bfloat16)
        Wt = W.t()
        r = {"ours": [], "lib": []}
        for _ in range(3):
            r["ours"].append(timeit(lambda: ops.gemm_nt(A, W, out, M)))
            r["lib"].append(timeit(lambda: torch.matmul(A, Wt, out=ref)))
        fl = 2.0 * M * N * K
        fl_lib = 2.0 * Mp * N * K
        o, l = sorted(r["ours"])[1], sorted(r["lib"])[1]
        err = float((out[:M].float() - ref[:M].float()).abs().max())
        print(f"M={M} K={K} N={N}: ours {o*1e6:7.1f} us {fl/o/1e12:6.0f} TF/s | library {l*1e6:7.1f} us {fl_lib/l/1e12:6.0f} TF/s | ratio {l/o:.2f} | max diff {err:.3f}", flush=True)


if __name__ == "__main__":
    main()
