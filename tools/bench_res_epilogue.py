#!/usr/bin/env python3
"""Forward GEMMs that add into the fp32 residual stream (proj, fc2: fp32 output + fp32 residual read) against the same products with a bf16 output:
what the residual epilogue costs per shape, bf16 and fp8 operands.   python tools/bench_res_epilogue.py"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from avsiam_amd import ops  # noqa: E402

BF16 = torch.bfloat16


def timeit(fn, iters=20):
    fn(); fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters * 1e-3


def main():
    dev = "cuda"
    for name, M, N, K in (("ViT-B proj P1", 95630, 768, 768), ("ViT-B fc2 P1", 95630, 768, 3072), ("ViT-B proj towers", 39552, 768, 768), ("ViT-B fc2 towers", 39552, 768, 3072),
                          ("decoder proj", 158208, 512, 512), ("decoder fc2", 158208, 512, 2048), ("T=1 proj", 11328, 768, 768), ("T=1 fc2", 11328, 768, 3072),
                          ("ViT-H proj", 51456, 1280, 1280), ("ViT-H fc2", 51456, 1280, 5120)):
        Mp = ops.pad_rows(M, 256)
        A = (torch.randn(Mp, K, device=dev) * 0.5).to(BF16)
        W = (torch.randn(N, K, device=dev) * 0.03).to(BF16)
        b = torch.randn(N, device=dev)
        res = torch.randn(Mp, N, device=dev)
        of = torch.zeros(Mp, N, device=dev)
        ob = torch.zeros(Mp, N, device=dev, dtype=BF16)
        t_res = timeit(lambda: ops.gemm_nt(A, W, of, M, bias=b, res=res))
        t_f32 = timeit(lambda: ops.gemm_nt(A, W, of, M, bias=b))
        t_b16 = timeit(lambda: ops.gemm_nt(A, W, ob, M, bias=b))
        A8, W8 = ops.quantize_fp8(A, 100.0), ops.quantize_fp8(W, 1000.0)
        t8_res = timeit(lambda: ops.gemm_nt_fp8(A8, W8, of, M, 1e-5, bias=b, res=res))
        t8_b16 = timeit(lambda: ops.gemm_nt_fp8(A8, W8, ob, M, 1e-5, bias=b))
        fl = 2.0 * M * N * K
        hbm = (M * K * 2 + M * N * 8) / 5.5e12
        print(f"{name:18s} M={M:6d} N={N:4d} K={K:4d}: bf16 operands: fp32+res {t_res * 1e6:7.1f} us ({fl / t_res * 1e-12:5.0f} TF/s)  fp32 {t_f32 * 1e6:7.1f}  bf16 out {t_b16 * 1e6:7.1f}"
              f"   fp8 operands: fp32+res {t8_res * 1e6:7.1f}  bf16 out {t8_b16 * 1e6:7.1f}   (bytes of the fp32+res form at 5.5 TB/s: {hbm * 1e6:5.1f} us)", flush=True)


if __name__ == "__main__":
    main()
