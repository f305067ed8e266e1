#!/usr/bin/env python3
"""The eight GEMMs of a block in fp8 mode 3 (forward on e4m3 operands, input gradients on e5m2 x e4m3, 8-bit-only outputs where the step has
them), one shape set per model: us and TFLOP/s per kind.   python tools/bench_fp8_kinds.py [--model vit_huge14|vit_base]"""
import argparse
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from avsiam_amd import ops  # noqa: E402

BF16, F32, U8 = torch.bfloat16, torch.float32, torch.uint8


def timeit(fn, iters=10):
    fn(); fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters * 1e-3


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--model", default="vit_huge14")
    ap.add_argument("--rows", type=int, default=0)
    args = ap.parse_args()
    dev = "cuda"
    R, D, H = (51456, 1280, 5120) if args.model == "vit_huge14" else (95630, 768, 3072)
    if args.rows:
        R = args.rows
    Rp = ops.pad_rows(R, 256)

    def q(rows, cols, e5m2=False):
        x = torch.randn(rows, cols, device=dev)
        return ops.quantize_fp8(x, 100.0 if not e5m2 else 1000.0, e5m2=e5m2)

    def w(n, k):
        return ops.quantize_fp8(torch.randn(n, k, device=dev) * 0.05, 1000.0)

    a_d, a_h, a_3d = q(Rp, D), q(Rp, H), q(Rp, 3 * D)
    g_d, g_h, g_3d = q(Rp, D, True), q(Rp, H, True), q(Rp, 3 * D, True)
    res = torch.randn(Rp, D, device=dev)
    gp = torch.randn(Rp, H, device=dev).to(BF16)
    alpha = 1.0 / (100.0 * 1000.0)
    total = 0.0
    kinds = []
    # forward
    o = torch.zeros(Rp, 3 * D, device=dev, dtype=BF16); W = w(3 * D, D); b = torch.randn(3 * D, device=dev)
    kinds.append(("qkv fwd   -> bf16", 2.0 * R * 3 * D * D, lambda o=o, W=W, b=b: ops.gemm_nt_fp8(a_d, W, o, R, alpha, bias=b, scale_cols=D, col_scale=0.125)))
    of = torch.zeros(Rp, D, device=dev); W2 = w(D, D); b2 = torch.randn(D, device=dev)
    kinds.append(("proj fwd  -> fp32 + res", 2.0 * R * D * D, lambda: ops.gemm_nt_fp8(a_d, W2, of, R, alpha, bias=b2, res=res)))
    og = torch.zeros(Rp, H, device=dev, dtype=BF16); o8 = torch.zeros(Rp, H, device=dev, dtype=U8); W3 = w(H, D); b3 = torch.randn(H, device=dev)
    kinds.append(("fc1 fwd   -> gelu' bf16 + e4m3(gelu)", 2.0 * R * H * D, lambda: ops.gemm_nt_fp8(a_d, W3, og, R, alpha, bias=b3, act=1, out8=o8, out8_scale=50.0)))
    og2 = torch.zeros(Rp, H, device=dev, dtype=BF16)
    kinds.append(("fc1 fwd   -> gelu' bf16 + gelu bf16 + e4m3", 2.0 * R * H * D, lambda: ops.gemm_nt_fp8(a_d, W3, og, R, alpha, bias=b3, act=1, out2=og2, out8=o8, out8_scale=50.0)))
    W4 = w(D, H)
    kinds.append(("fc2 fwd   -> fp32 + res", 2.0 * R * D * H, lambda: ops.gemm_nt_fp8(a_h, W4, of, R, alpha, bias=b2, res=res)))
    # input gradients (weights transposed)
    W4t = w(H, D); cs = torch.zeros(H, device=dev); d8 = torch.zeros(Rp, H, device=dev, dtype=U8)
    kinds.append(("fc2 dgrad -> e5m2 only (x gelu', colsum)", 2.0 * R * H * D, lambda: ops.gemm_nt_fp8(g_d, W4t, None, R, alpha, act=2, aux=gp, colsum=cs, out8=d8, out8_scale=1000.0, grad=True)))
    W3t = w(D, H); ob = torch.zeros(Rp, D, device=dev, dtype=BF16)
    kinds.append(("fc1 dgrad -> bf16", 2.0 * R * D * H, lambda: ops.gemm_nt_fp8(g_h, W3t, ob, R, alpha, grad=True)))
    W2t = w(D, D)
    kinds.append(("proj dgrad-> bf16", 2.0 * R * D * D, lambda: ops.gemm_nt_fp8(g_d, W2t, ob, R, alpha, grad=True)))
    Wt = w(D, 3 * D)
    kinds.append(("qkv dgrad -> bf16", 2.0 * R * D * 3 * D, lambda: ops.gemm_nt_fp8(g_3d, Wt, ob, R, alpha, grad=True)))
    for name, fl, fn in kinds:
        t = timeit(fn)
        if "+ gelu bf16" not in name:
            total += t
        print(f"{name:44s} {t * 1e6:8.1f} us {fl / t * 1e-12:7.0f} TF/s", flush=True)
    fl_all = 2.0 * R * D * (3 * D + D + 2 * H) * 2
    print(f"block (8 GEMMs, rows {R}, D {D}): {total * 1e6:.0f} us, {fl_all / total * 1e-12:.0f} TF/s", flush=True)


if __name__ == "__main__":
    main()
