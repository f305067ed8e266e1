#!/usr/bin/env python3
"""Micro-benchmark of the bf16 MFMA GEMMs at the hot-path shapes (interleaved variants in ONE process, random data).

    python tools/bench_gemm.py [--rows 95630] [--rounds 5]
"""
import argparse
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from avsiam_amd import _lib, ops  # noqa: E402


def timeit(fn, iters=10):
    fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters * 1e-3


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--rows", type=int, default=95630)
    ap.add_argument("--rounds", type=int, default=3)
    args = ap.parse_args()
    dev = "cuda"
    M = args.rows
    Mp = ops.pad_rows(M, 256)
    # (K, N, act): qkv, proj / dgrad-proj, fc1 (+GELU), fc2, fc2-dgrad (+GELU'), fc1-dgrad, qkv-dgrad, decoder fc1 / fc2
    shapes = [(768, 2304, 0), (768, 768, 0), (768, 3072, 1), (3072, 768, 0), (768, 3072, 2), (3072, 768, 0), (2304, 768, 0),
              (512, 2048, 1), (2048, 512, 0), (512, 2048, 2)]
    lib = _lib.load()
    for K, N, act in shapes:
        A = torch.zeros(Mp, K, device=dev, dtype=torch.bfloat16); A[:M] = torch.randn(M, K, device=dev).bfloat16()
        W = (torch.randn(N, K, device=dev) * 0.05).bfloat16()
        bias = torch.randn(N, device=dev)
        out = torch.zeros(Mp, N, device=dev, dtype=torch.bfloat16)
        out2 = torch.zeros(Mp, N, device=dev, dtype=torch.bfloat16) if act == 1 else None
        aux = torch.randn(Mp, N, device=dev).bfloat16() if act == 2 else None
        res = {}
        for _ in range(args.rounds):
            for tile, pp in ((128, 0), (256, 0), (256, 1)):
                lib.avs_gemm_set_tile(tile)
                lib.avs_gemm_set_persistent(pp)
                t = timeit(lambda: ops.gemm_nt(A, W, out, M, bias=bias, out2=out2, aux=aux, act=act))
                res.setdefault(f"{tile}{'p' if pp else ''}", []).append(2.0 * M * N * K / t / 1e12)
        lib.avs_gemm_set_tile(0)
        lib.avs_gemm_set_persistent(1)
        print(f"nt  M={M} K={K} N={N} act={act}: " + "  ".join(f"tile{t}: med {sorted(v)[len(v)//2]:.0f} max {max(v):.0f} TF/s" for t, v in res.items()), flush=True)
    for N1, N2 in [(2304, 768), (768, 768), (3072, 768), (768, 3072)]:
        A = torch.zeros(Mp, N1, device=dev, dtype=torch.bfloat16); A[:M] = torch.randn(M, N1, device=dev).bfloat16()
        B = torch.zeros(Mp, N2, device=dev, dtype=torch.bfloat16); B[:M] = torch.randn(M, N2, device=dev).bfloat16()
        C = torch.zeros(N1, N2, device=dev)
        res = {}
        for _ in range(args.rounds):
            for tile in (128, 256, 0):
                lib.avs_gemm_set_tile(tile)
                res.setdefault(tile or "auto", []).append(2.0 * M * N1 * N2 / timeit(lambda: ops.gemm_tn(A, B, C, M)) / 1e12)
        lib.avs_gemm_set_tile(0)
        print(f"tn  M={M} N1={N1} N2={N2}: " + "  ".join(f"tile{t}: med {sorted(v)[len(v)//2]:.0f} max {max(v):.0f} TF/s" for t, v in res.items()), flush=True)


if __name__ == "__main__":
    main()
