#!/usr/bin/env python3
"""Soak test of the 8-phase kernels: many repetitions, results compared BITWISE with the two-buffer forward/dgrad kernel
(same accumulation order) and within fp32 tolerance with a float64 reference for the wgrad (atomics reorder its sums)."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from avsiam_amd import _lib, ops  # noqa: E402


def main():
    reps = int(sys.argv[1]) if len(sys.argv) > 1 else 30
    dev = "cuda"
    lib = _lib.load()
    bad = 0
    torch.manual_seed(1)
    for M, N, K in [(95630, 768, 768), (95630, 2304, 768), (95630, 768, 3072), (158208, 512, 2048), (31360, 3072, 768), (39552, 768, 768)]:
        A = torch.randn(M, K, device=dev).bfloat16()
        W = (torch.randn(N, K, device=dev) * 0.05).bfloat16()
        bias = torch.randn(N, device=dev)
        res = torch.randn(M, N, device=dev)
        lib.avs_gemm_set_nt8(0)
        want_b = torch.zeros(M, N, device=dev, dtype=torch.bfloat16); ops.gemm_nt(A, W, want_b, M, bias=bias)
        want_f = torch.zeros(M, N, device=dev); ops.gemm_nt(A, W, want_f, M, bias=bias, res=res)
        lib.avs_gemm_set_nt8(1)
        nbad = 0
        for r in range(reps):
            out = torch.zeros_like(want_b); ops.gemm_nt(A, W, out, M, bias=bias)
            outf = torch.zeros_like(want_f); ops.gemm_nt(A, W, outf, M, bias=bias, res=res)
            nbad += (not torch.equal(out, want_b)) + (not torch.equal(outf, want_f))
        print(f"nt  M={M} N={N} K={K}: {reps} reps, {nbad} mismatching results", flush=True)
        bad += nbad
        del A, W, res, want_b, want_f
    for M, N1, N2 in [(95630, 2304, 768), (95630, 768, 3072), (158208, 2048, 512), (31360, 3072, 768)]:
        Mp = ops.pad_rows(M, 64)
        A = torch.zeros(Mp, N1, device=dev, dtype=torch.bfloat16); A[:M] = torch.randn(M, N1, device=dev).bfloat16()
        B = torch.zeros(Mp, N2, device=dev, dtype=torch.bfloat16); B[:M] = torch.randn(M, N2, device=dev).bfloat16()
        ref = (A.float().t() @ B.float()).double()
        scale = float(ref.abs().max())
        nbad = 0
        for r in range(reps):
            C = torch.zeros(N1, N2, device=dev); ops.gemm_tn(A, B, C, M)
            err = float((C.double() - ref).abs().max()) / scale
            nbad += err > 2e-5
        print(f"tn  M={M} N1={N1} N2={N2}: {reps} reps, {nbad} beyond tolerance (last rel err {err:.2e})", flush=True)
        bad += nbad
        del A, B, ref
    print("FAILED" if bad else "all ok")
    sys.exit(1 if bad else 0)


if __name__ == "__main__":
    main()
