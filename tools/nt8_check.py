#!/usr/bin/env python3
"""Correctness screen of the 8-phase nt GEMM against torch (several shapes, repeated: races show up as rare wrong tiles)."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from avsiam_amd import _lib, ops  # noqa: E402


def main():
    dev = "cuda"
    lib = _lib.load()
    lib.avs_gemm_set_nt8(1)
    torch.manual_seed(0)
    reps = int(sys.argv[1]) if len(sys.argv) > 1 else 5
    bad = 0
    for M, N, K in [(131072, 256, 128), (30000, 512, 128), (30000, 512, 256), (25700, 768, 768), (65536, 256, 128), (95630, 768, 3072), (60000, 2304, 768), (70001, 512, 2048), (57345, 256, 64 * 5)]:
        A = torch.randn(M, K, device=dev).bfloat16()
        W = (torch.randn(N, K, device=dev) * 0.05).bfloat16()
        bias = torch.randn(N, device=dev)
        ref = (A.float() @ W.float().t() + bias)
        res = torch.randn(M, N, device=dev)
        for r in range(reps):
            out = torch.zeros(M, N, device=dev, dtype=torch.bfloat16)
            ops.gemm_nt(A, W, out, M, bias=bias)
            e0 = (out.float() - ref).abs().max().item()
            outf = torch.zeros(M, N, device=dev)
            ops.gemm_nt(A, W, outf, M, bias=bias, res=res)
            e1 = (outf - ref - res).abs().max().item()
            pre = torch.zeros(M, N, device=dev, dtype=torch.bfloat16); act = torch.zeros_like(pre)
            ops.gemm_nt(A, W, pre, M, bias=bias, out2=act, act=1)
            x = ref.double().requires_grad_(True)                      # act 1 stores gelu'(x) in `out` (since round 3) and gelu(x) in out2
            g = torch.nn.functional.gelu(x)
            g.sum().backward()
            e2 = max((pre.double() - x.grad).abs().max().item(), ((act.double() - g.detach()).abs() / (1 + g.detach().abs())).max().item() * 4)
            dp = torch.zeros_like(pre)
            ops.gemm_nt(A, W, dp, M, aux=pre, act=2)
            ok = e0 < 0.13 and e1 < 2e-2 and e2 < 0.05 and bool(torch.isfinite(dp.float()).all())
            bad += not ok
            print(f"M={M} N={N} K={K} rep {r}: bf16 {e0:.4f} f32+res {e1:.5f} gelu pair {e2:.4f} {'ok' if ok else 'BAD'}", flush=True)
    print("FAILED" if bad else "all ok")
    sys.exit(1 if bad else 0)


if __name__ == "__main__":
    main()
