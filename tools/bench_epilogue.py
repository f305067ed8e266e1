#!/usr/bin/env python3
"""nt GEMM epilogue variants at the pass-1 row count, interleaved in one process (3 rounds; ignore the first)."""
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
    M = int(sys.argv[1]) if len(sys.argv) > 1 else 95630
    D, Hd = 768, 3072
    r = lambda n, dt=torch.bfloat16: (torch.randn(M, n, device=dev) * 0.5).to(dt)  # noqa: E731
    w = lambda n, k: (torch.randn(n, k, device=dev) * 0.03).bfloat16()  # noqa: E731
    xD, xH, oD, oH, oH2, fD, rD = r(D), r(Hd), r(D), r(Hd), r(Hd), r(D, torch.float32), r(D, torch.float32)
    Wp, W1, W2, W2t = w(D, D), w(Hd, D), w(D, Hd), w(Hd, D)
    bD, bH = torch.randn(D, device=dev), torch.randn(Hd, device=dev)
    v = {"proj bf16": lambda: ops.gemm_nt(xD, Wp, oD, M),
         "proj f32": lambda: ops.gemm_nt(xD, Wp, fD, M, bias=bD),
         "proj f32+res": lambda: ops.gemm_nt(xD, Wp, fD, M, bias=bD, res=rD),
         "fc2 bf16": lambda: ops.gemm_nt(xH, W2, oD, M),
         "fc2 f32+res": lambda: ops.gemm_nt(xH, W2, fD, M, bias=bD, res=rD),
         "fc1 plain": lambda: ops.gemm_nt(xD, W1, oH, M, bias=bH),
         "fc1 gelu": lambda: ops.gemm_nt(xD, W1, oH, M, bias=bH, out2=oH2, act=1),
         "fc2dg gelu'": lambda: ops.gemm_nt(xD, W2t, oH, M, aux=oH2, act=2)}
    from avsiam_amd import _lib
    lib = _lib.load()
    for rnd in range(3):
        for nt8 in (0, 1):
            lib.avs_gemm_set_nt8(nt8)
            print(("8-phase " if nt8 else "2-buffer") + "  " + "  ".join(f"{k}: {timeit(f) * 1e6:.1f}" for k, f in v.items()), flush=True)
    lib.avs_gemm_set_nt8(0)


if __name__ == "__main__":
    main()
