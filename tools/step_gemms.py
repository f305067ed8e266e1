#!/usr/bin/env python3
"""Where the forward/dgrad GEMM time of one training step goes: every (stack, GEMM kind) of the B=64, T=10 step timed
in isolation (random data, one process), next to its MFMA floor (2.5 PFLOP/s) and its HBM floor (algorithmic bytes at
8 TB/s).

    python tools/step_gemms.py [--iters 10]
"""
import argparse
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from avsiam_amd import ops  # noqa: E402

BF16, F32 = torch.bfloat16, torch.float32


def timeit(fn, iters):
    fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters * 1e-3


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--iters", type=int, default=10)
    args = ap.parse_args()
    dev = "cuda"
    # (name, rows, D, hidden, layers)
    stacks = [("P1 shared ViT", 95630, 768, 3072, 12), ("P2 audio tower", 8192, 768, 3072, 11), ("P2 video tower", 31360, 768, 3072, 11),
              ("P2 joint", 39552, 768, 3072, 1), ("P2 decoder", 158208, 512, 2048, 8)]
    grand = 0.0
    grand_w = 0.0
    for name, M, D, Hd, layers in stacks:
        Mp = ops.pad_rows(M, 256)
        rnd = lambda n, dt=BF16: (torch.randn(Mp, n, device=dev) * 0.5).to(dt)  # noqa: E731
        xD, xH, x3 = rnd(D), rnd(Hd), rnd(3 * D)
        oD, oH, o3, oH2 = rnd(D), rnd(Hd), rnd(3 * D), rnd(Hd)
        fD, rD = rnd(D, F32), rnd(D, F32)
        w = lambda n, k: (torch.randn(n, k, device=dev) * 0.03).to(BF16)  # noqa: E731
        Wqkv, Wproj, Wfc1, Wfc2 = w(3 * D, D), w(D, D), w(Hd, D), w(D, Hd)
        Wqkv_t, Wfc1_t, Wfc2_t = w(D, 3 * D), w(D, Hd), w(Hd, D)
        b3, bD, bH = torch.randn(3 * D, device=dev), torch.randn(D, device=dev), torch.randn(Hd, device=dev)
        # kind -> (callable, N, K, algorithmic bytes per row)
        kinds = [
            ("qkv fwd", lambda: ops.gemm_nt(xD, Wqkv, o3, M, bias=b3, scale_cols=D, col_scale=0.18), 3 * D, D, 2 * D + 6 * D),
            ("proj fwd (+res f32)", lambda: ops.gemm_nt(xD, Wproj, fD, M, bias=bD, res=rD), D, D, 2 * D + 8 * D),
            ("fc1 fwd (gelu, 2 out)", lambda: ops.gemm_nt(xD, Wfc1, oH, M, bias=bH, out2=oH2, act=1), Hd, D, 2 * D + 4 * Hd),
            ("fc2 fwd (+res f32)", lambda: ops.gemm_nt(xH, Wfc2, fD, M, bias=bD, res=rD), D, Hd, 2 * Hd + 8 * D),
            ("fc2 dgrad (gelu')", lambda: ops.gemm_nt(xD, Wfc2_t, oH, M, aux=oH2, act=2), Hd, D, 2 * D + 4 * Hd),
            ("fc1 dgrad", lambda: ops.gemm_nt(xH, Wfc1_t, oD, M), D, Hd, 2 * Hd + 2 * D),
            ("proj dgrad", lambda: ops.gemm_nt(xD, Wproj, oD, M), D, D, 4 * D),
            ("qkv dgrad", lambda: ops.gemm_nt(x3, Wqkv_t, oD, M), D, 3 * D, 6 * D + 2 * D),
        ]
        tot = 0.0
        print(f"== {name}: rows {M}, D {D}, x{layers} layers", flush=True)
        for kname, fn, N, K, bpr in kinds:
            t = timeit(fn, args.iters)
            fl = 2.0 * M * N * K
            t_mfma, t_hbm = fl / 2.5e15, M * bpr / 8e12
            tot += t * layers
            print(f"   {kname:24s} {t * 1e6:8.1f} us  {fl / t / 1e12:7.1f} TF/s   floors: mfma {t_mfma * 1e6:7.1f} us  hbm {t_hbm * 1e6:7.1f} us"
                  f"   x{layers} = {t * layers * 1e3:6.2f} ms", flush=True)
        print(f"   stack total {tot * 1e3:.2f} ms/step", flush=True)
        grand += tot
        # weight gradients: C[N1,N2] += dY[M,N1]^T . X[M,N2]
        totw = 0.0
        for kname, dy, x, N1, N2 in [("qkv wgrad", x3, xD, 3 * D, D), ("proj wgrad", xD, oD, D, D), ("fc1 wgrad", xH, xD, Hd, D), ("fc2 wgrad", xD, xH, D, Hd)]:
            C = torch.zeros(N1, N2, device=dev)
            dy[M:].zero_(); x[M:].zero_()
            t = timeit(lambda: ops.gemm_tn(dy, x, C, M), args.iters)
            fl = 2.0 * M * N1 * N2
            totw += t * layers
            print(f"   {kname:24s} {t * 1e6:8.1f} us  {fl / t / 1e12:7.1f} TF/s   floors: mfma {fl / 2.5e15 * 1e6:7.1f} us  hbm {M * 2 * (N1 + N2) / 8e12 * 1e6:7.1f} us"
                  f"   x{layers} = {t * layers * 1e3:6.2f} ms", flush=True)
        print(f"   stack wgrad total {totw * 1e3:.2f} ms/step", flush=True)
        grand_w += totw
        del xD, xH, x3, oD, oH, o3, oH2, fD, rD
    print(f"all forward+dgrad GEMMs: {grand * 1e3:.2f} ms/step; all wgrad GEMMs: {grand_w * 1e3:.2f} ms/step")


if __name__ == "__main__":
    main()
