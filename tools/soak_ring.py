#!/usr/bin/env python3
"""Soak / race screen of the round-5 LDS-DMA ring kernels: many repetitions over many shapes, results compared BITWISE with the kernels
they mirror - gemm_nt_ring_kernel<., 4 | 2> against gemm_nt_kernel<., 2, 4> (knob gemm_ring 2 vs 0), and the attention ring forward / dQ
against the register-staged kernels (knob attn_ring 1 vs 0) - with a second stream streaming 1 GiB back and forth meanwhile (uneven
load: the ring's counted waits must hold whatever the memory system is doing).   python tools/soak_ring.py [reps]"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from avsiam_amd import _lib, ops  # noqa: E402


def main():
    reps = int(sys.argv[1]) if len(sys.argv) > 1 else 40
    dev = "cuda"
    _lib.load()
    torch.manual_seed(3)
    noise_stream = torch.cuda.Stream()
    big = torch.zeros(256 << 20, device=dev)                 # 1 GiB
    bad = 0
    gen = torch.Generator().manual_seed(7)
    shapes = [(708, 768, 768), (1979, 768, 3072), (2832, 512, 2048), (130, 3072, 768), (64, 256, 256), (1000, 1536, 512), (3000, 768, 768),
              (257, 2304, 768), (4096, 768, 768), (5000, 512, 2048)]
    shapes += [(int(torch.randint(1, 6000, (1,), generator=gen)), 128 * int(torch.randint(1, 25, (1,), generator=gen)),
                64 * int(torch.randint(4, 49, (1,), generator=gen))) for _ in range(12)]
    for M, N, K in shapes:
        A = torch.randn(M, K, device=dev).bfloat16()
        W = (torch.randn(N, K, device=dev) * 0.05).bfloat16()
        bias = torch.randn(N, device=dev)
        res = torch.randn(M, N, device=dev)

        def run():
            o1 = torch.zeros(M, N, device=dev, dtype=torch.bfloat16); ops.gemm_nt(A, W, o1, M, bias=bias)
            o2 = torch.zeros(M, N, device=dev); ops.gemm_nt(A, W, o2, M, bias=bias, res=res)
            o3 = torch.zeros(M, N, device=dev, dtype=torch.bfloat16); o4 = torch.zeros_like(o3); ops.gemm_nt(A, W, o3, M, bias=bias, out2=o4, act=1)
            return o1, o2, o3, o4

        _lib.tuning_set("gemm_ring", 0)
        want = run()
        _lib.tuning_set("gemm_ring", 2)
        nbad = 0
        for r in range(reps):
            with torch.cuda.stream(noise_stream):
                big.add_(1.0)
            got = run()
            nbad += sum(not torch.equal(g, w) for g, w in zip(got, want))
        wgs = -(-M // 128) * (N // 128)
        print(f"gemm  M={M:5d} N={N:5d} K={K:5d} ({wgs:4d} workgroups of 128 x 128): {reps} reps, {nbad} mismatching results", flush=True)
        bad += nbad
    for H, hd in ((12, 64), (16, 32)):
        for tr in (128, 64):
            D = H * hd
            lens = [2472, 196, 49, 618, 65, 64, 1, 128, 129, 512, 63, 300, 1000]
            rows = sum(lens); rp = ops.pad_rows(rows)
            qkv = torch.zeros(rp, 3 * D, device=dev, dtype=torch.bfloat16)
            x = torch.randn(rows, 3 * D, device=dev); x[:, :D] *= ops.attn_q_scale(hd); qkv[:rows] = x.bfloat16()
            dout = torch.zeros(rp, D, device=dev, dtype=torch.bfloat16); dout[:rows] = torch.randn(rows, D, device=dev).bfloat16()
            tiles = ops.AttnTiles(lens, dev, tile_rows=tr)

            def arun():
                out = torch.zeros(rp, D, device=dev, dtype=torch.bfloat16); lse = torch.zeros(H, rp, device=dev)
                ops.attn_fwd(qkv, tiles, H, out, lse)
                dq = torch.zeros_like(qkv); delta = torch.zeros_like(lse)
                ops.attn_bwd(qkv, tiles, H, out, dout, lse, delta, dq)
                return out, lse, dq, delta

            _lib.tuning_set("attn_ring", 0)
            want = arun()
            _lib.tuning_set("attn_ring", 1)
            nbad = 0
            for r in range(reps):
                with torch.cuda.stream(noise_stream):
                    big.add_(1.0)
                got = arun()
                nbad += sum(not torch.equal(g, w) for g, w in zip(got, want))
            _lib.tuning_set("attn_ring", 0)
            print(f"attention  H={H} hd={hd} tile {tr}: {reps} reps, {nbad} mismatching results", flush=True)
            bad += nbad
    torch.cuda.synchronize()
    print("FAILED" if bad else "all ok")
    return 1 if bad else 0


if __name__ == "__main__":
    raise SystemExit(main())
