#!/usr/bin/env python3
"""Do two half-chip GEMMs running side by side (two streams, staggered epilogues) beat the same two GEMMs run one after the
other on the whole chip?   AVSIAM_NT_GRID=128 python tools/bench_stagger.py --streams 2   vs   python tools/bench_stagger.py"""
import argparse
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from avsiam_amd import ops  # noqa: E402

BF16, F32 = torch.bfloat16, torch.float32


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--streams", type=int, default=1)
    ap.add_argument("--iters", type=int, default=10)
    ap.add_argument("--tag", default="")
    args = ap.parse_args()
    dev = "cuda"
    M, D, Hd = 95630 // 2, 768, 3072
    Mp = ops.pad_rows(M, 256)
    sets = []
    for _ in range(2):
        rnd = lambda n, dt=BF16: (torch.randn(Mp, n, device=dev) * 0.5).to(dt)  # noqa: E731
        w = lambda n, k: (torch.randn(n, k, device=dev) * 0.03).to(BF16)  # noqa: E731
        sets.append(dict(xD=rnd(D), xH=rnd(Hd), o3=rnd(3 * D), oH=rnd(Hd), oH2=rnd(Hd), fD=rnd(D, F32), rD=rnd(D, F32), Wqkv=w(3 * D, D), Wproj=w(D, D),
                         Wfc1=w(Hd, D), Wfc2=w(D, Hd), b3=torch.randn(3 * D, device=dev), bD=torch.randn(D, device=dev), bH=torch.randn(Hd, device=dev)))

    def layer(s):
        ops.gemm_nt(s["xD"], s["Wqkv"], s["o3"], M, bias=s["b3"])
        ops.gemm_nt(s["xD"], s["Wproj"], s["fD"], M, bias=s["bD"], res=s["rD"])
        ops.gemm_nt(s["xD"], s["Wfc1"], s["oH"], M, bias=s["bH"], out2=s["oH2"], act=1)
        ops.gemm_nt(s["xH"], s["Wfc2"], s["fD"], M, bias=s["bD"], res=s["rD"])

    streams = [torch.cuda.Stream() for _ in range(2)]

    def run():
        if args.streams == 1:
            for s in sets:
                layer(s)
        else:
            cur = torch.cuda.current_stream()
            for st, s in zip(streams, sets):
                st.wait_stream(cur)
                with torch.cuda.stream(st):
                    layer(s)
            for st in streams:
                cur.wait_stream(st)

    run(); run(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(args.iters):
        run()
    e1.record(); torch.cuda.synchronize()
    t = e0.elapsed_time(e1) / args.iters
    fl = 2 * 2.0 * M * D * (3 * D + D + Hd + Hd)
    print(f"{args.tag:12s} streams={args.streams} grid={os.environ.get('AVSIAM_NT_GRID', 'all')}: {t * 1e3:8.1f} us for two half-batch layers  {fl / t / 1e9:7.1f} TF/s", flush=True)


if __name__ == "__main__":
    main()
