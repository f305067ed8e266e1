#!/usr/bin/env python3
"""Weight-gradient GEMM: time against the number of splits of the token dimension (every workgroup adds one 256 x 256 fp32 tile with atomics,
so a launch moves workgroups x 256 KiB of atomic traffic whatever its shape).   python tools/bench_tn_splits.py [M]"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from avsiam_amd import ops  # noqa: E402


def timeit(fn, iters=20):
    fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters * 1e3


def main():
    dev = "cuda"
    for M in ([int(sys.argv[1])] if len(sys.argv) > 1 else [2832, 11328, 27419, 95630]):
        Mp = ops.pad_rows(M, 64)
        for N1, N2 in ((3072, 768), (2304, 768), (768, 768)):
            A = torch.zeros(Mp, N1, device=dev, dtype=torch.bfloat16); A[:M] = (torch.randn(M, N1, device=dev) * 0.5).bfloat16()
            B = torch.zeros(Mp, N2, device=dev, dtype=torch.bfloat16); B[:M] = (torch.randn(M, N2, device=dev) * 0.5).bfloat16()
            C = torch.zeros(N1, N2, device=dev)
            tiles = (N1 // 256) * (N2 // 256)
            msg = f"M={M:6d} {N1}x{N2} ({tiles} tiles, default splits {max(1, 256 // tiles)}):"
            for s in sorted({1, 2, 3, 4, 5, 7, 9, 14, 28, max(1, 256 // tiles)}):
                if s * tiles > 256 or s > (Mp // 64):
                    continue
                t = timeit(lambda: ops.gemm_tn(A, B, C, M, s))
                msg += f"  s={s}: {t:6.1f} us"
            print(msg, flush=True)


if __name__ == "__main__":
    main()
