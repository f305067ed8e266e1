#!/usr/bin/env python3
"""Forward / input-gradient GEMMs of the reference's batch-4 step (708 - 2832 rows): two-buffer kernel vs the 4-slot LDS-DMA ring kernel,
interleaved rounds in one process (medians).   python tools/bench_small_gemm.py"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from avsiam_amd import _lib, ops  # noqa: E402


def timeit(fn, iters=50):
    fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters * 1e3


def main():
    dev = "cuda"
    for M in (708, 1979, 2832, 5664, 11328):
        for N, K in ((768, 768), (2304, 768), (3072, 768), (768, 3072), (512, 2048), (2048, 512)):
            A = torch.randn(M, K, device=dev).bfloat16()
            W = (torch.randn(N, K, device=dev) * 0.05).bfloat16()
            out = torch.zeros(M, N, device=dev, dtype=torch.bfloat16)
            res = {0: [], 1: [], 2: []}
            for _ in range(5):
                for ring in (0, 1, 2):
                    _lib.tuning_set("gemm_ring", ring)
                    res[ring].append(timeit(lambda: ops.gemm_nt(A, W, out, M)))
            _lib.tuning_set("gemm_ring", 2)
            med = lambda v: sorted(v)[len(v) // 2]
            wgs = -(-M // 128) * (N // 128)
            print(f"M={M:6d} N={N:5d} K={K:5d} ({wgs:4d} workgroups): two-buffer {med(res[0]):7.1f} us  ring {med(res[1]):7.1f} us  half-height tiles + ring {med(res[2]):7.1f} us  "
                  f"({2.0 * M * N * K / med(res[2]) / 1e6:6.1f} TFLOP/s)", flush=True)


if __name__ == "__main__":
    main()
