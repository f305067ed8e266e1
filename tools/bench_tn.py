#!/usr/bin/env python3
"""wgrad (tn) GEMM: 8-phase vs two-buffer kernel at the pass-1 shapes, interleaved in one process."""
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
    dev = "cuda"
    lib = _lib.load()
    M = int(sys.argv[1]) if len(sys.argv) > 1 else 95630
    Mp = ops.pad_rows(M, 64)
    shapes = [(2304, 768), (3072, 768), (768, 3072), (768, 768)] if M < 120000 else [(1536, 512), (2048, 512), (512, 2048), (512, 512)]
    ops_ = {}
    for N1, N2 in shapes:
        A = torch.zeros(Mp, N1, device=dev, dtype=torch.bfloat16); A[:M] = (torch.randn(M, N1, device=dev) * 0.5).bfloat16()
        B = torch.zeros(Mp, N2, device=dev, dtype=torch.bfloat16); B[:M] = (torch.randn(M, N2, device=dev) * 0.5).bfloat16()
        C = torch.zeros(N1, N2, device=dev)
        ops_[(N1, N2)] = (lambda A=A, B=B, C=C: ops.gemm_tn(A, B, C, M))
    for rnd in range(3):
        for v in (0, 1, 2):
            lib.avs_gemm_set_nt8(1 if v else 0)
            lib.avs_gemm_set_tile(256 if v == 2 else 0)
            print(("8-phase forced-256 " if v == 2 else "8-phase " if v else "2-buffer") + "  " + "  ".join(f"{k[0]}x{k[1]}: {timeit(f) * 1e6:.1f} us ({2.0 * M * k[0] * k[1] / timeit(f) / 1e12:.0f} TF/s)" for k, f in ops_.items()), flush=True)
    lib.avs_gemm_set_nt8(1)
    lib.avs_gemm_set_tile(0)


if __name__ == "__main__":
    main()
