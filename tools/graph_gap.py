#!/usr/bin/env python3
"""Per-launch cost of a dependent chain of short kernels: eager launches through the C ABI vs one captured hipGraph."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from avsiam_amd import ops  # noqa: E402


def main():
    dev = "cuda"
    M, K, N = 4096, 768, 768
    A = (torch.randn(M, K, device=dev) * 0.5).bfloat16()
    W = (torch.randn(N, K, device=dev) * 0.03).bfloat16()
    o1 = torch.zeros(M, N, device=dev, dtype=torch.bfloat16)
    o2 = torch.zeros(M, N, device=dev, dtype=torch.bfloat16)
    n = 400

    def chain():
        for _ in range(n // 2):
            ops.gemm_nt(A, W, o1, M)
            ops.gemm_nt(o1, W, o2, M)

    def timed(fn, reps=5):
        fn(); torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(reps):
            fn()
        e1.record(); torch.cuda.synchronize()
        return e0.elapsed_time(e1) / reps * 1e3 / n          # us per launch

    t_eager = timed(chain)
    s = torch.cuda.Stream()
    s.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(s):
        chain()
    torch.cuda.current_stream().wait_stream(s)
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        chain()
    t_graph = timed(g.replay)
    print(f"{n} dependent launches: eager {t_eager:.2f} us/launch, graph replay {t_graph:.2f} us/launch")


if __name__ == "__main__":
    main()
