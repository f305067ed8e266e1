#!/usr/bin/env python3
"""Runs one wgrad shape a few times (for rocprofv3 --pmc runs).  usage: pmc_tn.py N1 N2 [tile]"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from avsiam_amd import _lib, ops  # noqa: E402

N1, N2 = int(sys.argv[1]), int(sys.argv[2])
tile = int(sys.argv[3]) if len(sys.argv) > 3 else 256
M = 95630
lib = _lib.load()
lib.avs_gemm_set_tile(tile)
dev = "cuda"
Mp = ops.pad_rows(M, 256)
A = torch.zeros(Mp, N1, device=dev, dtype=torch.bfloat16); A[:M] = torch.randn(M, N1, device=dev).bfloat16()
B = torch.zeros(Mp, N2, device=dev, dtype=torch.bfloat16); B[:M] = torch.randn(M, N2, device=dev).bfloat16()
C = torch.zeros(N1, N2, device=dev)
for _ in range(6):
    ops.gemm_tn(A, B, C, M)
torch.cuda.synchronize()
