#!/usr/bin/env python3
"""Runs one GEMM shape a few times (for rocprofv3 --pmc runs).  usage: pmc_gemm.py K N act pp [tile]"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from avsiam_amd import _lib, ops  # noqa: E402

K, N, act, pp = (int(x) for x in sys.argv[1:5])
tile = int(sys.argv[5]) if len(sys.argv) > 5 else 256
M = 95630
lib = _lib.load()
lib.avs_gemm_set_tile(tile)
lib.avs_gemm_set_persistent(pp)
dev = "cuda"
Mp = ops.pad_rows(M, 256)
A = torch.zeros(Mp, K, device=dev, dtype=torch.bfloat16); A[:M] = torch.randn(M, K, device=dev).bfloat16()
W = (torch.randn(N, K, device=dev) * 0.05).bfloat16()
bias = torch.randn(N, device=dev)
out = torch.zeros(Mp, N, device=dev, dtype=torch.bfloat16)
out2 = torch.zeros(Mp, N, device=dev, dtype=torch.bfloat16) if act == 1 else None
aux = torch.randn(Mp, N, device=dev).bfloat16() if act == 2 else None
for _ in range(6):
    ops.gemm_nt(A, W, out, M, bias=bias, out2=out2, aux=aux, act=act)
torch.cuda.synchronize()
