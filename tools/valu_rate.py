#!/usr/bin/env python3
"""Issue cost (shader cycles per wave64 instruction per SIMD) of the VALU instructions the attention softmax is made of.
    python tools/valu_rate.py"""
import ctypes
import os
import subprocess

import torch

HERE = os.path.dirname(os.path.abspath(__file__))


def main():
    src, so = os.path.join(HERE, "csrc", "valu_rate.hip"), os.path.join(HERE, "csrc", "valu_rate.so")
    if not os.path.exists(so) or os.path.getmtime(so) < os.path.getmtime(src):
        subprocess.check_call(["hipcc", "--offload-arch=gfx950", "-O2", "-shared", "-fPIC", src, "-o", so])
    lib = ctypes.CDLL(so)
    lib.valu_rate.argtypes = [ctypes.c_int, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_void_p]
    blocks, iters = 256, 4096                                   # one workgroup per CU of 4 / 8 / 16 waves: 1 / 2 / 4 waves per SIMD
    out = torch.zeros(blocks * 1024, device="cuda")
    cyc = torch.zeros(blocks, 2, dtype=torch.int64, device="cuda")
    for kind, name in ((1, "v_mul_f32"), (7, "v_fma_f32"), (2, "v_max3_f32"), (3, "v_cvt_pk_bf16_f32"), (4, "v_pk_mul_f32 (2 elements)"),
                       (5, "v_pk_add_f32 (2 elements)"), (0, "v_exp_f32"), (6, "v_rcp_f32")):
        res = []
        for wps in (1, 2, 4):
            for _ in range(2):
                cyc[:, 0] = torch.iinfo(torch.int64).max
                cyc[:, 1] = 0
                lib.valu_rate(kind, out.data_ptr(), cyc.data_ptr(), blocks, 256 * wps, iters, torch.cuda.current_stream().cuda_stream)
            torch.cuda.synchronize()
            res.append((cyc[:, 1] - cyc[:, 0]).double().mean().item() / (iters * 16 * wps))      # cycles of the SIMD per instruction it issued
        print(f"{name:28s} SIMD cycles per wave64 instruction with 1 / 2 / 4 waves per SIMD: " + " / ".join(f"{c:5.2f}" for c in res), flush=True)


if __name__ == "__main__":
    main()
