#!/usr/bin/env python3
"""LayerNorm backward rate against the relative placement of its three operand matrices (same offsets from 2-MiB-aligned bases = same DRAM
channel / bank for the same row index?).   python tools/ln_offset_probe.py"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from avsiam_amd import ops  # noqa: E402

BF16 = torch.bfloat16


def timeit(fn, iters=20):
    fn(); fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters * 1e-3


def carve(nbytes_off, rows, D, dtype, dev):
    n = rows * D
    esz = torch.tensor([], dtype=dtype).element_size()
    raw = torch.zeros(n * esz + (8 << 20), device=dev, dtype=torch.uint8)
    base = (-raw.data_ptr()) % (2 << 20)                # 2-MiB aligned, then the offset under test
    v = raw[base + nbytes_off: base + nbytes_off + n * esz].view(dtype).view(rows, D)
    return v, raw


def main():
    dev = "cuda"
    rows, D = 95630, 768
    rp = ops.pad_rows(rows, 128)
    for offs in ((0, 0, 0, 0), (0, 512, 1024, 1536), (0, 4096 + 256, 8192 + 512, 12288 + 768), (0, 65536 + 1024, 131072 + 2048, 196608 + 3072),
                 (0, 1 << 20, 512 << 10, 256 << 10)):
        keep = []
        x, r = carve(offs[0], rp, D, torch.float32, dev); keep.append(r); x.normal_()
        dy, r = carve(offs[1], rp, D, BF16, dev); keep.append(r); dy.normal_()
        dres, r = carve(offs[2], rp, D, BF16, dev); keep.append(r); dres.normal_()
        dxb, r = carve(offs[3], rp, D, BF16, dev); keep.append(r)
        g0, g1 = torch.randn(D, device=dev), torch.randn(D, device=dev)
        dg0, db0, dg1, db1, dcol = (torch.zeros(D, device=dev) for _ in range(5))
        mean, rstd = torch.zeros(rp, device=dev), torch.ones(rp, device=dev)
        mod = (torch.arange(rows, device=dev) >= rows // 4).to(torch.uint8)
        ws = torch.zeros(ops.layernorm_ws(rows, D), device=dev)
        t = timeit(lambda: ops.layernorm_bwd(dy, x, mean, rstd, g0, None, dg0, db0, ws, rows, g1, dg1, db1, mod, None, dres, dxb, dcol))
        print(f"offsets {offs}: {t * 1e6:7.1f} us {rows * D * 10.0 / t / 1e9:6.0f} GB/s   (x % 2MiB = {x.data_ptr() % (2 << 20)}, dy {dy.data_ptr() % (2 << 20)}, "
              f"dres {dres.data_ptr() % (2 << 20)}, dx {dxb.data_ptr() % (2 << 20)})", flush=True)


if __name__ == "__main__":
    main()
