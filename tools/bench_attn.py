#!/usr/bin/env python3
"""Attention throughput vs sequence length (random data, one process).  usage: bench_attn.py"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from avsiam_amd import ops  # noqa: E402


def timeit(fn, iters=5):
    fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters * 1e-3


def run_case(name, H, hd, lens, dev, tile_rows=(128, 64)):
    D = H * hd
    rows = sum(lens)
    rp = ops.pad_rows(rows)
    qkv = torch.zeros(rp, 3 * D, device=dev, dtype=torch.bfloat16)
    qkv[:rows] = torch.randn(rows, 3 * D, device=dev).bfloat16()
    out = torch.zeros(rp, D, device=dev, dtype=torch.bfloat16)
    lse = torch.zeros(H, rp, device=dev)
    dout = torch.zeros(rp, D, device=dev, dtype=torch.bfloat16)
    dout[:rows] = torch.randn(rows, D, device=dev).bfloat16()
    dqkv = torch.zeros_like(qkv)
    delta = torch.zeros_like(lse)
    fl = 4.0 * sum(L * L for L in lens) * D
    msg = f"{name:34s} rows={rows:6d}:"
    for tr in tile_rows:
        tiles = ops.AttnTiles(lens, dev, tile_rows=tr)
        if "--ring-ab" in sys.argv:              # register-staged vs LDS-DMA ring kernels, interleaved rounds in this one process (median)
            from avsiam_amd import _lib
            res = {0: ([], []), 1: ([], [])}
            for _ in range(5):
                for ring in (0, 1):
                    _lib.tuning_set("attn_ring", ring)
                    res[ring][0].append(timeit(lambda: ops.attn_fwd(qkv, tiles, H, out, lse), 10))
                    res[ring][1].append(timeit(lambda: ops.attn_bwd(qkv, tiles, H, out, dout, lse, delta, dqkv), 10))
            _lib.tuning_set("attn_ring", 0)
            med = lambda v: sorted(v)[len(v) // 2]
            msg += (f"  [tile {tr}] fwd staged {med(res[0][0])*1e6:7.1f} ring {med(res[1][0])*1e6:7.1f} us ({fl/med(res[1][0])/1e12:5.0f} TF/s)"
                    f"  bwd staged {med(res[0][1])*1e6:7.1f} ring {med(res[1][1])*1e6:7.1f} us ({2.5*fl/med(res[1][1])/1e12:5.0f} TF/s)")
            continue
        tf = timeit(lambda: ops.attn_fwd(qkv, tiles, H, out, lse), 10)
        tb = timeit(lambda: ops.attn_bwd(qkv, tiles, H, out, dout, lse, delta, dqkv), 10)
        msg += (f"  [tile {tr}] fwd {tf*1e6:7.1f} us {fl/tf/1e12:6.1f} TF/s {rows*8.0*D/tf/1e9:6.0f} GB/s"
                f"  bwd {tb*1e6:7.1f} us {2.5*fl/tb/1e12:6.1f} TF/s {rows*24.0*D/tb/1e9:6.0f} GB/s")
        if "--g8" in sys.argv:          # the fp8 modes' variants: e4m3 copy of the output / e5m2 copy of dqkv written as well (and, lean, no bf16 dK / dV)
            rec = ops.Fp8Records(2, dev, fmax=ops.BF8_MAX)
            rec.q[:, 0], rec.q[:, 1], rec.q[:, 2] = 1000.0, 1e-3, 1e30      # (scale, 1 / scale, amax seen: as in a calibrated step no wave issues the amax atomic)
            o8 = torch.zeros(rp, D, device=dev, dtype=torch.uint8)
            d8 = torch.zeros(rp, 3 * D, device=dev, dtype=torch.uint8)
            tf8 = timeit(lambda: ops.attn_fwd(qkv, tiles, H, out, lse, out8=o8, q8=rec.rec(0)), 10)
            tb8 = timeit(lambda: ops.attn_bwd(qkv, tiles, H, out, dout, lse, delta, dqkv, dqkv8=d8, q8=rec.rec(1)), 10)
            tb8l = timeit(lambda: ops.attn_bwd(qkv, tiles, H, out, dout, lse, delta, dqkv, dqkv8=d8, q8=rec.rec(1), kv_bf16=False), 10)
            msg += f"  [+8-bit copies] fwd {tf8*1e6:7.1f} us  bwd {tb8*1e6:7.1f} us  bwd without bf16 dK/dV {tb8l*1e6:7.1f} us"
    if hd == 80 and min(lens) <= 64:
        # head dim 80: the fused kernel exists for 64-row workgroups only
        f = [ops.AttnSeqs(lens, dev, 0, 64)]
        for tr in tile_rows:
            tl = ops.AttnTiles(lens, dev, tile_rows=tr, min_len=64)

            def mixed80():
                if tl.ntiles:
                    ops.attn_bwd(qkv, tl, H, out, dout, lse, delta, dqkv)
                ops.attn_bwd_fused(qkv, f[0], H, out, dout, lse, dqkv)
            msg += f"  [fused<=64 + tile {tr}] bwd {timeit(mixed80, 10)*1e6:7.1f} us"
            if "--g8" in sys.argv:
                def mixed80l():
                    if tl.ntiles:
                        ops.attn_bwd(qkv, tl, H, out, dout, lse, delta, dqkv, dqkv8=d8, q8=rec.rec(1), kv_bf16=False)
                    ops.attn_bwd_fused(qkv, f[0], H, out, dout, lse, dqkv, dqkv8=d8, q8=rec.rec(1), kv_bf16=False)
                msg += f" (lean e5m2 {timeit(mixed80l, 10)*1e6:7.1f} us)"
    if hd in (32, 64) and min(lens) <= 128:
        # the backward as the engine runs it: sequences of at most 128 tokens through the fused kernel, the rest through the two kernels
        f = [sq for sq in (ops.AttnSeqs(lens, dev, 0, 64), ops.AttnSeqs(lens, dev, 64, 128)) if sq.nseq]
        long_lens = [L for L in lens if L > 128]
        tl = ops.AttnTiles(lens, dev, tile_rows=64 if sum(long_lens) / max(1, len(long_lens)) < 256 else 128, min_len=128)

        def mixed():
            if tl.ntiles:
                ops.attn_bwd(qkv, tl, H, out, dout, lse, delta, dqkv)
            for sq in f:
                ops.attn_bwd_fused(qkv, sq, H, out, dout, lse, dqkv)
        tm = timeit(mixed, 10)
        msg += f"  [fused<=128] bwd {tm*1e6:7.1f} us"
        if hd == 64 and any(128 < L <= 224 for L in lens):
            # round 6: sequences of 129 .. 224 tokens through the 7-wave fused kernel as well (what the engine runs now)
            f2 = f + [ops.AttnSeqs(lens, dev, 128, 224)]
            ll = [L for L in lens if L > 224]
            tl2 = ops.AttnTiles(lens, dev, tile_rows=64 if sum(ll) / max(1, len(ll)) < 256 else 128, min_len=224)

            def mixed2():
                if tl2.ntiles:
                    ops.attn_bwd(qkv, tl2, H, out, dout, lse, delta, dqkv)
                for sq in f2:
                    ops.attn_bwd_fused(qkv, sq, H, out, dout, lse, dqkv)
            tm2 = timeit(mixed2, 10)
            msg += f"  [fused<=224] bwd {tm2*1e6:7.1f} us"
    print(msg, flush=True)


def step_mixes(dev):
    """the sequence mixes of the bench step (batch 64, 10 frames): contrastive pass (five keep ratios), MAE towers, joint layers, decoder"""
    vid = [196] * 130 + [156] * 130 + [117] * 130 + [78] * 130 + [39] * 120
    aud = [512] * 13 + [409] * 13 + [307] * 13 + [204] * 13 + [102] * 12
    if "--decoder-only" in __import__("sys").argv:            # for counter runs (tools/pmc_attn.sh)
        return run_case("decoder (64x2472)", 16, 32, [2472] * 64, dev, tile_rows=(128,))
    run_case("contrastive pass (audio+video)", 12, 64, aud + vid, dev)
    run_case("contrastive pass, video only", 12, 64, vid, dev)
    run_case("contrastive pass, audio only", 12, 64, aud, dev)
    run_case("MAE towers (64x128 + 640x49)", 12, 64, [128] * 64 + [49] * 640, dev)
    run_case("MAE joint layers (64x618)", 12, 64, [618] * 64, dev)
    run_case("decoder (64x2472)", 16, 32, [2472] * 64, dev, tile_rows=(128,))


def huge_mixes(dev):
    """the sequence mixes of the ViT-H/14 step (batch 64, 10 frames; 16 heads of 80): contrastive pass, MAE towers, joint layers, decoder"""
    vid = [256] * 130 + [204] * 130 + [153] * 130 + [102] * 130 + [51] * 120
    aud = [657] * 13 + [525] * 13 + [394] * 13 + [262] * 13 + [131] * 12
    tr = (128, 64) if "--tile64" in sys.argv else (128,)
    run_case("H/14 contrastive pass (audio+video)", 16, 80, aud + vid, dev, tile_rows=tr)
    run_case("H/14 contrastive pass, video only", 16, 80, vid, dev, tile_rows=tr)
    run_case("H/14 MAE towers (64x164 + 640x64)", 16, 80, [164] * 64 + [64] * 640, dev, tile_rows=tr)
    run_case("H/14 MAE joint layers (64x804)", 16, 80, [804] * 64, dev, tile_rows=tr)
    run_case("H/14 decoder (64x3217)", 16, 32, [3217] * 64, dev, tile_rows=(128,))


def main():
    dev = "cuda"
    import sys
    if "--step" in sys.argv:
        return step_mixes(dev)
    if "--huge" in sys.argv:
        return huge_mixes(dev)
    for H, hd, L, rows_target in [(12, 64, 39, 40000), (12, 64, 49, 40000), (12, 64, 78, 40000), (12, 64, 117, 40000), (12, 64, 128, 40000),
                                  (12, 64, 156, 40000), (12, 64, 196, 40000), (12, 64, 307, 40000), (12, 64, 512, 40000), (12, 64, 618, 40000),
                                  (16, 32, 708, 80000), (16, 32, 2472, 160000)]:
        D = H * hd
        nseq = max(1, rows_target // L)
        lens = [L] * nseq
        rows = nseq * L
        rp = ops.pad_rows(rows)
        qkv = torch.zeros(rp, 3 * D, device=dev, dtype=torch.bfloat16)
        qkv[:rows] = torch.randn(rows, 3 * D, device=dev).bfloat16()
        out = torch.zeros(rp, D, device=dev, dtype=torch.bfloat16)
        lse = torch.zeros(H, rp, device=dev)
        dout = torch.zeros(rp, D, device=dev, dtype=torch.bfloat16)
        dout[:rows] = torch.randn(rows, D, device=dev).bfloat16()
        dqkv = torch.zeros_like(qkv)
        delta = torch.zeros_like(lse)
        fl = 4.0 * nseq * L * L * D
        msg = f"H={H} hd={hd} L={L:5d} nseq={nseq:5d}:"
        for tr in (128, 64):
            tiles = ops.AttnTiles(lens, dev, tile_rows=tr)
            tf = timeit(lambda: ops.attn_fwd(qkv, tiles, H, out, lse))
            tb = timeit(lambda: ops.attn_bwd(qkv, tiles, H, out, dout, lse, delta, dqkv))
            msg += f"  [tile {tr}] fwd {tf*1e6:7.1f} us {fl/tf/1e12:6.1f} TF/s bwd {tb*1e6:7.1f} us {2.5*fl/tb/1e12:6.1f} TF/s"
        print(msg, flush=True)


if __name__ == "__main__":
    main()
