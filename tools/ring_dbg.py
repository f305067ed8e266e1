"""Ring vs register-staged attention kernels, per sequence (debugging aid of round 5): max |difference| of the forward output and of dq for
each sequence length of a mixed batch, hd 64 and 32, 128- and 64-row tiles.   python tools/ring_dbg.py   (GPU box)"""
import sys, torch
import os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from avsiam_amd import ops, _lib
dev = 'cuda'
for H, hd in ((12, 64), (16, 32)):
    for tr in (128, 64):
        D = H * hd
        lens = [39, 128, 177, 512, 1, 65, 2472]
        rows = sum(lens); rp = ops.pad_rows(rows)
        torch.manual_seed(0)
        qkv = torch.zeros(rp, 3 * D, device=dev, dtype=torch.bfloat16)
        x = torch.randn(rows, 3 * D, device=dev); x[:, :D] *= ops.attn_q_scale(hd); qkv[:rows] = x.bfloat16()
        dout = torch.zeros(rp, D, device=dev, dtype=torch.bfloat16); dout[:rows] = torch.randn(rows, D, device=dev).bfloat16()
        tiles = ops.AttnTiles(lens, dev, tile_rows=tr)
        res = {}
        for ring in (0, 1):
            _lib.tuning_set("attn_ring", ring)
            out = torch.zeros(rp, D, device=dev, dtype=torch.bfloat16); lse = torch.zeros(H, rp, device=dev)
            ops.attn_fwd(qkv, tiles, H, out, lse)
            dq = torch.zeros_like(qkv); delta = torch.zeros_like(lse)
            ops.attn_bwd(qkv, tiles, H, out if ring == 0 else res[0][0], dout, lse if ring == 0 else res[0][1], delta, dq)
            torch.cuda.synchronize()
            res[ring] = (out, lse, dq, delta)
        r0 = 0
        msg = f"H={H} hd={hd} tile={tr}:"
        for L in lens:
            sl = slice(r0, r0 + L)
            msg += f"  L={L}: out {float((res[0][0][sl].float() - res[1][0][sl].float()).abs().max()):.3g} dq {float((res[0][2][sl, :D].float() - res[1][2][sl, :D].float()).abs().max()):.3g}"
            r0 += L
        print(msg, " |out| staged", float(res[0][0][:rows].float().abs().mean()), "ring", float(res[1][0][:rows].float().abs().mean()), flush=True)
