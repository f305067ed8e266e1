#!/usr/bin/env python3
"""Finite-difference probe of the training losses at full size (which step sizes a gradient check can use):
central differences along the fp32-parameter part of the gradient for several step sizes.   python tools/fd_probe.py"""
import math
import os
import random
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from avsiam_amd.config import AVSiamConfig  # noqa: E402
from avsiam_amd.maskplan import make_contrastive_plan, make_mae_plan  # noqa: E402
from avsiam_amd.models import CAVMAE_BASE  # noqa: E402
from avsiam_amd.param_spec import P1, P2, build_spec  # noqa: E402
from avsiam_amd.weights import synth_inputs  # noqa: E402

B, T = 64, 10
cfg = AVSiamConfig(frames=T)
m = CAVMAE_BASE(cfg=cfg, init_seed=11, init_mode="random", verbose=False).cuda()
m.publish_grads = False
a, v = synth_inputs(cfg, B, 5)
a, v = a.cuda(), v.cuda()
for which in (P2, P1):
    gen = torch.Generator().manual_seed(6)
    plan = make_mae_plan(cfg, B, gen) if which == P2 else make_contrastive_plan(cfg, B, gen, random.Random(6))
    kw = dict(mae_loss_weight=1 if which == P2 else 0, contrast_loss_weight=0 if which == P2 else 1, mask_plan=plan)
    out = m(a, v, **kw)
    out[0].backward()
    lo, hi = m.arena.range[which]
    g = m.arena.g[lo:hi].clone()
    sel = torch.zeros(hi - lo)
    for info in build_spec(cfg):
        if info.live & which and info.kind in ("bias", "ln_w", "ln_b"):
            o = m.arena.offset[info.name] - lo
            sel[o:o + math.prod(info.shape)] = 1.0
    g = g * sel.cuda()
    gn = float(g.double().norm())
    d = (g / gn).float()
    w0 = m.arena.p[lo:hi].clone()
    print(f"pass {which}: loss {out[0].item():.6f}  |g_fp32params| {gn:.6f}", flush=True)
    prev = None
    for frac in (5e-3, 2.5e-3, 1e-3, 5e-4, 2.5e-4, 1e-4):
        eps = frac * abs(out[0].item()) / gn
        vals = []
        for sgn in (1.0, -1.0):
            m.arena.p[lo:hi].copy_(w0 + sgn * eps * d)
            m.mark_weights_changed()
            with torch.no_grad():
                vals.append(m(a, v, **kw)[0].item())
        fd = (vals[0] - vals[1]) / (2 * eps)
        rich = (4 * fd - prev) / 3 if prev is not None and False else None
        print(f"   step {frac:7.5f}: L+ {vals[0]:.6f} L- {vals[1]:.6f}  fd {fd:.6f}  fd/|g| {fd / gn:.4f}", flush=True)
        prev = fd
    m.arena.p[lo:hi].copy_(w0)
    m.mark_weights_changed()
