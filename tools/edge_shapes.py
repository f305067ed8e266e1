#!/usr/bin/env python3
"""Edge shapes through both passes (finite losses, gradients present) and against the oracle where cheap: batch 1, odd batches, several frames, recompute
fractions, the shared activation pool, the deterministic mode - with the round-6 defaults (dead-row pruning, 8-bit gelu').  usage: python tools/edge_shapes.py"""
import os
import random
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from avsiam_amd.config import AVSiamConfig, EngineOptions, vit_large  # noqa: E402
from avsiam_amd.maskplan import make_contrastive_plan, make_mae_plan  # noqa: E402
from avsiam_amd.models import CAVMAE_BASE, CAVMAE_LARGE  # noqa: E402
from avsiam_amd.traintest_cavmae_base import train_step  # noqa: E402
from avsiam_amd.weights import synth_inputs, synth_state  # noqa: E402
from oracle import ref_cpu  # noqa: E402


def main():
    torch.set_num_threads(16)
    cases = [("B1_T1", AVSiamConfig(audio_tokens=128), 1, {}), ("B2_T1", AVSiamConfig(audio_tokens=128), 2, {}), ("B7_T3", AVSiamConfig(audio_tokens=128, frames=3), 7, {}),
             ("B6_T2_rc05_pool", AVSiamConfig(audio_tokens=128, frames=2), 6, {"options": EngineOptions(recompute="0.5"), "share_pass_buffers": True}),
             ("B5_T1_det", AVSiamConfig(audio_tokens=512), 5, {"deterministic": True}), ("B3_T2_fp32stream", AVSiamConfig(audio_tokens=128, frames=2), 3, {"grad_stream": "fp32"}),
             ("L_B3_T2", vit_large(audio_tokens=128, frames=2, depth=3), 3, {"cls": CAVMAE_LARGE})]
    bad = 0
    only = sys.argv[1:]
    for name, cfg, B, kw in cases:
        if only and name not in only:
            continue
        print(f"{name}: start", flush=True)
        cls = kw.pop("cls", CAVMAE_BASE)
        a, v = synth_inputs(cfg, B, 5)
        m = cls(cfg=cfg, init_seed=3, init_mode="random", verbose=False, plan_seed=1, **kw).cuda()
        m.publish_grads = False
        outs = [[float(x.item()) for x in train_step(m, a.cuda(), v.cuda(), 1e-4)] for _ in range(2)]
        fin = all(x == x and abs(x) < 1e4 for o in outs for x in o)
        # against the oracle: one MAE forward with an injected plan
        m.publish_grads = True
        gen = torch.Generator().manual_seed(2)
        plan = make_mae_plan(cfg, B, gen)
        P = {k: p.detach().cpu().clone() for k, p in m._params.items() if k in synth_state(cfg, 3, "random", include_dead=False)}
        with torch.no_grad():
            out = m(a.cuda(), v.cuda(), mae_loss_weight=1, contrast_loss_weight=0, mask_plan=plan)
            ref = ref_cpu.forward(P, cfg, a, v, plan, mae_loss_weight=1, contrast_loss_weight=0)
        rel = abs(out[0].item() - ref[0].item()) / abs(ref[0].item())
        eng = m._engine("mae", B)
        ok = fin and rel < 2e-3
        bad += not ok
        print(f"{name:20s} prune={eng.prune} lq={eng.st_dec.lq} losses {outs[-1][:4]} mae vs oracle rel {rel:.2e} {'ok' if ok else 'FAIL'}", flush=True)
        del m
        torch.cuda.empty_cache()
    sys.exit(1 if bad else 0)


if __name__ == "__main__":
    main()
