#!/usr/bin/env python3
"""A few dozen training steps at the bench shape on fixed synthetic data: the losses must stay finite and fall (the model
memorises the one batch) - a quick end-to-end check of the whole step (both passes, both Adam updates, device-drawn mask
plans, grouped towers, second-stream weight gradients).   python tools/train_sanity.py [--steps 40]"""
import argparse
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from avsiam_amd.config import AVSiamConfig  # noqa: E402
from avsiam_amd.models import CAVMAE_BASE  # noqa: E402
from avsiam_amd.traintest_cavmae_base import train_step  # noqa: E402
from avsiam_amd.weights import synth_inputs  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--steps", type=int, default=40)
    ap.add_argument("--batch", type=int, default=64)
    ap.add_argument("--frames", type=int, default=10)
    ap.add_argument("--lr", type=float, default=2e-4)
    args = ap.parse_args()
    cfg = AVSiamConfig(frames=args.frames)
    m = CAVMAE_BASE(cfg=cfg, verbose=False, plan_seed=1).cuda()
    m.publish_grads = False
    a, v = synth_inputs(cfg, args.batch, 3)
    a, v = a.cuda(), v.cuda()
    first = last = None
    for i in range(args.steps):
        out = [float(x.item()) for x in train_step(m, a, v, args.lr)]
        assert all(x == x and abs(x) < 1e6 for x in out), (i, out)
        if i % 5 == 0 or i == args.steps - 1:
            print(f"step {i:3d}: loss_mae {out[0]:.4f} (a {out[1]:.4f} v {out[2]:.4f})  loss_c {out[3]:.4f}  c_acc {out[4]:.3f}", flush=True)
        first = first or out
        last = out
    assert last[0] < first[0] and last[3] < first[3], (first, last)
    if hasattr(m, "fp8_saturation_events") and os.environ.get("AVSIAM_FP8", "0") != "0":
        print(f"fp8 mode {os.environ['AVSIAM_FP8']}: saturation events over {args.steps} steps: {m.fp8_saturation_events():.0f}", flush=True)
    print("ok: losses fell", flush=True)


if __name__ == "__main__":
    main()
