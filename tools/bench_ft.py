#!/usr/bin/env python3
"""Throughput of the fine-tuned model's inference modes (CAVMAEFT_BASE, forward only) and, with --errors, the measured
error of every golden case against the reference's vectors.

    python tools/bench_ft.py [--batch 64] [--frames 10] [--iters 5] [--errors]
"""
import argparse
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from avsiam_amd.config import AVSiamConfig  # noqa: E402
from avsiam_amd.models import CAVMAEFT_BASE  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--batch", type=int, default=64)
    ap.add_argument("--frames", type=int, default=10)
    ap.add_argument("--iters", type=int, default=5)
    ap.add_argument("--errors", action="store_true")
    args = ap.parse_args()
    cfg = AVSiamConfig()
    if args.errors:
        from tests.helpers import FT_CASES, ft_case_inputs, ft_outputs_as_dict, load_golden
        m = CAVMAEFT_BASE(527, init_seed=4321, init_mode="random").cuda()
        for name in FT_CASES:
            d = load_golden(name)
            a, v = ft_case_inputs(d, cfg)
            out = m(a.cuda(), v.cuda(), str(d["mode"]), is_eval=bool(d["is_eval"]))
            for k, t in ft_outputs_as_dict(d, out).items():
                if k.startswith("tokens"):
                    continue
                g, r = t.double().cpu().reshape(-1, 527), torch.from_numpy(d[k]).double().reshape(-1, 527)
                cos = torch.nn.functional.cosine_similarity(g, r, dim=1).min().item()
                print(f"{name:14s} {k:6s} max|err| {float((g - r).abs().max()):.4f}  min row cosine {cos:.6f}  (sigma {float(r.std()):.3f})", flush=True)
    B, T = args.batch, args.frames
    m = CAVMAEFT_BASE(527).cuda()
    a = torch.randn(B, cfg.audio_len, cfg.n_mels, device="cuda")
    v = torch.randn(B, T, 3, cfg.img_size, cfg.img_size, device="cuda")
    v1 = v[:, :1].contiguous()
    for name, fn in (("audioonly", lambda: m(a, None, "audioonly")), ("videoonly", lambda: m(None, v, "videoonly")),
                     ("retrieval", lambda: m(a, v, "retrieval")), ("mm_grad (1 frame)", lambda: m(a, v1, "mm_grad")),
                     ("mm_grad eval (10 frames)", lambda: m(a, v, "mm_grad", is_eval=True))):
        if "eval" in name and T != 10 or name == "retrieval" and T <= 5:
            continue
        fn(); torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(args.iters):
            fn()
        torch.cuda.synchronize()
        dt = (time.perf_counter() - t0) / args.iters
        print(f"{name:26s} B={B} T={T if 'video' in name or 'eval' in name or name == 'retrieval' else 1}: {dt * 1e3:8.2f} ms/batch  {B / dt:9.1f} clips/s", flush=True)


if __name__ == "__main__":
    main()
