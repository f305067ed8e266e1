#!/usr/bin/env python3
"""The reference's entry point on synthetic batches at a given per-GPU batch, eager against --graph-step: wall seconds per step of the whole loop
(data copy, step, meters).   python tools/train_loop_rate.py [--batch 4] [--steps 300]"""
import argparse
import os
import subprocess
import sys
import tempfile
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--batch", type=int, default=4)
    ap.add_argument("--steps", type=int, default=300)
    args = ap.parse_args()
    for flag in ([], ["--graph-step"], [], ["--graph-step"]):
        times = []
        for steps in (20, 20 + args.steps):                 # two runs: the difference removes start-up, warm-up, validation and checkpointing
            with tempfile.TemporaryDirectory() as d:
                cmd = [sys.executable, "-m", "avsiam_amd.run_cavmae_pretrain_base", "--batch-size", str(args.batch), "--lr", "2e-4", "--frames", "1",
                       "--steps-per-epoch", str(steps), "--n-epochs", "1", "--n-print-steps", "100000", "--exp-dir", d] + flag
                t0 = time.time()
                subprocess.run(cmd, cwd=ROOT, check=True, stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
                times.append(time.time() - t0)
        dt = (times[1] - times[0]) / args.steps
        print(f"batch {args.batch} {'--graph-step' if flag else 'eager      '}: {dt * 1e3:6.2f} ms/step  {args.batch / dt:7.1f} samples/s", flush=True)


if __name__ == "__main__":
    main()
