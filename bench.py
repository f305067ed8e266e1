#!/usr/bin/env python3
"""bench.py - AV pre-training samples/s of the AVSiam hot path on N MI355X (one process per GPU).

    python bench.py --gpus 1 --steps 10 --warmup 3
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P \\
        bench.py --gpus N --steps K --warmup W

A step = one full reference step (/root/reference/src/traintest_cavmae_base.py:131-152) on one per-GPU batch:
contrastive pass fwd+bwd -> grad all-reduce -> Adam#1 -> MAE pass fwd+bwd -> grad all-reduce -> Adam#2, including
drawing the mask plans.  Inputs are synthetic AudioSet-shaped tensors already resident in HBM.  Workload =
BASELINE.json configs[1]: ViT-B/16, 10 frames x 196 + 512 audio tokens, 75 % mask, batch 64 per GPU, bf16 MFMA
operands with fp32 accumulation.  Weak scaling (per-GPU batch fixed).  Prints ONE JSON line on rank 0.
"""
import argparse
import json
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

PEAK_BF16_TFLOPS = 2500.0      # dense bf16 MFMA peak, /opt/skills/guides/MI355X_MICROARCH.md (chip-level parameters)


def cpu_baseline(cfg, seconds_budget=30.0):
    """The oracle (oracle/ref_cpu.py, a validated CPU restatement of the reference) timed on this node's host
    cores on a bounded sample of the same workload: one full step (both passes fwd+bwd + both Adam updates) at B=2."""
    from avsiam_amd.flops import gflop_per_sample
    from avsiam_amd.maskplan import make_contrastive_plan, make_mae_plan
    from avsiam_amd.weights import synth_inputs, synth_state
    from oracle import ref_cpu
    import random
    cores = os.cpu_count() or 1
    try:
        cores = len(os.sched_getaffinity(0))
    except Exception:
        pass
    cores = min(cores, int(os.environ.get("AVSIAM_CPU_THREADS", 16)))     # a 1-GPU box's CPU share is 16 cores
    torch.set_num_threads(cores)
    B = 2
    P = {k: t.clone().requires_grad_(True) for k, t in synth_state(cfg, 0, "init", include_dead=False).items()}
    a, v = synth_inputs(cfg, B, 87)
    gen = torch.Generator().manual_seed(0)
    pm, pc = make_mae_plan(cfg, B, gen), make_contrastive_plan(cfg, B, gen, random.Random(0))
    params = list(P.values())
    opt1 = torch.optim.Adam(params, 2e-4, weight_decay=5e-7, betas=(0.95, 0.999))
    opt2 = torch.optim.Adam(params, 2e-4, weight_decay=5e-7, betas=(0.95, 0.999))
    t0 = time.time()
    out = ref_cpu.forward(P, cfg, a, v, pc, mae_loss_weight=0, contrast_loss_weight=1)
    opt1.zero_grad(); out[0].backward(); opt1.step()
    out = ref_cpu.forward(P, cfg, a, v, pm, mae_loss_weight=1, contrast_loss_weight=0)
    opt2.zero_grad(); out[0].backward(); opt2.step()
    dt = time.time() - t0
    return {"value": B / dt, "unit": "samples/s", "cores": cores, "kind": "port",
            "sample": f"1 full step (contrastive + MAE fwd/bwd, 2x Adam) at batch {B} of the same config, fp32 torch CPU, {dt:.1f} s",
            "gflops": B * gflop_per_sample(cfg, B) / dt}


def pmc_traffic(args):
    """HBM bytes per launch of the dominant kernel, measured OFFLINE with rocprofv3 PMC passes of this same command
    (FETCH_SIZE / WRITE_SIZE in separate runs, FETCH doubled per MI355X_MICROARCH.md) and stored under profiles/;
    null when the stored measurement is for a different workload."""
    path = os.path.join(ROOT, "profiles", "r01", "traffic_gemm_nt.json")
    if not os.path.exists(path) or args.batch != 64 or args.frames != 10 or args.audio_tokens != 512 or args.model != "vit_base":
        return None
    try:
        with open(path) as f:
            return float(json.load(f)["hbm_bytes_per_launch"])
    except Exception:
        return None


def log(msg):
    if int(os.environ.get("RANK", 0)) == 0:
        print(f"[bench {time.strftime('%H:%M:%S')}] {msg}", file=sys.stderr, flush=True)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--batch", type=int, default=64, help="per-GPU batch")
    ap.add_argument("--frames", type=int, default=10)
    ap.add_argument("--audio-tokens", type=int, default=512)
    ap.add_argument("--model", choices=("vit_base", "vit_large"), default="vit_base",
                    help="vit_base = BASELINE.json's metric (configs[1]); vit_large = configs[3]'s shape, an extra data point")
    ap.add_argument("--lr", type=float, default=2e-4)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-kernel-events", action="store_true")
    ap.add_argument("--all-kernel-events", action="store_true",
                    help="HIP-event timing of every kernel family (default: the dominant kernel, gemm_nt, only - each timed "
                         "launch costs the stream ~5 us)")
    args = ap.parse_args()

    from avsiam_amd import _lib, ops
    from avsiam_amd.config import AVSiamConfig
    from avsiam_amd.flops import gflop_per_sample
    from avsiam_amd.models import CAVMAE_BASE
    from avsiam_amd.traintest_cavmae_base import train_step

    world = int(os.environ.get("WORLD_SIZE", 1))
    rank = int(os.environ.get("RANK", 0))
    local = int(os.environ.get("LOCAL_RANK", 0))
    if world != args.gpus:
        if world == 1 and args.gpus > 1:
            raise SystemExit("launch with torch.distributed.run --nproc-per-node N for --gpus N")
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU: the hot path runs only on the HIP kernels")
    _lib.load()
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)
    import torch.distributed as dist
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)

    if args.model == "vit_large":
        from avsiam_amd.config import vit_large
        cfg = vit_large(audio_tokens=args.audio_tokens, frames=args.frames)
    else:
        cfg = AVSiamConfig(audio_tokens=args.audio_tokens, frames=args.frames)
    mname = "ViT-B/16" if args.model == "vit_base" else "ViT-L/16"
    torch.manual_seed(87 + rank)
    log(f"building model (frames={args.frames}, batch={args.batch}/GPU, world={world})")
    model = CAVMAE_BASE(cfg=cfg, verbose=False, plan_seed=87 + rank).to(dev)
    model.set_distributed(world, rank)
    model.publish_grads = False
    from avsiam_amd.weights import synth_inputs
    a, v = synth_inputs(cfg, args.batch, 87 + rank)
    a, v = a.to(dev), v.to(dev)

    def sync():
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    log("inputs resident; warm-up")
    for i in range(args.warmup):
        last = train_step(model, a, v, args.lr)
        torch.cuda.synchronize()
        log(f"warm-up step {i} done, mem {torch.cuda.max_memory_allocated() / 2**30:.1f} GiB")
    sync()
    if not args.no_kernel_events:
        # default: every 5th gemm_nt launch (282 per step, co-prime to 5: over 5 or 10 steps every launch position is timed
        # equally often); --all-kernel-events: every launch of every kernel family (costs ~5 ms/step of event overhead)
        ops.prof = ops.KernelProfiler() if args.all_kernel_events else ops.KernelProfiler(("gemm_nt",), stride=5)
    t0 = time.perf_counter()
    for _ in range(args.steps):
        last = train_step(model, a, v, args.lr)
    sync()
    dt = time.perf_counter() - t0
    log(f"timed region: {args.steps} steps in {dt:.3f} s")
    prof, ops.prof = ops.prof, None
    tmax = torch.tensor([dt], device=dev, dtype=torch.float64)
    if world > 1:
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
    dt = float(tmax.item())
    losses = [float(x.item()) for x in last]
    if rank == 0:
        sps = world * args.batch * args.steps / dt
        gf = gflop_per_sample(cfg, args.batch)
        line = {
            "metric": f"AV pretrain samples/sec ({mname}, 75% mask)", "value": sps, "unit": "samples/s", "n_gpus": world,
            "steps": args.steps, "warmup": args.warmup, "ms_per_step": 1e3 * dt / args.steps, "higher_is_better": True,
            "scaling": "weak", "vs_baseline": None, "dtype": "bf16", "data": "synthetic",
            "config": {"workload": f"AVSiam pretrain step (contrastive + MAE passes, 2x Adam), {mname}, {args.frames} frames x196 + "
                                   f"{args.audio_tokens} audio tokens, 75% mask, batch {args.batch}/GPU",
                       "global_batch": world * args.batch, "frames": args.frames, "audio_tokens": args.audio_tokens,
                       "parallelism": f"dp{world}", "gflop_per_sample": gf},
            "model_tflops": sps * gf / 1e3, "mfu_vs_dense_bf16_peak": sps * gf / 1e3 / (world * PEAK_BF16_TFLOPS),
            "final_losses": {"loss_mae": losses[0], "loss_mae_a": losses[1], "loss_mae_v": losses[2], "loss_c": losses[3], "c_acc": losses[4]},
        }
        if prof is not None:
            s = prof.summary()
            mm = [k for k in s if k.startswith("gemm_nt")]
            flops = sum(s[k]["work"] for k in mm)
            ms = sum(s[k]["total_ms"] for k in mm)
            n = sum(s[k]["launches"] for k in mm)
            nd = sum(s[k]["dispatches"] for k in mm)
            ach = flops / (ms * 1e-3) / 1e12
            line["roofline"] = {"bound": "mfma", "kernel": "gemm_nt8_kernel / gemm_nt_kernel (forward + dgrad bf16 MFMA GEMMs, one avs_gemm_nt_bf16 call = one launch)", "achieved": ach,
                                "peak": PEAK_BF16_TFLOPS, "unit": "TFLOP/s", "frac": ach / PEAK_BF16_TFLOPS, "traffic": pmc_traffic(args),
                                "launches": n, "avg_launch_us": 1e3 * ms / n, "flops_per_launch": flops / n,
                                "dispatches": nd, "avg_dispatch_us": 1e3 * ms / nd,
                                "sampling": "all launches" if args.all_kernel_events else "every 5th launch of the timed region"}
            line["kernels"] = {k: {"launches": x["launches"], "total_ms": round(x["total_ms"], 3), "avg_us": round(x["avg_us"], 2),
                                   "rate_T_per_s": round(x["rate"] / 1e12, 3)} for k, x in sorted(s.items())}
        if world == 1 and not args.no_cpu_baseline:
            log("timing the CPU baseline (oracle) on the host cores")
            try:
                line["cpu_baseline"] = cpu_baseline(cfg)
            except Exception as e:                                       # the baseline is a report, never a gate
                line["cpu_baseline"] = {"value": None, "error": repr(e)}
        print(json.dumps(line), flush=True)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
