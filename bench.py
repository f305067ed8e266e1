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
PEAK_FP8_TFLOPS = 5000.0       # dense fp8 MFMA peak, same table (fp8 launches are priced against THIS, never against the bf16 peak)
PEAK_HBM_GBS = 8000.0          # HBM3E peak, same table (6.3 TB/s is what a float4 copy achieves)


def _cpu_leg(cfg, batch, warmup, steps, cores, label):
    """`warmup` + `steps` full reference steps (both passes forward + backward + both Adam updates) of the oracle at one shape"""
    import random
    from avsiam_amd.flops import gflop_per_sample
    from avsiam_amd.maskplan import make_contrastive_plan, make_mae_plan
    from avsiam_amd.weights import synth_inputs, synth_state
    from oracle import ref_cpu
    B = batch
    P = {k: t.clone().requires_grad_(True) for k, t in synth_state(cfg, 0, "init", include_dead=False).items()}
    a, v = synth_inputs(cfg, B, 87)
    gen = torch.Generator().manual_seed(0)
    pm, pc = make_mae_plan(cfg, B, gen), make_contrastive_plan(cfg, B, gen, random.Random(0))
    params = list(P.values())
    opt1 = torch.optim.Adam(params, 2e-4, weight_decay=5e-7, betas=(0.95, 0.999))
    opt2 = torch.optim.Adam(params, 2e-4, weight_decay=5e-7, betas=(0.95, 0.999))

    def step():
        out = ref_cpu.forward(P, cfg, a, v, pc, mae_loss_weight=0, contrast_loss_weight=1)
        opt1.zero_grad(); out[0].backward(); opt1.step()
        out = ref_cpu.forward(P, cfg, a, v, pm, mae_loss_weight=1, contrast_loss_weight=0)
        opt2.zero_grad(); out[0].backward(); opt2.step()

    for i in range(warmup):
        t0 = time.time()
        step()
        log(f"cpu baseline [{label}]: warm-up step {i} took {time.time() - t0:.1f} s ({cores} threads)")
    times = []
    for i in range(steps):
        t0 = time.time()
        step()
        times.append(time.time() - t0)
        log(f"cpu baseline [{label}]: timed step {i} took {times[-1]:.1f} s")
    dt = sum(times) / len(times)
    return {"config": label, "value": B / dt, "unit": "samples/s", "batch": B, "frames": cfg.frames, "audio_tokens": cfg.audio_tokens,
            "warmup_steps": warmup, "timed_steps": steps, "step_seconds": [round(t, 3) for t in times],
            "gflop_per_sample": gflop_per_sample(cfg, B), "gflops": B * gflop_per_sample(cfg, B) / dt}


def cpu_baseline(cfg, batch=2, warmup=1, steps=3, survey_legs=True):
    """SURVEY.md 8(d) / BASELINE.md section 3: the oracle (oracle/ref_cpu.py, the CPU restatement of the reference validated against
    it) timed on THIS node's host cores in the same run, fp32 torch, all the cores the process may use - full reference steps (both
    passes forward + backward + both Adam updates), each a bounded sample:
      main  a small batch of the SAME configuration as the GPU line (its `value` is the object's `value`)
      C1    BASELINE.json configs[0]: batch 4, 1 frame, 128 audio tokens           (survey_legs; ViT-B only)
      C2    configs[1]'s shape under the reference's one-frame semantics: batch 8, 1 frame, 512 audio tokens
    A reported baseline, never a target."""
    import dataclasses
    import platform
    cores, cores_note = usable_cores()
    torch.set_num_threads(cores)
    cpu_model = platform.processor() or "unknown"
    try:
        with open("/proc/cpuinfo") as f:
            for ln in f:
                if ln.startswith("model name"):
                    cpu_model = ln.split(":", 1)[1].strip()
                    break
    except Exception:
        pass
    main = _cpu_leg(cfg, batch, warmup, steps, cores, "same config as the GPU line, small batch")
    B, dt = batch, sum(main["step_seconds"]) / len(main["step_seconds"])
    out = {"value": main["value"], "unit": "samples/s", "cores": cores, "cores_note": cores_note, "kind": "port", "cpu_model": cpu_model,
           "torch": torch.__version__, "batch": B, "warmup_steps": warmup, "timed_steps": steps, "step_seconds": main["step_seconds"],
           "sample": f"{warmup} warm-up + {steps} timed full steps (contrastive + MAE fwd/bwd, 2x Adam) at batch {B} of the same config "
                     f"({cfg.frames} frames x{cfg.video_tokens} + {cfg.audio_tokens} audio tokens), fp32 torch CPU, {cores} threads, mean {dt:.2f} s/step",
           "gflops": main["gflops"]}
    if survey_legs and cfg.embed_dim == 768 and cfg.st == cfg.patch:
        legs = []
        for label, kw, b, w, n in (("C1: BASELINE configs[0] (batch 4, 1 frame x196 + 128 audio tokens)", {"frames": 1, "audio_tokens": 128}, 4, 1, 3),
                                   ("C2-shaped (batch 8, 1 frame x196 + 512 audio tokens: reference pre-training semantics)", {"frames": 1, "audio_tokens": 512}, 8, 1, 3)):
            try:
                legs.append(_cpu_leg(dataclasses.replace(cfg, **kw), b, w, n, cores, label))
            except Exception as e:                                   # a report, never a gate
                legs.append({"config": label, "value": None, "error": repr(e)})
        out["survey_configs"] = legs
    return out


def usable_cores():
    """Cores this process may actually use: the affinity mask, cut down to the cgroup CPU quota when there is one (a one-GPU box
    exposes every hardware thread of the host in the mask but grants a 16-CPU share: 256 threads against that quota thrash)."""
    try:
        n = len(os.sched_getaffinity(0))
    except Exception:
        n = os.cpu_count() or 1
    note = f"affinity mask {n}"
    quota = None
    try:
        with open("/sys/fs/cgroup/cpu.max") as f:                       # cgroup v2: "<quota> <period>" or "max <period>"
            q, per = f.read().split()
            if q != "max":
                quota = float(q) / float(per)
    except Exception:
        try:
            with open("/sys/fs/cgroup/cpu/cpu.cfs_quota_us") as f, open("/sys/fs/cgroup/cpu/cpu.cfs_period_us") as g:
                q, per = float(f.read()), float(g.read())
                if q > 0:
                    quota = q / per
        except Exception:
            pass
    if quota is not None and quota < n:
        n = max(1, int(quota))
        note += f", cgroup quota {quota:.1f} CPUs"
    env = os.environ.get("AVSIAM_CPU_THREADS")
    if env:
        n = max(1, min(n, int(env)))
        note += f", AVSIAM_CPU_THREADS={env}"
    elif quota is None and n > 32:
        # no quota visible: stay within the documented CPU share of a one-GPU box rather than oversubscribe a shared host
        n = 16
        note += ", no cgroup quota visible: limited to the 16-CPU share of a one-GPU box"
    return n, note


def _sha1(path):
    import hashlib
    with open(path, "rb") as f:
        return hashlib.sha1(f.read()).hexdigest()


def _pmc_file(args, name, kernel_file):
    """A stored PMC summary (profiles/rNN/<name>, newest round first) if it was measured on THIS workload and on the kernel source as
    it is now (SHA-1 stored beside the numbers); else None."""
    if args.batch != 64 or args.frames != 10 or args.audio_tokens != 512 or args.model != "vit_base" or args.fp8 or args.recompute:
        return None
    for rnd in ("r06", "r05", "r04", "r03", "r02"):
        path = os.path.join(ROOT, "profiles", rnd, name)
        if not os.path.exists(path):
            continue
        try:
            with open(path) as f:
                d = json.load(f)
            if d["source_sha1"].get(kernel_file) == _sha1(os.path.join(ROOT, "avsiam_amd", "csrc", kernel_file)):
                d["_file"] = os.path.relpath(path, ROOT)
                return d
        except Exception:
            pass
    return None


def pmc_traffic(args, kernel_file="gemm.hip", key="gemm_nt"):
    """HBM bytes per launch of a kernel family, measured OFFLINE with rocprofv3 PMC passes of this same command (FETCH_SIZE and
    WRITE_SIZE in separate runs, FETCH doubled per MI355X_MICROARCH.md) and stored in profiles/rNN/traffic.json together with the
    SHA-1 of the kernel source it was measured on.  null when the stored figure is for another workload or another kernel source."""
    d = _pmc_file(args, "traffic.json", kernel_file)
    try:
        return float(d["kernels"][key]["hbm_bytes_per_launch"]) if d else None
    except Exception:
        return None


def pmc_source(args, kernel_file="gemm.hip"):
    """Where the `traffic` / `pmc` figures of the line come from: they are NOT measured in this run (counters need rocprofv3 passes
    of their own) but read from committed summaries of such passes of this same command, attached only while the kernel source is
    byte-identical to the one they were measured on."""
    out = {}
    for name in ("traffic.json", "pmc_busy.json"):
        d = _pmc_file(args, name, kernel_file)
        if d:
            out[name] = {"file": d["_file"], "kernel_source": kernel_file, "kernel_source_sha1": d["source_sha1"].get(kernel_file),
                         "commit": d.get("commit"), "box": d.get("box"), "measured": d.get("date")}
    return out or None


def pmc_busy(args, kernel_file="gemm.hip", key="gemm_nt"):
    """Matrix-pipe busy fraction of a kernel family (SQ_VALU_MFMA_BUSY_CYCLES over all SIMD cycles; tools/pmc_busy.sh ->
    profiles/rNN/pmc_busy.json), gated on the kernel source's SHA-1 like the traffic figure."""
    d = _pmc_file(args, "pmc_busy.json", kernel_file)
    try:
        k = d["kernels"][key]
        return {"mfma_busy": k["mfma_busy"], "valu_active_of_wave_time": k["valu_active_of_wave_time"], "wait_any_of_wave_time": k["wait_any_of_wave_time"]} if d else None
    except Exception:
        return None


def launch_ranks(n, argv, dry_run=False):
    """`python bench.py --gpus N` started WITHOUT torch.distributed.run: this process - which has made no GPU call (importing torch
    initialises nothing) and makes none - starts `python -m torch.distributed.run --nproc-per-node N ... bench.py <same arguments>` as a
    CHILD process (never an exec: a process that touched the GPU must not be replaced, and this one stays the parent anyway), relays the
    child's stdout - rank 0's one JSON line - and returns the child's exit code (non-zero if any rank failed: torch.distributed.run
    propagates it).  The reference gets the same shape of launch from its shell script (run_pretrain_base.sh:75 torchrun -> env
    RANK / WORLD_SIZE / LOCAL_RANK -> utils.py:283-299).
    The child runs in a process group of its own and never outlives this process: SIGTERM / SIGINT / SIGHUP received here are
    forwarded to the whole group, and whatever ends the relay (an exception, the driver's timeout killing the parent's wait) terminates the
    group - SIGTERM, ten seconds of grace, SIGKILL - so that no orphaned rank keeps a GPU for the next `--gpus N` run to collide with.  The
    rendezvous port is taken by bind-and-close; should another process grab it before torchrun binds (the child then dies within seconds
    without a line), the launch is repeated on a fresh port, twice at most."""
    import signal
    import socket
    import subprocess
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")           # dmabuf IPC: without it RCCL's cross-process buffer sharing fails on this driver
    env.setdefault("OMP_NUM_THREADS", str(max(1, (os.cpu_count() or n) // n)))
    argv = [a for a in argv if a != "--dry-run-launch"]

    def command():
        with socket.socket() as s:                # a free rendezvous port on the loopback interface
            s.bind(("127.0.0.1", 0))
            port = s.getsockname()[1]
        return port, [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={n}", "--master-addr", "127.0.0.1",
                      "--master-port", str(port), os.path.abspath(__file__)] + argv

    if dry_run:
        port, cmd = command()
        print(json.dumps({"launch": cmd, "ranks": n, "env": {k: env[k] for k in ("HSA_ENABLE_IPC_MODE_LEGACY", "OMP_NUM_THREADS")},
                          "parent_touched_gpu": bool(torch.cuda.is_initialized())}))
        return 0

    def stop(child, sig=signal.SIGTERM, grace=10.0):
        """end the child's whole process group (torchrun + every rank)"""
        if child.poll() is not None:
            return
        try:
            os.killpg(child.pid, sig)
        except (ProcessLookupError, PermissionError):
            pass
        try:
            child.wait(timeout=grace)
        except subprocess.TimeoutExpired:
            try:
                os.killpg(child.pid, signal.SIGKILL)
            except (ProcessLookupError, PermissionError):
                pass
            child.wait()

    rc = 1
    for attempt in range(3):
        port, cmd = command()
        print(f"[bench] --gpus {n} without a launcher: starting {n} ranks as a child torch.distributed.run (port {port})", file=sys.stderr, flush=True)
        t0 = time.time()
        child = subprocess.Popen(cmd, env=env, stdout=subprocess.PIPE, text=True, start_new_session=True)
        old = {}

        def forward(signum, frame, child=child):
            stop(child, signal.SIGTERM)
            raise SystemExit(128 + signum)

        for sg in (signal.SIGTERM, signal.SIGINT, signal.SIGHUP):
            try:
                old[sg] = signal.signal(sg, forward)
            except ValueError:                    # not the main thread (a test harness): the finally below still cleans up
                pass
        lines = []
        try:
            for ln in child.stdout:
                lines.append(ln)
                sys.stdout.write(ln)
                sys.stdout.flush()
            rc = child.wait()
        finally:
            stop(child)
            for sg, h in old.items():
                signal.signal(sg, h)
        if rc != 0 and not lines and time.time() - t0 < 30.0 and attempt < 2:
            print(f"[bench] the launcher exited with {rc} after {time.time() - t0:.0f} s without a line (rendezvous port {port} taken?): once more on a fresh port",
                  file=sys.stderr, flush=True)
            continue
        break
    if rc == 0 and not any(ln.lstrip().startswith("{") for ln in lines):
        print("[bench] the ranks exited cleanly but rank 0 printed no JSON line", file=sys.stderr)
        return 1
    return rc


def secondary_shapes(args, dev):
    """The reference's OWN pre-training shapes on the GPU, after the timed region: the reference trains on ONE frame per
    clip (/root/reference/src/dataloader.py:471,519) at per-GPU batch 4 (egs/audioset/run_pretrain_base.sh:30-31); the headline workload
    (10 frames, batch 64) is BASELINE.json's, not the reference's.  Same step (both passes, both Adam updates, device-drawn plans), same
    kernels; `graph` = the step replayed from a captured hipGraph (avsiam_amd.graph_step.GraphedTrainStep) where the eager step is bound
    by the host's launch rate.  The CPU legs of `cpu_baseline.survey_configs` time these same shapes on the host."""
    import gc
    from avsiam_amd.config import AVSiamConfig
    from avsiam_amd.flops import gflop_per_sample
    from avsiam_amd.models import CAVMAE_BASE
    from avsiam_amd.traintest_cavmae_base import train_step
    from avsiam_amd.weights import synth_inputs
    out = []
    from avsiam_amd.config import vit_huge14
    shapes = [("C2 under the reference's one-frame semantics: batch 64, 1 frame x196 + 512 audio tokens", AVSiamConfig(frames=1, audio_tokens=512), 64, {}),
              ("reference launch geometry: batch 4 per GPU, 1 frame x196 + 512 audio tokens (run_pretrain_base.sh:30-31)", AVSiamConfig(frames=1, audio_tokens=512), 4, {}),
              ("C1: BASELINE configs[0], batch 4, 1 frame x196 + 128 audio tokens", AVSiamConfig(frames=1, audio_tokens=128), 4, {})]
    if args.secondary_huge:
        # BASELINE.json configs[4] on ONE GPU, driver-timed: models.CAVMAE_HUGE, fp8 mode 3 (e4m3 forward, e5m2 x e4m3 input and weight
        # gradients), both passes from one activation pool, nothing recomputed; 2 warm-up steps (the first calibrates the scales) + 3 timed
        shapes.append(("configs[4] on one GPU: ViT-H/14 (models.CAVMAE_HUGE), fp8 mode 3, batch 64, 10 frames x256 + 657 audio tokens, one activation pool",
                       vit_huge14(frames=10), 64, {"fp8_mode": "3", "share_pass_buffers": True, "steps": 3, "warm": 2, "cls": "CAVMAE_HUGE"}))
    for label, cfg, B, kw in shapes:
        try:
            import avsiam_amd.models as _models
            nsteps, nwarm = kw.get("steps", args.secondary_steps), kw.get("warm", 3)
            m = getattr(_models, kw.get("cls", "CAVMAE_BASE"))(cfg=cfg, verbose=False, plan_seed=87, fp8_mode=kw.get("fp8_mode"),
                                                               share_pass_buffers=kw.get("share_pass_buffers")).to(dev)
            m.publish_grads = False
            a, v = synth_inputs(cfg, B, 87)
            a, v = a.to(dev), v.to(dev)
            gf = gflop_per_sample(cfg, B)

            def timed(step_fn, n, warm=3):
                for _ in range(warm):
                    step_fn()
                torch.cuda.synchronize()
                t0 = time.perf_counter()
                for _ in range(n):
                    step_fn()
                torch.cuda.synchronize()
                return (time.perf_counter() - t0) / n

            dt = timed(lambda: train_step(m, a, v, args.lr), nsteps, nwarm)
            rec = {"workload": label, "batch": B, "frames": cfg.frames, "audio_tokens": cfg.audio_tokens, "value": B / dt, "unit": "samples/s",
                   "ms_per_step": 1e3 * dt, "gflop_per_sample": gf, "mfu": B / dt * gf / 1e3 / PEAK_BF16_TFLOPS, "steps": nsteps, "warmup": nwarm,
                   "dtype": "bf16" if not kw.get("fp8_mode") else "fp8 mode " + kw["fp8_mode"]}
            if kw.get("fp8_mode"):
                rec["mfu_note"] = "model FLOP rate against the dense BF16 peak (2.5 PFLOP/s), as every other `mfu` of this line; the fp8 GEMM launches are priced against 5 PFLOP/s by `bench.py --model vit_huge14 --fp8 --fp8-wgrad` (roofline_fp8)"
                rec["fp8_saturation_events"] = float(m.fp8_saturation_events())
                rec["peak_memory_gib"] = round(torch.cuda.max_memory_allocated() / 2 ** 30, 1)
            if B <= 8:
                try:
                    from avsiam_amd.graph_step import GraphedTrainStep
                    gs = GraphedTrainStep(m, a, v, args.lr)
                    dtg = timed(gs.step, args.secondary_steps)
                    rec["graph"] = {"value": B / dtg, "ms_per_step": 1e3 * dtg, "mfu": B / dtg * gf / 1e3 / PEAK_BF16_TFLOPS,
                                    "kernel_nodes": gs.kernel_nodes, "note": "the same step replayed from one captured hipGraph (plans still drawn per step: "
                                    "seeds and Adam step counts live in device memory)"}
                except Exception as e:                                  # a report, never a gate
                    rec["graph"] = {"value": None, "error": repr(e)}
            # the record's own value is the faster of the two ways the product can run this shape (the eager step is bound by the HOST's launch
            # rate at batch 4 - 792 launches in 15 - 20 ms, box-dependent - the replayed hipGraph is not); both stay in the record
            rec["mode"] = "eager"
            if rec.get("graph", {}).get("value") and rec["graph"]["value"] > rec["value"]:
                rec["eager"] = {k: rec[k] for k in ("value", "ms_per_step", "mfu")}
                rec.update({k: rec["graph"][k] for k in ("value", "ms_per_step", "mfu")})
                rec["mode"] = "graph"
            out.append(rec)
            log(f"secondary [{label}]: {rec['value']:.1f} samples/s, {rec['ms_per_step']:.2f} ms/step ({rec['mode']})" +
                (f"; graph {rec['graph']['ms_per_step']:.2f} ms/step" if rec.get("graph", {}).get("value") else ""))
        except Exception as e:
            out.append({"workload": label, "value": None, "error": repr(e)})
        finally:
            m = a = v = None
            gc.collect()
            torch.cuda.empty_cache()
    return out


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
    ap.add_argument("--model", choices=("vit_base", "vit_large", "vit_huge", "vit_huge14"), default="vit_base",
                    help="vit_base = BASELINE.json's metric (configs[1]); vit_large = configs[3]'s shape, vit_huge = configs[4]'s encoder width in bf16 "
                         "on 16x16 patches (use --batch 32: saved activations of batch 64 x 10 frames exceed 288 GB), vit_huge14 = the same on the "
                         "14x14 patch grid (256 tokens per frame, 657 audio tokens) - extra data points")
    ap.add_argument("--lr", type=float, default=2e-4)
    ap.add_argument("--fp8", action="store_true", help="fp8 (e4m3) forward GEMMs (EngineOptions.fp8): configs[4]'s fp8 MFMA path as an extra data point; "
                    "the line then says dtype fp8-forward/bf16-backward and is NOT the headline metric")
    ap.add_argument("--fp8-dgrad", action="store_true", help="with --fp8: the four input-gradient GEMMs of a block on e5m2 gradient operands too (fp8 mode 2)")
    ap.add_argument("--fp8-wgrad", action="store_true", help="with --fp8: input gradients AND the four weight gradients of a block on fp8 operands (fp8 mode 3)")
    ap.add_argument("--share-pass-buffers", action="store_true", help="the two passes of the step take their activation buffers from one pool "
                    "(CAVMAE_BASE(share_pass_buffers=True)): the card holds the larger pass instead of the sum - shapes that otherwise need --recompute")
    ap.add_argument("--recompute", nargs="?", const="1", default=None, metavar="FRACTION",
                    help="per-layer activation recompute (EngineOptions.recompute): for shapes whose saved activations do not fit the GPU, e.g. "
                         "--model vit_huge14 at batch 64; with a FRACTION (0.375) only that share of every stack's blocks is recomputed and "
                         "the rest of the 288 GB holds saved activations; `auto` takes the smallest share that leaves a tenth of the card "
                         "free after one step; never for the headline metric")
    ap.add_argument("--force-dp", action="store_true", help="form the RCCL process group and issue EVERY collective of the data-parallel step "
                    "(packed embedding all-gather, chunked overlapped gradient all-reduce) even at world size 1 - the most of the multi-GPU "
                    "path a one-GPU box can execute; AVSIAM_COMM=rccl selects the C ABI's communicator, AVSIAM_DP_WIRE=bf16 the bf16 wire")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--roofline-steps", type=int, default=2, help="steps of the separate single-stream pass that times the other kernel families")
    ap.add_argument("--no-kernel-events", action="store_true")
    ap.add_argument("--secondary-steps", type=int, default=10, help="timed steps of each reference-native shape measured after the timed region "
                    "(`secondary` in the line: one frame at batch 64, and BASELINE configs[0] = batch 4 / one frame / 128 audio tokens); 0 = skip")
    ap.add_argument("--no-secondary-huge", dest="secondary_huge", action="store_false",
                    help="skip the ViT-H/14 fp8 mode-3 line of `secondary` (BASELINE configs[4] on one GPU: ~190 GiB, ~1 minute)")
    ap.add_argument("--all-kernel-events", action="store_true",
                    help="HIP-event timing of every kernel family (default: the dominant kernel, gemm_nt, only - each timed "
                         "launch costs the stream ~5 us)")
    ap.add_argument("--dry-run-launch", action="store_true", help="with --gpus N > 1 outside torch.distributed.run: print the launch command and "
                    "the environment the ranks would get as one JSON line, start nothing (tests/test_bench_launcher_cpu.py)")
    args = ap.parse_args()
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        raise SystemExit(launch_ranks(args.gpus, sys.argv[1:], dry_run=args.dry_run_launch))
    # stdout carries exactly ONE line, the JSON: whatever libraries print on the way (RCCL's version banner when a process group
    # forms, ...) goes to stderr
    sys.stdout.flush()
    json_fd = os.dup(1)
    os.dup2(2, 1)

    from avsiam_amd import _lib, ops
    from avsiam_amd.config import AVSiamConfig
    from avsiam_amd.flops import gflop_per_sample
    from avsiam_amd.models import CAVMAE_BASE
    from avsiam_amd.traintest_cavmae_base import train_step

    world = int(os.environ.get("WORLD_SIZE", 1))
    rank = int(os.environ.get("RANK", 0))
    local = int(os.environ.get("LOCAL_RANK", 0))
    if world != args.gpus and not (args.force_dp and args.gpus == 1 and world == 1):
        raise SystemExit(f"bench: --gpus {args.gpus} but the launcher's WORLD_SIZE is {world}")
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU: the hot path runs only on the HIP kernels")
    _lib.load()
    # AVSIAM_BENCH_SHARE_GPU=1: REHEARSAL of the multi-rank path on a one-GPU box - every rank uses device 0, the group is gloo and the
    # collectives are staged through the host (comm.HostStagedComm; RCCL refuses two ranks on one device).  The line says so
    # (`config.rehearsal`) and its value is not a throughput of anything: the point is that `python bench.py --gpus N` - launcher,
    # rank environment, world-size branches, max-over-ranks timing - runs end to end (tests/test_boundary_gpu.py).
    share = os.environ.get("AVSIAM_BENCH_SHARE_GPU", "0") == "1"
    if share:
        local = 0
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)
    import torch.distributed as dist
    if world > 1 or args.force_dp:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29533")
        if share:
            dist.init_process_group("gloo", rank=rank, world_size=world)
        else:
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)
    cdev = torch.device("cpu") if share else dev          # where the few scalar collectives of this script live

    if args.model == "vit_large":
        from avsiam_amd.config import vit_large
        cfg = vit_large(audio_tokens=args.audio_tokens, frames=args.frames)
    elif args.model == "vit_huge":
        from avsiam_amd.config import vit_huge
        cfg = vit_huge(audio_tokens=args.audio_tokens, frames=args.frames)
    elif args.model == "vit_huge14":
        from avsiam_amd.config import vit_huge14
        cfg = vit_huge14(frames=args.frames)
    else:
        cfg = AVSiamConfig(audio_tokens=args.audio_tokens, frames=args.frames)
    mname = {"vit_base": "ViT-B/16", "vit_large": "ViT-L/16", "vit_huge": "ViT-H/16 (1280 wide, 32 layers, 16 heads of 80)",
             "vit_huge14": "ViT-H/14 (1280 wide, 32 layers, 16 heads of 80, 14x14 patches)"}[args.model]
    torch.manual_seed(87 + rank)
    log(f"building model (frames={args.frames}, batch={args.batch}/GPU, world={world})")
    # precision / recompute are options of the MODEL (config.EngineOptions; the AVSIAM_* environment seeds what the flags leave open)
    from avsiam_amd.config import EngineOptions
    fp8_mode = ("3" if args.fp8_wgrad else "2" if args.fp8_dgrad else "1") if args.fp8 else None
    recompute = [args.recompute if (args.recompute and args.recompute != "auto") else None]
    EngineOptions.from_env(fp8=fp8_mode, recompute=recompute[0])          # validates the values before anything is built
    comm = None
    if share and world > 1:
        from avsiam_amd.comm import HostStagedComm
        comm = HostStagedComm()
    if args.force_dp and world == 1:
        from avsiam_amd.comm import RcclComm, TorchDistComm
        comm = RcclComm(always=True) if os.environ.get("AVSIAM_COMM", "torch") == "rccl" else TorchDistComm(always=True)
    from avsiam_amd.weights import synth_inputs
    a, v = synth_inputs(cfg, args.batch, 87 + rank)
    a, v = a.to(dev), v.to(dev)

    def build():
        m = CAVMAE_BASE(cfg=cfg, verbose=False, plan_seed=87 + rank, share_pass_buffers=args.share_pass_buffers, fp8_mode=fp8_mode,
                        recompute=recompute[0]).to(dev)
        m.set_distributed(world, rank, comm)
        m.publish_grads = False
        return m

    if args.recompute == "auto":
        # the smallest recomputed share of every stack's blocks with which one whole step fits the card and leaves a tenth of it free
        # (every rank must arrive at the same answer: the verdict of a candidate is the AND over ranks)
        import gc
        model = None
        for frac in ("0", "0.125", "0.25", "0.375", "0.5", "0.625", "0.75", "1"):
            recompute[0] = frac
            try:
                model = build()
                train_step(model, a, v, args.lr)
                torch.cuda.synchronize()
                free, total = torch.cuda.mem_get_info()
                ok = free >= 0.10 * total
                log(f"--recompute auto: fraction {frac}: {(total - free) / 2**30:.1f} of {total / 2**30:.1f} GiB in use -> {'taken' if ok else 'too close'}")
            except torch.OutOfMemoryError:
                ok = False
                log(f"--recompute auto: fraction {frac}: out of memory")
            if world > 1 or args.force_dp:
                flag = torch.tensor([1 if ok else 0], device=cdev)
                dist.all_reduce(flag, op=dist.ReduceOp.MIN)
                ok = bool(flag.item())
            if ok:
                args.recompute = frac
                break
            model = None
            gc.collect()
            torch.cuda.empty_cache()
        if model is None:
            raise SystemExit("bench: --recompute auto: the shape does not fit even with every block recomputed")
        if args.recompute == "0":
            args.recompute = None
        torch.cuda.reset_peak_memory_stats()          # the rejected candidates' peaks are not this run's
    else:
        model = build()

    def sync():
        torch.cuda.synchronize()
        if world > 1 or args.force_dp:
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
    model.flush_deferred()                       # AVSIAM_DP_DEFER: the last step's MAE-only update belongs to the timed work
    sync()
    dt = time.perf_counter() - t0
    log(f"timed region: {args.steps} steps in {dt:.3f} s")
    prof, ops.prof = ops.prof, None
    # The other kernels that matter: weight-gradient GEMM, decoder attention (hd 32), LayerNorm backward.  In the timed
    # region the weight gradients run on a second stream beside attention / LayerNorm backward, so a launch-to-end time there
    # includes waiting for CUs.  Their rooflines are therefore taken in a short SEPARATE pass after the timed region, with
    # everything on one stream (EngineOptions.wgrad_stream "0") and HIP events around every launch of those families.
    prof2 = None
    if not args.no_kernel_events and world == 1 and args.roofline_steps > 0:
        mode = model.options.wgrad_stream
        model.set_options(wgrad_stream="0")
        train_step(model, a, v, args.lr)
        torch.cuda.synchronize()
        ops.prof = ops.KernelProfiler()                  # every launch of every kernel family (outside the timed region: the event cost is free here)
        for _ in range(args.roofline_steps):
            train_step(model, a, v, args.lr)
        torch.cuda.synchronize()
        prof2, ops.prof = ops.prof, None
        model.set_options(wgrad_stream=mode)
    # per-rank times of the timed region (a straggler must be visible in the line): max = the metric's clock, min beside it
    tmax = torch.tensor([dt], device=cdev, dtype=torch.float64)
    tmin = tmax.clone()
    rank_ms = None
    if world > 1 or args.force_dp:
        if dist.get_world_size() != args.gpus:
            raise SystemExit(f"bench: the process group has {dist.get_world_size()} ranks, --gpus says {args.gpus}")
        every = [torch.zeros_like(tmax) for _ in range(dist.get_world_size())]
        dist.all_gather(every, tmax)
        rank_ms = [round(1e3 * float(t.item()) / args.steps, 3) for t in every]
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        dist.all_reduce(tmin, op=dist.ReduceOp.MIN)
    dt_min = float(tmin.item())
    dt = float(tmax.item())
    # the headline workload's peak, read BEFORE the secondary shapes build their own models beside this one (the figure used to
    # include them - 106 GiB reported for a 79.8 GiB step)
    peak_gib = round(torch.cuda.max_memory_allocated() / 2 ** 30, 1)
    losses = [float(x.item()) for x in last]
    # everything the line says about the headline model is read NOW: the model and its activations are dropped before the secondary shapes
    # (the ViT-H/14 fp8 line alone needs ~190 GiB of the card)
    minfo = {"options": model.options.describe(), "grad_stream": model.options.grad_stream, "fp8_lean": model.options.fp8_lean, "dp_wire": model.dp_wire,
             "defer_p2": bool(model.defer_p2), "last_reduce_messages": model.last_reduce_messages,
             "pool": ({"share_pass_buffers": True, "activation_pool_gib": round(model._pool.nbytes() / 2 ** 30, 2),
                       "activation_pool_live_gib_last_pass": round(model._pool.used() / 2 ** 30, 2)} if args.share_pass_buffers and model._pool is not None else {})}
    secondary = None
    if world == 1 and not args.force_dp and args.secondary_steps > 0 and args.model == "vit_base" and not (args.fp8 or args.recompute):
        import gc
        model.release_buffers()
        model = last = None
        gc.collect()
        torch.cuda.empty_cache()
        torch.cuda.reset_peak_memory_stats()
        secondary = secondary_shapes(args, dev)
    if rank == 0:
        sps = world * args.batch * args.steps / dt
        gf = gflop_per_sample(cfg, args.batch)
        line = {
            "metric": f"AV pretrain samples/sec ({mname}, 75% mask)", "value": sps, "unit": "samples/s", "n_gpus": world,
            "steps": args.steps, "warmup": args.warmup, "ms_per_step": 1e3 * dt / args.steps, "higher_is_better": True,
            "scaling": "weak", "vs_baseline": None, "dtype": ("fp8(e4m3)-forward/fp8(e5m2 x e4m3)-input-and-weight-gradients" if args.fp8_wgrad else "fp8(e4m3)-forward/fp8(e5m2 x e4m3)-input-gradients/bf16-weight-gradients" if args.fp8_dgrad else "fp8(e4m3)-forward/bf16-backward") if args.fp8 else "bf16", "data": "synthetic",
            "config": {"workload": f"AVSiam pretrain step (contrastive + MAE passes, 2x Adam), {mname}, {args.frames} frames x{cfg.video_tokens} + "
                                   f"{cfg.audio_tokens} audio tokens, 75% mask, batch {args.batch}/GPU",
                       "global_batch": world * args.batch, "frames": args.frames, "audio_tokens": cfg.audio_tokens,
                       "parallelism": f"dp{world}", "gflop_per_sample": gf,
                       "residual_gradient_stream": minfo["grad_stream"],   # bf16 (default) | fp32
                       "options": minfo["options"],
                       # what the collective library itself reports: the size of the process group the step's collectives ran in
                       **({"collectives": {"backend": dist.get_backend(), "library": "gloo, staged through the host (rehearsal)" if share else "RCCL (torch.distributed 'nccl' on ROCm)", "group_world_size": dist.get_world_size(),
                                           "comm": "host-staged" if share else os.environ.get("AVSIAM_COMM", "torch"), "allreduce_messages_last_backward": minfo["last_reduce_messages"],
                                           # what the first scaling line has to say about itself: CUs every persistent kernel leaves to the
                                           # collectives, whether the gradient all-reduce overlaps the backward, and what travels on xGMI
                                           "cu_reserve": _lib.tuning_get("cu_reserve"), "persistent_cu_slots": _lib.load().avs_persistent_cu_slots(),
                                           "gradient_allreduce": ("chunked, overlapped with the backward (AVSIAM_DP_OVERLAP=1)" if os.environ.get("AVSIAM_DP_OVERLAP", "1") != "0"
                                                                  else "one blocking message after the backward (AVSIAM_DP_OVERLAP=0)"),
                                           "wire_format": minfo["dp_wire"], "defer_mae_only_update": minfo["defer_p2"],
                                           "embedding_allgather": "one packed [2, B, D] fp32 message per rank and step"}}
                          if (world > 1 or args.force_dp) else {}), **({"activation_recompute": True if args.recompute == "1" else float(args.recompute)} if args.recompute else {}),
                       **({"fp8_8bit_only_outputs": minfo["fp8_lean"]} if args.fp8_wgrad else {}),
                       "peak_memory_gib": peak_gib,
                       **minfo["pool"],
                       **({"rehearsal": "AVSIAM_BENCH_SHARE_GPU=1: all ranks on ONE GPU, gloo + host-staged collectives - not a throughput figure"} if share else {}),
                       **({"force_dp": {"comm": os.environ.get("AVSIAM_COMM", "torch"), "wire": os.environ.get("AVSIAM_DP_WIRE", "fp32"),
                                        "overlap": os.environ.get("AVSIAM_DP_OVERLAP", "1"), "defer_mae_only": os.environ.get("AVSIAM_DP_DEFER", "0"),
                                        "allreduce_messages_last_backward": minfo["last_reduce_messages"]}}
                          if args.force_dp else {})},
            "model_tflops": sps * gf / 1e3, "mfu_vs_dense_bf16_peak": sps * gf / 1e3 / (world * PEAK_BF16_TFLOPS),
            "final_losses": {"loss_mae": losses[0], "loss_mae_a": losses[1], "loss_mae_v": losses[2], "loss_c": losses[3], "c_acc": losses[4]},
        }
        if rank_ms is not None:
            line["ranks"] = {"world_size_of_group": dist.get_world_size(), "ms_per_step_by_rank": rank_ms,
                             "ms_per_step_max": round(1e3 * dt / args.steps, 3), "ms_per_step_min": round(1e3 * dt_min / args.steps, 3)}
        if secondary is not None:
            line["secondary"] = secondary
        if prof is not None:
            s = prof.summary()
            mm = [k for k in s if k.startswith("gemm_nt") and k != "gemm_nt_fp8"]      # the bf16 launches (all of them without --fp8)
            flops = sum(s[k]["work"] for k in mm)
            ms = sum(s[k]["total_ms"] for k in mm)
            n = sum(s[k]["launches"] for k in mm)
            nd = sum(s[k]["dispatches"] for k in mm)
            ach = flops / (ms * 1e-3) / 1e12
            line["roofline"] = {"bound": "mfma", "kernel": "gemm_nt8_kernel / gemm_nt_kernel (forward + dgrad bf16 MFMA GEMMs, one avs_gemm_nt_bf16 call = one launch)", "achieved": ach,
                                "peak": PEAK_BF16_TFLOPS, "unit": "TFLOP/s", "frac": ach / PEAK_BF16_TFLOPS, "traffic": pmc_traffic(args),
                                "pmc": pmc_busy(args), "pmc_source": pmc_source(args), "launches": n, "avg_launch_us": 1e3 * ms / n, "flops_per_launch": flops / n,
                                "dispatches": nd, "avg_dispatch_us": 1e3 * ms / nd,
                                "sampling": "all launches" if args.all_kernel_events else "every 5th launch of the timed region"}
            if "gemm_nt_fp8" in s:              # --fp8: the e4m3 forward GEMMs are a kernel family of their own, against the fp8 peak
                x = s["gemm_nt_fp8"]
                a8 = x["rate"] / 1e12
                line["roofline_fp8"] = {"bound": "mfma", "kernel": "gemm_nt8_kernel<., FP8> (forward GEMMs on e4m3 operands" + (" and input-gradient GEMMs on e5m2 x e4m3" if (args.fp8_dgrad or args.fp8_wgrad) else "") + ", v_mfma_f32_16x16x128_f8f6f4)",
                                        "achieved": a8, "peak": PEAK_FP8_TFLOPS, "unit": "TFLOP/s", "frac": a8 / PEAK_FP8_TFLOPS, "traffic": None,
                                        "launches": x["launches"], "avg_launch_us": x["avg_us"], "flops_per_launch": x["work"] / x["launches"]}
                line["roofline"]["kernel"] += (" - the bf16 launches of the fp8 mode (patch embedding, decoder embedding, prediction heads: no fp8 instantiation)" if (args.fp8_dgrad or args.fp8_wgrad) else " - the bf16 launches of the fp8 mode (backward; forward GEMMs without an fp8 instantiation)")
            line["kernels"] = {k: {"launches": x["launches"], "total_ms": round(x["total_ms"], 3), "avg_us": round(x["avg_us"], 2),
                                   "rate_T_per_s": round(x["rate"] / 1e12, 3)} for k, x in sorted(s.items())}
        if prof2 is not None:
            s2 = prof2.summary()

            def fam(keys, bound, peak, unit, kernel, note=None, traffic=None, use_bytes=False, pmc=None):
                ks = [k for k in s2 if k in keys]
                if not ks:
                    return None
                work = sum(s2[k]["bytes" if use_bytes else "work"] for k in ks)
                ms = sum(s2[k]["total_ms"] for k in ks)
                n = sum(s2[k]["launches"] for k in ks)
                ach = work / (ms * 1e-3) / (1e12 if unit == "TFLOP/s" else 1e9)
                d = {"kernel": kernel, "bound": bound, "achieved": ach, "peak": peak, "unit": unit, "frac": ach / peak, "traffic": traffic,
                     "launches": n, "avg_launch_us": 1e3 * ms / n, "work_per_launch": work / n, "ms_per_step": ms / args.roofline_steps}
                if pmc:
                    d["pmc"] = pmc
                if note:
                    d["note"] = note
                return d

            more = [
                fam(("gemm_tn",), "mfma", PEAK_BF16_TFLOPS, "TFLOP/s", "gemm_tn8_kernel / gemm_tn_kernel (weight-gradient bf16 MFMA GEMMs, 2*M*N1*N2 FLOP per launch)",
                    traffic=pmc_traffic(args, "gemm.hip", "gemm_tn"), pmc=pmc_busy(args, "gemm.hip", "gemm_tn")),
                fam(("gemm_tn_fp8",), "mfma", PEAK_FP8_TFLOPS, "TFLOP/s", "gemm_tn8f_kernel (--fp8-wgrad: a block's weight gradients on e5m2 gradient x e4m3 activation "
                    "operands, v_mfma_f32_32x32x64_f8f6f4 on ds_read_b64_tr_b8 fragments; against the fp8 peak)"),
                fam(("attn_fwd_hd32", "attn_bwd_hd32"), "mfma", PEAK_BF16_TFLOPS, "TFLOP/s",
                    "attn_fwd / attn_bwd_* <32> (decoder attention, hd 32; 4 / 8 * sum L^2 * D algorithmic FLOP per forward / backward launch - the "
                    "backward's recomputation of S is not counted)",
                    note="VALU-bound beside the matrix pipe: one v_exp_f32 (8 issue cycles per wave) per score against 128 FLOP of MFMA work at hd 32 "
                         "caps these kernels near 0.5 of the MFMA peak before any other VALU work", traffic=pmc_traffic(args, "attention.hip", "attn_hd32"),
                    pmc=pmc_busy(args, "attention.hip", "attn_hd32")),
                fam(("attn_fwd_hd64", "attn_bwd_hd64"), "mfma", PEAK_BF16_TFLOPS, "TFLOP/s", "attn_* <64> (encoder attention, hd 64)",
                    note="sequences of 39-196 tokens (618 in the two joint layers): the time follows the ROWS, not the FLOP - see the hbm entry of the same launches",
                    traffic=pmc_traffic(args, "attention.hip", "attn_hd64"), pmc=pmc_busy(args, "attention.hip", "attn_hd64")),
                fam(("attn_fwd_hd64", "attn_bwd_hd64"), "hbm", PEAK_HBM_GBS, "GB/s", "attn_* <64> against the HBM roof: q, k, v, o (and dO, dq, dk, dv) "
                    "once per kernel that needs them = rows*D*2*(4 forward | 12 backward: two kernels) algorithmic bytes per launch",
                    traffic=pmc_traffic(args, "attention.hip", "attn_hd64"), use_bytes=True),
                fam(("layernorm_bwd",), "hbm", PEAK_HBM_GBS, "GB/s", "ln_bwd_kernel (LayerNorm backward + residual-gradient add + bf16 copy + column sums; "
                    "algorithmic bytes per launch = rows*D*(dy 2 + x 4 + dres 2 + dx_bf16 2) with the bf16 residual-gradient stream, "
                    "(2+4+4+4+2) with AVSIAM_GRAD_STREAM=fp32)", traffic=pmc_traffic(args, "layernorm.hip", "ln_bwd")),
                fam(("layernorm_fwd",), "hbm", PEAK_HBM_GBS, "GB/s", "ln_fwd_kernel (rows*D*(4+2) algorithmic bytes per launch)",
                    traffic=pmc_traffic(args, "layernorm.hip", "ln_fwd")),
            ]
            # every kernel family's own time per step (single stream: a launch's duration is the kernel's), so that the step can be added up
            # from the line alone; what the sum leaves of the timed region's ms_per_step is torch's two gradient zero-fills, launch gaps
            # and - in the timed region only - what the second stream overlaps
            fams = {k: round(x["total_ms"] / args.roofline_steps, 3) for k, x in sorted(s2.items())}
            line["roofline_more"] = {"pass": f"{args.roofline_steps} extra steps after the timed region, single stream (AVSIAM_WGRAD_STREAM=0 schedule), HIP events on every "
                                             "launch of every kernel family; the headline `value` and `roofline` come from the timed region",
                                     "kernels": [m for m in more if m], "ms_per_step_by_family": fams,
                                     "kernel_sum_ms_per_step": round(sum(fams.values()), 3),
                                     "launches_per_step": sum(x["launches"] for x in s2.values()) // args.roofline_steps}
        if world == 1 and not args.no_cpu_baseline:
            log("timing the CPU baseline (oracle) on the host cores")
            try:
                line["cpu_baseline"] = cpu_baseline(cfg)
            except Exception as e:                                       # the baseline is a report, never a gate
                line["cpu_baseline"] = {"value": None, "error": repr(e)}
        sys.stdout.flush()
        os.write(json_fd, (json.dumps(line) + "\n").encode())
    if world > 1 or args.force_dp:
        dist.barrier()
        if comm is not None and hasattr(comm, "close"):
            comm.close()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
