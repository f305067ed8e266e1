"""Where does the run-to-run distance of a 3-step training run come from, and does AVSIAM_DP_DEFER add to it?

VERDICT r3 item 1a.  tests/test_dp_gpu.py::test_deferred_mae_only_update_leaves_the_same_weights (round 3) compared the weights after
3 steps of the deferred schedule against ONE pair of undeferred runs and failed once on the driver's box with 2.2x the pair's distance,
all of it OUTSIDE the segment the deferral touches.  This tool repeats the same schedule many times per mode and prints, per arena
segment [pass-1 only | shared | MAE only], the squared distance to a reference run (run 0 of the undeferred mode), the number of elements
that differ by more than lr / 2, and - for the contrastive range - the distance after every step (the growth shows how much of it is
amplification of the first step's sign flips rather than anything the schedule does).

    python tools/defer_study.py --runs 12 --procs 1 [--lr 1e-3] [--steps 3]      # one process, loop-back collectives (world size 1)
    python tools/defer_study.py --runs 8 --procs 2                                # two processes sharing the GPU (gloo through the host)
    AVSIAM_WGRAD_STREAM=0 python tools/defer_study.py ...                          # single-stream schedule

Writes one JSON line per run to stdout (rank 0) and a table at the end.
"""
import argparse
import json
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def one_run(cfg, world, rank, comm_factory, defer, lr, steps, batch, a, v):
    from avsiam_amd.models import CAVMAE_BASE
    from avsiam_amd.param_spec import P1, P2
    from avsiam_amd.traintest_cavmae_base import train_step
    m = CAVMAE_BASE(cfg=cfg, init_seed=3, init_mode="random", verbose=False, plan_seed=77 + rank).cuda()
    m.publish_grads = False
    m.defer_p2 = defer
    m.set_distributed(world, rank, comm_factory())
    b1, b2 = m.arena.range[P1]
    per_step = []
    for _ in range(steps):
        train_step(m, a, v, lr)
        per_step.append(m.arena.p[b1:b2].detach().clone())        # the contrastive range: never pending, no flush needed
    m.state_dict()                                                 # flushes a pending MAE-only update
    w = m.arena.p[:m.arena.live_end].detach().clone()
    seg = {"p1": (b1, m.arena.range[P2][0]), "shared": (m.arena.range[P2][0], b2), "p2": (b2, m.arena.live_end)}
    del m
    return w, per_step, seg


def dist2(x, y):
    return float(((x.double() - y.double()) ** 2).sum())


def worker(rank, world, port, args, q):
    import torch.distributed as dist
    from avsiam_amd.comm import LocalComm
    from avsiam_amd.config import AVSiamConfig
    from avsiam_amd.weights import synth_inputs
    if world > 1:
        os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
        dist.init_process_group("gloo", rank=rank, world_size=world)
        from tests.helpers import HostStagedComm
        factory = HostStagedComm
    else:
        class Loop(LocalComm):                  # world size 1, but the data-parallel branch of the model runs (reducer, deferral)
            active = True
        factory = Loop
    cfg = AVSiamConfig(audio_tokens=128)
    a, v = synth_inputs(cfg, args.batch, 50 + rank)
    a, v = a.cuda(), v.cuda()
    ref = None
    rows = []
    for mode in ("undeferred", "deferred"):
        for r in range(args.runs):
            w, per_step, seg = one_run(cfg, world, rank, factory, mode == "deferred", args.lr, args.steps, args.batch, a, v)
            if ref is None:
                ref = (w, per_step)
                wn = {k: float((w[lo:hi].double() ** 2).sum()) for k, (lo, hi) in seg.items()}
                continue
            row = {"mode": mode, "run": r, "rank": rank}
            for k, (lo, hi) in seg.items():
                row[f"d2_{k}"] = dist2(w[lo:hi], ref[0][lo:hi])
                row[f"flips_{k}"] = int(((w[lo:hi] - ref[0][lo:hi]).abs() > args.lr / 2).sum())
            row["rel_all"] = (sum(row[f"d2_{k}"] for k in seg) / sum(wn.values())) ** 0.5
            row["rel_p2"] = (row["d2_p2"] / wn["p2"]) ** 0.5
            row["d2_contrastive_by_step"] = [dist2(x, y) for x, y in zip(per_step, ref[1])]
            rows.append(row)
            if rank == 0:
                print(json.dumps(row), flush=True)
    if world > 1:
        dist.destroy_process_group()
    q.put((rank, rows))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--runs", type=int, default=12)
    ap.add_argument("--procs", type=int, default=1)
    ap.add_argument("--lr", type=float, default=1e-3)
    ap.add_argument("--steps", type=int, default=3)
    ap.add_argument("--batch", type=int, default=3)
    ap.add_argument("--port", type=int, default=29791)
    args = ap.parse_args()
    import torch.multiprocessing as mp
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=worker, args=(r, args.procs, args.port, args, q)) for r in range(args.procs)]
    for p in procs:
        p.start()
    res = dict(q.get(timeout=3000) for _ in range(args.procs))
    for p in procs:
        p.join(timeout=120)
    rows = res[0]
    print(f"# procs={args.procs} lr={args.lr} steps={args.steps} batch={args.batch} AVSIAM_WGRAD_STREAM={os.environ.get('AVSIAM_WGRAD_STREAM', '2')}"
          f"  (reference = run 0 of the undeferred mode; d2 = squared distance to it)")
    print(f"{'mode':<11}{'run':>4}{'rel_all':>10}{'rel_p2':>10}{'d2_p1':>11}{'d2_shared':>11}{'d2_p2':>11}{'flips_p1':>9}{'flips_sh':>9}{'flips_p2':>9}   d2(contrastive range) after step 1..n")
    for r in rows:
        print(f"{r['mode']:<11}{r['run']:>4}{r['rel_all']:>10.5f}{r['rel_p2']:>10.5f}{r['d2_p1']:>11.3e}{r['d2_shared']:>11.3e}{r['d2_p2']:>11.3e}"
              f"{r['flips_p1']:>9}{r['flips_shared']:>9}{r['flips_p2']:>9}   " + " ".join(f"{x:.3e}" for x in r["d2_contrastive_by_step"]))
    for mode in ("undeferred", "deferred"):
        xs = sorted(r["rel_all"] for r in rows if r["mode"] == mode)
        if xs:
            print(f"# {mode}: n={len(xs)} rel_all min {xs[0]:.5f} median {xs[len(xs) // 2]:.5f} max {xs[-1]:.5f} max/median {xs[-1] / xs[len(xs) // 2]:.2f}")


if __name__ == "__main__":
    main()
