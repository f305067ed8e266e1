"""One training step from identical weights, several times per mode: how far apart are the all-reduced gradients (and the Adam moments) of
two runs of the SAME schedule, and of a deferred vs an undeferred run?  (diagnostic for tests/test_dp_gpu.py part A)
    python tools/defer_onestep.py [--procs 2] [--lr 1e-6]"""
import argparse
import math
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def worker(rank, world, port, args, q):
    import torch.distributed as dist
    from avsiam_amd.comm import LocalComm
    from avsiam_amd.config import AVSiamConfig
    from avsiam_amd.models import CAVMAE_BASE
    from avsiam_amd.param_spec import P1, P2
    from avsiam_amd.traintest_cavmae_base import train_step
    from avsiam_amd.weights import synth_inputs
    if world > 1:
        os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
        dist.init_process_group("gloo", rank=rank, world_size=world)
        from tests.helpers import HostStagedComm
        factory = HostStagedComm
    else:
        class Loop(LocalComm):
            active = True
        factory = Loop
    cfg = AVSiamConfig(audio_tokens=128)
    a, v = synth_inputs(cfg, 3, 50 + rank)
    a, v = a.cuda(), v.cuda()
    snaps, traces = [], []
    for defer in [False, True] * args.repeat:
        m = CAVMAE_BASE(cfg=cfg, init_seed=3, init_mode="random", verbose=False, plan_seed=77 + rank).cuda()
        m.publish_grads = False
        m.defer_p2 = defer
        m.set_distributed(world, rank, factory())
        if args.p1_only:
            out = m(a, v, mae_loss_weight=0, contrast_loss_weight=1)
            out[0].backward()
        elif args.p2_only:
            out = m(a, v, mae_loss_weight=1, contrast_loss_weight=0)
            out[0].backward()
        else:
            train_step(m, a, v, args.lr)
        m.flush_deferred()
        if args.trace:                                # checksums along the MAE pass: where does an outlier run first leave the others?
            eng = m._engine("mae", 3)
            st = eng.st_t if eng.grouped else eng.st_v
            tr = {"plan_tok": int(eng.row_tok_all.long().sum()), "plan_src": int(eng.src_row.long().sum()), "mask": float(eng.mask_all.sum())}
            for i in range(0, st.nblocks + 1):
                tr[f"tower_x{i}"] = float(st.x[i][:st.rows].double().sum())
            tr["tower_qkv0"] = float(st.qkv[0][:st.rows].double().sum()); tr["tower_att0"] = float(st.att[0][:st.rows].double().sum())
            tr["tower_ln1_0"] = float(st.ln1[0][:st.rows].double().sum()); tr["tower_xmid0"] = float(st.xmid[0][:st.rows].double().sum())
            tr["tower_fc1_0"] = float(st.fc1[0][:st.rows].double().sum()); tr["tower_act0"] = float(st.act[0][:st.rows].double().sum())
            tr["mm_out"] = float(eng.st_mm.out[:eng.st_mm.rows].double().sum())
            sm = eng.st_mm
            for i in range(sm.nblocks):
                for nm in ("x", "ln1", "qkv", "att", "xmid", "ln2", "fc1", "act", "lse"):
                    t = getattr(sm, nm)[i]
                    tr[f"mm{i}_{nm}"] = float((t[:, :sm.rows] if nm == "lse" else t[:sm.rows]).double().sum())
                tr[f"mm{i}_stats"] = [float(x[:sm.rows].double().sum()) for x in sm.stats[i]]
            tr["fstat_a"] = [float(x[:eng.rows_a].double().sum()) for x in eng.fstat_a]; tr["fstat_v"] = [float(x[:eng.rows_v].double().sum()) for x in eng.fstat_v]
            for i in range(0, eng.st_dec.nblocks + 1):
                tr[f"dec_x{i}"] = float(eng.st_dec.x[i][:eng.st_dec.rows].double().sum())
            tr["losses"] = [float(x) for x in eng.losses.tolist()]
            tr["cols_a"] = float(eng.emb_a.cols[:eng.emb_a.rows].double().sum()); tr["cols_v"] = float(eng.emb_v.cols[:eng.emb_v.rows].double().sum())
            traces.append(tr)
        b1, b2 = m.arena.range[P1]
        b12, end = m.arena.range[P2]
        m1 = m._opt_state[P1]["m"].clone() if P1 in m._opt_state else None      # 0.05 x the contrastive pass's reduced gradient
        snaps.append((defer, m.arena.g[:end].clone(), m1))
        names = [(n, m.arena.offset[n], m.arena.offset[n] + math.prod(m.arena.info[n].shape)) for n in m.arena.names if m.arena.info[n].live]
        del m
    rel = lambda x, y: float((x.double() - y.double()).norm() / x.double().norm())
    if True:
        for i in range(len(snaps)):
            for j in range(i + 1, len(snaps)):
                gi, gj = snaps[i][1], snaps[j][1]
                worst = max(rel(gi[b1:b12], gj[b1:b12]), rel(gi[b12:b2], gj[b12:b2]), rel(gi[b2:end], gj[b2:end]))
                if snaps[i][2] is not None:
                    worst = max(worst, rel(snaps[i][2], snaps[j][2]))
                if worst > 1e-5:                      # an event: say which parameters carry it (g of the step's end; m1 = pass 1's gradient image)
                    print(f"[rank {rank}] EVENT run {i} (defer={snaps[i][0]}) vs run {j} (defer={snaps[j][0]}): worst segment rel {worst:.3e}", flush=True)
                    for what, xi, xj, base in (("g", gi, gj, 0), ("m1", snaps[i][2], snaps[j][2], b1)):
                        if xi is None:
                            continue
                        bad = []
                        for n, lo, hi in names:
                            lo2, hi2 = lo - base, hi - base
                            if lo2 < 0 or hi2 > xi.numel():
                                continue
                            nrm = float(xi[lo2:hi2].double().norm())
                            if nrm > 0:
                                r = float((xi[lo2:hi2].double() - xj[lo2:hi2].double()).norm()) / nrm
                                if r > 1e-5:
                                    bad.append((r, n))
                        bad.sort(reverse=True)
                        print(f"[rank {rank}]   {what}: {len(bad)} tensors differ > 1e-5; worst: " + ", ".join(f"{n} {r:.2e}" for r, n in bad[:12]), flush=True)
                if rank != 0:
                    continue
                print(f"defer={snaps[i][0]!s:5} vs defer={snaps[j][0]!s:5}: g rel  p1-only {rel(gi[b1:b12], gj[b1:b12]):.3e}  shared {rel(gi[b12:b2], gj[b12:b2]):.3e}  "
                      f"mae-only {rel(gi[b2:end], gj[b2:end]):.3e}   max|d| shared {float((gi[b12:b2] - gj[b12:b2]).abs().max()):.3e} of max|g| {float(gi[b12:b2].abs().max()):.3e}", flush=True)
    if args.trace:
        import collections
        keys = list(traces[0].keys())
        for k in keys:
            vals = [repr(t[k]) for t in traces]
            c = collections.Counter(vals)
            if len(c) > 1:
                major = c.most_common(1)[0][0]
                odd = [(i, v) for i, v in enumerate(vals) if v != major]
                print(f"[rank {rank}] TRACE {k}: majority {major}; other runs: {odd[:6]}", flush=True)
        print(f"[rank {rank}] TRACE done ({len(traces)} runs, {len(keys)} quantities)", flush=True)
    if world > 1:
        dist.destroy_process_group()
    q.put(rank)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--procs", type=int, default=1)
    ap.add_argument("--lr", type=float, default=1e-6)
    ap.add_argument("--p1-only", action="store_true")
    ap.add_argument("--repeat", type=int, default=2)
    ap.add_argument("--trace", action="store_true", help="checksums of the MAE pass's forward buffers per run; reports where runs differ")
    ap.add_argument("--p2-only", action="store_true")
    args = ap.parse_args()
    import torch.multiprocessing as mp
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=worker, args=(r, args.procs, 29795, args, q)) for r in range(args.procs)]
    for p in procs:
        p.start()
    for _ in procs:
        q.get(timeout=1200)
    for p in procs:
        p.join(timeout=60)


if __name__ == "__main__":
    main()
