"""Data-parallel path on CPU with the gloo backend, world_size 2.

1. The oracle with the reference's GatherLayer semantics on 2 ranks reproduces the golden vectors the unmodified
   reference produced under 2-rank gloo (tests/golden/c_w2_b3_r{0,1}.npz).
2. The build's DP scheme - all-gather of the embeddings WITHOUT a backward collective (own slice x W, because every
   rank computes the identical global loss: gather_layer.py:35-37 sums W identical copies) followed by ONE
   all-reduce(SUM) over the contiguous live range of the flat gradient arena and a 1/W scale - yields exactly the
   gradient of the global loss, on a small model, through the same arena/all-reduce code the GPU path uses.
"""
import os
import random

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from avsiam_amd.config import AVSiamConfig
from avsiam_amd.weights import synth_inputs, synth_state
from tests.helpers import check_grads_against_golden, golden_plan, load_golden


def _init(rank, world, port):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    torch.set_num_threads(max(1, 8 // world))


def _golden_worker(rank, world, port, q):
    try:
        from oracle import ref_cpu
        _init(rank, world, port)
        d = load_golden(f"c_w2_b3_r{rank}")
        cfg = AVSiamConfig()
        a, v = synth_inputs(cfg, int(d["batch"]), int(d["input_seed"]))
        P = {k: t.clone().requires_grad_(True) for k, t in synth_state(cfg, int(d["weight_seed"]), "random", include_dead=False).items()}
        extras = {}
        out = ref_cpu.forward(P, cfg, a, v, golden_plan(d), mae_loss_weight=0, contrast_loss_weight=1, extras=extras)
        out[0].backward()
        got = np.array([out[i].item() for i in (0, 1, 2, 3, 4, 7)])
        np.testing.assert_allclose(got, d["out_scalars"], rtol=1e-5, atol=1e-6)
        np.testing.assert_allclose(extras["logits"].detach().numpy(), d["logits"], rtol=1e-4, atol=2e-4)
        grads = {k: p.grad for k, p in P.items()}
        from avsiam_amd.param_spec import build_spec
        for s in build_spec(cfg):
            grads.setdefault(s.name, None)
        check_grads_against_golden(d, grads, rel_l2=1e-4)
        q.put((rank, "ok"))
    except Exception as e:  # pragma: no cover
        import traceback
        q.put((rank, traceback.format_exc()))
    finally:
        if dist.is_initialized():
            dist.destroy_process_group()


def _run(worker, world=2, port=29731, args=()):
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=worker, args=(r, world, port, q) + args) for r in range(world)]
    for p in procs:
        p.start()
    res = [q.get(timeout=600) for _ in range(world)]
    for p in procs:
        p.join(timeout=60)
    for rank, msg in res:
        assert msg == "ok", f"rank {rank}: {msg}"


def test_oracle_gatherlayer_matches_reference_golden_w2():
    _run(_golden_worker, port=29731)


SMALL = dict(embed_dim=128, depth=2, num_heads=2, dec_dim=64, dec_depth=1, dec_heads=2, audio_tokens=64, n_classes=16)


def _dp_worker(rank, world, port, q):
    try:
        from oracle import ref_cpu
        from avsiam_amd.maskplan import make_contrastive_plan
        from avsiam_amd.models import CAVMAE_BASE
        from avsiam_amd.param_spec import P1
        _init(rank, world, port)
        cfg = AVSiamConfig(**SMALL)
        B = 3                                              # per-GPU batch is fixed (weak scaling); the PLANS differ per rank
        model = CAVMAE_BASE(cfg=cfg, init_seed=11, init_mode="random", verbose=False)
        from tests.helpers import HostStagedComm
        comm = HostStagedComm()
        model.set_distributed(world, rank, comm)
        arena = model.arena
        # every rank knows all inputs/plans so it can also evaluate the single-process global reference
        ins = [synth_inputs(cfg, B, 50 + r) for r in range(world)]
        plans = [make_contrastive_plan(cfg, B, torch.Generator().manual_seed(r), random.Random(r)) for r in range(world)]
        P = {k: p.detach().clone().requires_grad_(True) for k, p in model._params.items() if arena.info[k].live}
        # ---- global reference: one process, all W*B samples; per-rank group structure kept by evaluating the two
        # encoders per rank and one global loss (what W ranks + GatherLayer + DDP-mean compute together)
        reps = [ref_cpu.forward_encoder_mmixed(P, cfg, a, v, pl) for (a, v), pl in zip(ins, plans)]
        ca = torch.cat([r[0] for r in reps]); cv = torch.cat([r[1] for r in reps])
        loss, _, _ = ref_cpu.contrastive(ca.mean(1), cv.mean(1), cfg.temperature)
        loss.backward()
        want = {k: p.grad.clone() for k, p in P.items() if p.grad is not None}
        # ---- the build's DP scheme on this rank
        for p in P.values():
            p.grad = None
        ca_r, cv_r = ref_cpu.forward_encoder_mmixed(P, cfg, ins[rank][0], ins[rank][1], plans[rank])
        own = torch.cat([ca_r.mean(1), cv_r.mean(1)])                       # [2B, D] like ContrastivePass.reps
        allr = torch.zeros(world * 2 * B, cfg.embed_dim)
        dist.all_gather_into_tensor(allr, own.detach().contiguous())        # c2, no autograd
        allr = allr.view(world, 2 * B, cfg.embed_dim)
        A = allr[:, :B].reshape(world * B, -1).clone().requires_grad_(True)
        V = allr[:, B:].reshape(world * B, -1).clone().requires_grad_(True)
        l2, _, _ = ref_cpu.contrastive(A, V, cfg.temperature)
        assert abs(l2.item() - loss.item()) < 1e-6
        l2.backward()
        dA, dV = A.grad[rank * B:(rank + 1) * B] * world, V.grad[rank * B:(rank + 1) * B] * world   # own slice x W (c3 elided)
        own.backward(torch.cat([dA, dV]))
        g = arena.ensure_grads()
        g.zero_()
        for k, p in P.items():
            if p.grad is not None:
                arena.gview(k).copy_(p.grad)
        # c1: the gradients were written into the arena by hand, so the reduction runs here: all-reduce(SUM) over the live
        # range and, with average=True, DDP's mean (the fused training step keeps the 1/W for the Adam kernel instead)
        model.allreduce_grads(P1, average=True, already_reduced=False)
        lo, hi = arena.range[P1]
        assert sum(comm.messages) == hi - lo and all(f == 1.0 for f in model._grad_scale.values())
        for k, w in want.items():
            got = arena.gview(k)
            assert torch.allclose(got, w, rtol=2e-4, atol=1e-7), (k, float((got - w).abs().max()))
        # parameters outside the pass-1 live range were not touched by the collective
        assert float(g[hi:].abs().max()) == 0.0
        # One backward over BOTH passes leaves the SUM in the union of the two live ranges and owes DDP's 1/W everywhere; settling
        # it pass by pass must scale the parameters the passes share ONCE (the factor is tracked per arena segment)
        from avsiam_amd.param_spec import P2
        lo1, hi2 = arena.range[P1][0], arena.range[P2][1]
        g[lo1:hi2] = float(world)
        model._reduced = {P1: True, P2: True}
        model._owe(P1, 1.0 / world)
        model._owe(P2, 1.0 / world)
        model.allreduce_grads(P1)
        model.allreduce_grads(P2)
        assert torch.equal(g[lo1:hi2], torch.ones(hi2 - lo1)), (float(g[lo1:hi2].min()), float(g[lo1:hi2].max()))
        q.put((rank, "ok"))
    except Exception:  # pragma: no cover
        import traceback
        q.put((rank, traceback.format_exc()))
    finally:
        if dist.is_initialized():
            dist.destroy_process_group()


def test_dp_scheme_equals_global_gradient_w2():
    _run(_dp_worker, port=29741)


@pytest.mark.parametrize("world,port", [(4, 29745), (8, 29749)])
def test_dp_scheme_equals_global_gradient_w4_w8(world, port):
    """The same algebra at the world sizes the scaling bench runs (own-slice x W, SUM all-reduce over the live range, 1/W):
    every rank draws its own mask plan, so the per-rank group structures differ."""
    _run(_dp_worker, world=world, port=port)


def _reducer_worker(rank, world, port, q):
    """comm.GradReducer under a real backend: chunks declared in backward order, some never declared, must equal ONE
    all-reduce of the whole range - and must leave everything outside [lo, hi) alone."""
    try:
        from avsiam_amd.comm import GradReducer, TorchDistComm
        _init(rank, world, port)
        comm = TorchDistComm()
        n, lo, hi = 10_000, 100, 9_000
        g = torch.arange(n, dtype=torch.float32) * (rank + 1)
        want = g.clone()
        want[lo:hi] = torch.arange(n, dtype=torch.float32)[lo:hi] * sum(r + 1 for r in range(world))
        red = GradReducer(comm, g, lo, hi, min_elems=1500, overlap=True)
        for a, b in ((8000, 8500), (7000, 8000), (4000, 5000), (3000, 4000), (50, 300)):      # touching, out of order, clipped
            red.ready(a, b)
        red.finish()
        assert torch.equal(g, want), float((g - want).abs().max())
        assert 2 <= red.messages <= 8
        g2 = torch.ones(n) * (rank + 1)
        blocking = GradReducer(comm, g2, lo, hi, overlap=False)
        blocking.ready(200, 300)
        blocking.finish()
        assert blocking.messages == 1 and float(g2[lo]) == sum(r + 1 for r in range(world)) and float(g2[0]) == rank + 1
        # bf16 on the wire: the sum of the ranks' bf16-rounded chunks, written back as fp32; outside [lo, hi) untouched
        gen = torch.Generator().manual_seed(5 + rank)
        g3 = torch.randn(n, generator=gen)
        mine = g3.clone()
        parts = [torch.randn(n, generator=torch.Generator().manual_seed(5 + r)) for r in range(world)]
        exact = sum(parts)
        wire = GradReducer(comm, g3, lo, hi, min_elems=1500, overlap=True, wire="bf16")
        for a, b in ((8000, 8500), (7000, 8000), (4000, 5000)):
            wire.ready(a, b)
        wire.finish()
        assert torch.equal(g3[:lo], mine[:lo]) and torch.equal(g3[hi:], mine[hi:])
        assert torch.equal(g3[lo:hi], g3[lo:hi].bfloat16().float())                      # what arrived is a bf16 value
        err = (g3[lo:hi] - exact[lo:hi]).abs().max().item()
        assert err <= 2.0 ** -7 * world * exact[lo:hi].abs().max().item(), err             # bf16 rounding of inputs and of the partial sums
        assert torch.dot(g3[lo:hi], exact[lo:hi]) / (g3[lo:hi].norm() * exact[lo:hi].norm()) > 0.99999
        # deferral (AVSIAM_DP_DEFER): no message straddles the boundary; finish(defer_from=boundary) waits for the messages below it
        # and leaves the ones above pending until wait_deferred(); the result equals one all-reduce of the range
        g4 = torch.arange(n, dtype=torch.float32) * (rank + 1)
        bnd = 4500
        dfr = GradReducer(comm, g4, lo, hi, min_elems=1500, overlap=True, boundary=bnd)
        for a, b in ((8000, 8500), (7000, 8000), (4000, 5000), (3000, 4000)):          # (4000, 5000) straddles: sent as two messages
            dfr.ready(a, b)
        dfr.finish(defer_from=bnd)
        assert dfr.deferred and all(a >= bnd for _, a, _ in dfr.deferred) and not dfr.handles
        below = sum(1 for _, a, _ in dfr.deferred)
        assert 1 <= below < dfr.messages
        dfr.wait_deferred()
        assert not dfr.deferred and torch.equal(g4, want), float((g4 - want).abs().max())
        with pytest.raises(AssertionError):
            GradReducer(comm, g4.clone(), lo, hi, boundary=bnd).finish(defer_from=bnd + 1)
        # deferral with the bf16 wire and ONE staging buffer shared by two reducers (the model keeps one across steps): the
        # MAE pass's reducer [3000, 9000) leaves its messages above the boundary staged and in flight; the next contrastive reducer
        # [100, 4500) stages its own range in the same buffer.  Indexed by absolute arena offset the two never alias, so the deferred
        # values that land in g afterwards are still the MAE pass's sums.
        shared = torch.zeros(n, dtype=torch.bfloat16)
        parts5 = [torch.randn(n, generator=torch.Generator().manual_seed(40 + r)) for r in range(world)]
        g5 = parts5[rank].clone()
        p2 = GradReducer(comm, g5, 3000, hi, min_elems=1500, overlap=True, wire="bf16", staging=shared, staging_lo=0, boundary=bnd)
        for a, b in ((8000, 8500), (5000, 8000), (3000, 5000)):
            p2.ready(a, b)
        p2.finish(defer_from=bnd)
        assert p2.staging is shared and p2.deferred
        want_hi = sum(p.bfloat16() for p in parts5).float()[bnd:hi] if world == 2 else None      # two addends: one rounding, order-free
        g5[lo:bnd] = parts5[rank][lo:bnd] * 3.0                                                    # "the next contrastive backward"
        p1 = GradReducer(comm, g5, lo, bnd, min_elems=1500, overlap=True, wire="bf16", staging=shared, staging_lo=0)
        p1.ready(lo, bnd)
        p1.finish()
        assert p1.staging is shared
        p2.wait_deferred()
        if want_hi is not None:
            assert torch.equal(g5[bnd:hi], want_hi), "the deferred bf16 messages were overwritten by the next reducer's staging"
        # a shared buffer that does not cover the reducer's range in absolute offsets is not used
        small = GradReducer(comm, g5.clone(), lo, hi, wire="bf16", staging=torch.zeros(hi - lo, dtype=torch.bfloat16), staging_lo=0)
        assert small.staging.numel() == hi - lo and small.staging_lo == lo
        q.put((rank, "ok"))
    except Exception:  # pragma: no cover
        import traceback
        q.put((rank, traceback.format_exc()))
    finally:
        if dist.is_initialized():
            dist.destroy_process_group()


def test_grad_reducer_chunks_equal_one_allreduce_w2():
    _run(_reducer_worker, port=29753)


def test_grad_reducer_rejects_overlapping_ranges():
    from avsiam_amd.comm import GradReducer

    class Fake:
        world, rank = 2, 0

        def all_reduce_async(self, t):
            class H:
                def wait(self):
                    pass
            return H()

    red = GradReducer(Fake(), torch.zeros(100), 0, 100, min_elems=1000, overlap=True)
    red.ready(0, 50)
    red.ready(40, 60)
    with pytest.raises(AssertionError):
        red.finish()
