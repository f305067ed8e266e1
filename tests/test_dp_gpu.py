"""The data-parallel path on the REAL kernels: 2 processes share the single GPU of the test box (the collectives are
injected: tests.helpers.HostStagedComm = gloo through the host - RCCL refuses two ranks on one device), each runs the
contrastive pass of its own batch, and the results are
compared with what the unmodified reference produced with 2 gloo ranks (tests/golden/c_w2_b3_r{0,1}.npz): the global
[6,6] logits, the loss, and every live tensor's LOCAL gradient (before the mean all-reduce) - i.e. the build's
"own slice x W, no backward collective" must equal GatherLayer's all-reduce + slice (gather_layer.py:35-37).
Then the flat all-reduce + 1/W must make both ranks hold identical averaged gradients."""
import os

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from avsiam_amd.config import AVSiamConfig
from avsiam_amd.weights import synth_inputs
from tests.helpers import golden_plan, load_golden

pytestmark = pytest.mark.gpu


def _worker(rank, world, port, q):
    try:
        os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
        dist.init_process_group("gloo", rank=rank, world_size=world)
        from avsiam_amd.models import CAVMAE_BASE
        from avsiam_amd.param_spec import P1
        d = load_golden(f"c_w2_b3_r{rank}")
        cfg = AVSiamConfig()
        B = int(d["batch"])
        a, v = synth_inputs(cfg, B, int(d["input_seed"]))
        m = CAVMAE_BASE(cfg=cfg, init_seed=int(d["weight_seed"]), init_mode="random", verbose=False).cuda()
        from tests.helpers import HostStagedComm
        comm = HostStagedComm()
        m.set_distributed(world, rank, comm)
        m.reduce_in_backward = False                    # first look at the LOCAL gradients, as the golden file holds them
        out = m(a.cuda(), v.cuda(), mae_loss_weight=0, contrast_loss_weight=1, mask_plan=golden_plan(d))
        out[0].backward()
        got = np.array([out[i].item() for i in (0, 4)])
        np.testing.assert_allclose(got, d["out_scalars"][[0, 4]], rtol=2e-3)
        eng = m._engine("contrastive", B)
        np.testing.assert_allclose(eng.total.cpu().numpy(), d["logits"], atol=0.01)
        from tests.helpers import gpu_grads_vs_golden
        gpu_grads_vs_golden(d, lambda n: m._params[n].grad, f"golden_c_w2_b3_r{rank}", l2_rel=0.01, samp_rel=0.3, sum_rel=0.25)
        # c1: one all-reduce over the live range, then 1/W -> both ranks hold the same mean gradient
        m.allreduce_grads(P1, average=True)
        lo, hi = m.arena.range[P1]
        assert comm.messages == [hi - lo]
        mean = m.arena.g[lo:hi].cpu()
        other = [torch.empty_like(mean) for _ in range(world)]
        dist.all_gather(other, mean)
        assert torch.equal(other[0], other[1])
        assert float(mean.abs().sum()) > 0
        # the default path: loss.backward() reduces by itself, in chunks declared by the backward schedule (comm.GradReducer),
        # and .grad then holds DDP's mean - equal to the one-message result above up to the order of the fp32 atomics
        m.reduce_in_backward = True
        comm.messages.clear()
        out = m(a.cuda(), v.cuda(), mae_loss_weight=0, contrast_loss_weight=1, mask_plan=golden_plan(d))
        out[0].backward()
        assert len(comm.messages) >= 2 and sum(comm.messages) == hi - lo, comm.messages
        again = m.arena.g[lo:hi].cpu()
        cos = float(torch.dot(again.double(), mean.double()) / (again.double().norm() * mean.double().norm()))
        assert cos > 0.99999 and abs(float(again.norm() / mean.norm()) - 1) < 1e-4, cos
        g0 = m._params["vit_base.blocks.3.mlp.fc1.weight"].grad
        assert g0.data_ptr() == m.arena.gview("vit_base.blocks.3.mlp.fc1.weight").data_ptr()
        m.allreduce_grads(P1)                            # nothing left to do: no further message
        assert sum(comm.messages) == hi - lo
        q.put((rank, "ok"))
    except Exception:  # pragma: no cover
        import traceback
        q.put((rank, traceback.format_exc()))
    finally:
        if dist.is_initialized():
            dist.destroy_process_group()


def test_two_ranks_on_one_gpu_match_reference_w2_golden():
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker, args=(r, 2, 29761, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = [q.get(timeout=900) for _ in range(2)]
    for p in procs:
        p.join(timeout=120)
    for rank, msg in res:
        assert msg == "ok", f"rank {rank}: {msg}"


def _defer_worker(rank, world, port, q):
    """AVSIAM_DP_DEFER: the MAE-only parameters' all-reduce stays in flight after backward and their Adam update is applied before
    the next MAE forward - N training steps must leave the same weights as the undeferred schedule (the same per-element updates;
    run-to-run only the order of the weight-gradient atomics differs), and both ranks must hold bit-identical weights."""
    try:
        os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
        dist.init_process_group("gloo", rank=rank, world_size=world)
        from avsiam_amd.models import CAVMAE_BASE
        from avsiam_amd.param_spec import P1, P2
        from avsiam_amd.traintest_cavmae_base import train_step
        from tests.helpers import HostStagedComm
        cfg = AVSiamConfig(audio_tokens=128)
        B = 3
        a, v = synth_inputs(cfg, B, 50 + rank)
        a, v = a.cuda(), v.cuda()
        finals = []
        for defer in (False, False, True):
            m = CAVMAE_BASE(cfg=cfg, init_seed=3, init_mode="random", verbose=False, plan_seed=77 + rank).cuda()
            m.publish_grads = False
            m.defer_p2 = defer
            comm = HostStagedComm()
            m.set_distributed(world, rank, comm)
            for step in range(3):
                train_step(m, a, v, 1e-3)
                if defer:
                    assert m._deferred is not None and "adam" in m._deferred          # the MAE-only update is pending between steps
            lo, b = m.arena.range[P2][0], m.arena.range[P1][1]
            sd = m.state_dict()                                                        # flushes the pending update
            assert m._deferred is None
            if defer:
                assert any(n > 0 for n in comm.messages) and m.last_reduce_messages >= 2
            w = m.arena.p[:m.arena.live_end].detach().cpu().clone()
            other = [torch.empty_like(w) for _ in range(world)]
            dist.all_gather(other, w)
            assert torch.equal(other[0], other[1]), "ranks diverged"
            finals.append(w)
            assert len(sd) == 963
            del m
        # Two runs of the SAME schedule differ, too: the weight-gradient atomics land in another order, and Adam turns a gradient
        # element of noise level into a +-lr step.  That floor (run 0 vs run 1, both undeferred) is the yardstick: the deferred run
        # must sit at the same distance - over the whole arena and inside the MAE-only segment the deferral touches.
        def rel(x, y, sl=slice(None)):
            return float((x[sl].double() - y[sl].double()).norm() / x[sl].double().norm())
        seg = slice(b, finals[0].numel())
        floor, floor_seg = rel(finals[0], finals[1]), rel(finals[0], finals[1], seg)
        got, got_seg = rel(finals[0], finals[2]), rel(finals[0], finals[2], seg)
        assert got <= 2.0 * floor + 1e-6 and got_seg <= 2.0 * floor_seg + 1e-6, (floor, got, floor_seg, got_seg)
        assert floor < 0.05, floor
        q.put((rank, "ok"))
    except Exception:  # pragma: no cover
        import traceback
        q.put((rank, traceback.format_exc()))
    finally:
        if dist.is_initialized():
            dist.destroy_process_group()


def test_deferred_mae_only_update_leaves_the_same_weights():
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_defer_worker, args=(r, 2, 29765, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = [q.get(timeout=900) for _ in range(2)]
    for p in procs:
        p.join(timeout=120)
    for rank, msg in res:
        assert msg == "ok", f"rank {rank}: {msg}"
