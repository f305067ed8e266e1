"""The reference training step (traintest_cavmae_base.py:131-152) end to end on the GPU: contrastive pass -> Adam#1 ->
MAE pass (sees the updated weights) -> Adam#2, against the oracle driven by torch.optim.Adam with the reference's
hyper-parameters; plus a short run of the train() loop and of validate()."""
import argparse
import random

import pytest
import torch

from avsiam_amd.config import AVSiamConfig
from avsiam_amd.maskplan import make_contrastive_plan, make_mae_plan
from avsiam_amd.weights import synth_inputs, synth_state

pytestmark = pytest.mark.gpu


def test_train_step_matches_oracle_adam():
    from avsiam_amd.models import CAVMAE_BASE
    from avsiam_amd.traintest_cavmae_base import train_step
    from oracle import ref_cpu
    cfg = AVSiamConfig(audio_tokens=128)
    B, lr = 4, 2e-4
    a, v = synth_inputs(cfg, B, 3)
    gen = torch.Generator().manual_seed(9)
    pm, pc = make_mae_plan(cfg, B, gen), make_contrastive_plan(cfg, B, gen, random.Random(9))
    m = CAVMAE_BASE(cfg=cfg, init_seed=21, init_mode="random", verbose=False).cuda()
    m.publish_grads = False
    before = {k: p.detach().cpu().clone() for k, p in m._params.items()}
    out = train_step(m, a.cuda(), v.cuda(), lr, plans=(pm, pc))
    torch.cuda.synchronize()
    # oracle: same sequence with torch Adam (two optimizers over the same parameters, traintest:64-66)
    torch.set_num_threads(16)
    P = {k: t.clone().requires_grad_(True) for k, t in synth_state(cfg, 21, "random", include_dead=False).items()}
    params = list(P.values())
    o1 = torch.optim.Adam(params, lr, weight_decay=5e-7, betas=(0.95, 0.999))
    o2 = torch.optim.Adam(params, lr, weight_decay=5e-7, betas=(0.95, 0.999))
    r1 = ref_cpu.forward(P, cfg, a, v, pc, mae_loss_weight=0, contrast_loss_weight=1)
    o1.zero_grad(); r1[0].backward(); o1.step()
    r2 = ref_cpu.forward(P, cfg, a, v, pm, mae_loss_weight=1, contrast_loss_weight=0)
    o2.zero_grad(); r2[0].backward(); o2.step()
    assert abs(out[3].item() - r1[4].item()) <= 2e-2 * abs(r1[4].item())            # loss_c
    assert abs(out[0].item() - r2[0].item()) <= 2e-2 * abs(r2[0].item())            # MAE loss AFTER the contrastive update
    # parameter updates: first Adam step moves every element by ~lr*sign(g); compare update directions on big tensors
    checked = 0
    for k in ("vit_base.blocks.0.attn.qkv.weight", "vit_base.blocks.11.mlp.fc2.weight", "ast_base.blocks.5.mlp.fc1.weight",
              "decoder_blocks.3.attn.proj.weight", "mm_layer_1.mlp.fc1.weight", "decoder_pred_v.weight", "vit_base.norm_a.weight",
              "vit_base.patch_embed.proj.weight"):
        d_hip = (m._params[k].detach().cpu() - before[k]).double().reshape(-1)
        d_ref = (P[k].detach() - before[k]).double().reshape(-1)
        assert d_ref.norm() > 0, k
        cos = float(torch.dot(d_hip, d_ref) / (d_hip.norm() * d_ref.norm()))
        assert cos > 0.9, (k, cos)
        assert abs(float(d_hip.norm() / d_ref.norm()) - 1) < 0.1, k
        checked += 1
    assert checked == 8
    # parameters no pass touches are not updated (Adam skips grad=None parameters)
    for k in ("vit_base.head.weight", "ast_base.pos_embed", "vit_base.blocks.0.norm1.weight"):
        assert torch.equal(m._params[k].detach().cpu(), before[k]), k


def test_train_loop_and_validate(tmp_path):
    from avsiam_amd.models import CAVMAE_BASE
    from avsiam_amd.traintest_cavmae_base import SyntheticAVLoader, train, validate
    cfg = AVSiamConfig(audio_tokens=128)
    m = CAVMAE_BASE(cfg=cfg, verbose=False, plan_seed=1)
    args = argparse.Namespace(n_epochs=1, batch_size=4, lr=1e-3, lrscheduler_start=10, lrscheduler_step=5, lrscheduler_decay=0.5,
                              n_print_steps=1, exp_dir=str(tmp_path), save_model=True, rank=0, gpu=0, world_size=1, steps_per_epoch=6,
                              masking_ratio=0.75, masking_ratio_a=0.75, mask_mode="unstructured", mae_loss_weight=3.0, contrast_loss_weight=0.01)
    val = SyntheticAVLoader(cfg, 4, 1, "cuda", seed=5)
    train(m, None, [val, None], [None, None], None, args, None)
    sd = torch.load(tmp_path / "models" / "audio_model.1.pth")
    assert len(sd) == 963 and all(k.startswith("module.") for k in sd)
    ev = validate(m, val, None, args)
    assert all(map(lambda x: x == x, ev)) and ev[0] > 0            # finite losses
    # the saved checkpoint loads back through the reference's consumer path (strip 'module.')
    m2 = CAVMAE_BASE(cfg=cfg, verbose=False)
    missing = m2.load_state_dict({k[len("module."):]: v for k, v in sd.items()}, strict=True)
    assert not missing.missing_keys and not missing.unexpected_keys
    assert torch.equal(m2._params["decoder_embed.weight"].cpu(), m._params["decoder_embed.weight"].cpu())


@pytest.mark.parametrize("mode,recompute", [("1", "1"), ("3", "1"), ("3", "0.5")])
def test_fp8_forward_with_recompute_trains_at_the_14x14_geometry(mode, recompute):
    """The combination behind the configs[4]-like bench line (ViT-H/14 geometry, fp8 GEMMs, per-layer activation recompute - whole or a
    fraction of every stack -, device-drawn plans): a few training steps on one fixed batch stay finite and reduce the MAE loss.
    mode "1": fp8 forward; mode "3": fp8 forward, input gradients and weight gradients (the recomputed blocks re-write the e4m3 copies the
    weight gradients read)."""
    from avsiam_amd import engine
    from avsiam_amd.config import vit_huge14
    from avsiam_amd.models import CAVMAE_BASE
    from avsiam_amd.traintest_cavmae_base import train_step
    cfg = vit_huge14(frames=2, depth=2)
    a, v = synth_inputs(cfg, 4, 41)
    a, v = a.cuda(), v.cuda()
    if True:
        m = CAVMAE_BASE(cfg=cfg, init_seed=7, init_mode="random", verbose=False, plan_seed=9, fp8_mode=mode, recompute=recompute).cuda()
        m.publish_grads = False
        hist, sat = [], []
        for _ in range(10):
            out = train_step(m, a, v, 2e-4)
            hist.append([float(x.item()) for x in out])
            sat.append(m.fp8_saturation_events())
        # delayed scaling: the scales follow the weights / activations over the optimizer steps.  A tensor is clipped only when its
        # |max| more than doubles against the 16-step history within one step (margin 2); that is counted, not silent: right after the
        # random start a few tensors may do so (the decoder's mask rows leave exactly zero), none in the steady steps that follow.
        # Every GEMM of every stack is calibrated, and the state survives a checkpoint round trip into a fresh model.
        assert sat[-1] <= 4 and sat[-1] == sat[-5], sat
        st = m.fp8_state()
        assert st and all(len(v["seen"]) > 0 and float(v["q"][:, 0].max()) > 0 for v in st.values()), list(st)
        m2 = CAVMAE_BASE(cfg=cfg, init_seed=7, init_mode="random", verbose=False, plan_seed=9, fp8_mode=mode, recompute=recompute).cuda()
        m2.load_state_dict(m.state_dict())
        m2.load_fp8_state(st)
        m2.publish_grads = False
        with torch.no_grad():
            plans = m.draw_plans(4)
            o1 = m(a, v, mae_loss_weight=1, contrast_loss_weight=0, mask_plan=plans[0])
            o2 = m2(a, v, mae_loss_weight=1, contrast_loss_weight=0, mask_plan=plans[0])
        assert abs(o1[0].item() - o2[0].item()) <= 2e-3 * abs(o1[0].item()), (o1[0].item(), o2[0].item())
    assert all(x == x and abs(x) < 1e4 for h in hist for x in h), hist
    assert hist[-1][0] < hist[0][0], (hist[0], hist[-1])          # loss_mae


@pytest.mark.parametrize("mode", ["0", "3"])
def test_shared_activation_pool_is_the_same_training(mode):
    """CAVMAE_BASE(share_pass_buffers=True) (engine.BufferPool): both passes of the step take their activation buffers from the same
    memory.  From the second step on every pooled buffer of a pass holds the OTHER pass's data when its forward starts, pad rows included
    (the weight-gradient GEMMs contract over them, the fp8 calibration takes their |max|): at lr = 0 - identical weights in every step -
    the gradients of both passes must be the ones of a model with private buffers, step after step; the pool must be smaller than the
    private buffers; misuse (a combined-loss forward with gradients, a backward after the other pass's forward) must raise."""
    from avsiam_amd import engine
    from avsiam_amd.models import CAVMAE_BASE
    from avsiam_amd.traintest_cavmae_base import train_step
    cfg = AVSiamConfig(audio_tokens=128, frames=2)
    B = 5                                            # 5 x (32 + 2 x 49) rows etc.: no stack's row count is a multiple of 64
    a, v = synth_inputs(cfg, B, 17)
    a, v = a.cuda(), v.cuda()
    gen = torch.Generator().manual_seed(4)
    pm, pc = make_mae_plan(cfg, B, gen), make_contrastive_plan(cfg, B, gen, random.Random(4))
    if True:
        grads = {}
        for shared in (False, True):
            m = CAVMAE_BASE(cfg=cfg, init_seed=3, init_mode="random", verbose=False, share_pass_buffers=shared, fp8_mode=mode).cuda()
            m.publish_grads = False
            torch.cuda.synchronize()
            base = torch.cuda.memory_allocated()                  # (parameters, gradients: everything but the passes' buffers)
            out = []
            for step in range(3):
                o = train_step(m, a, v, 0.0, plans=(pm, pc))
                out.append([float(x.item()) for x in o])
                assert all(x == x for x in out[-1]), (shared, step, out[-1])
            torch.cuda.synchronize()
            grads[shared] = (m.arena.g.detach().clone(), out, torch.cuda.memory_allocated() - base)
            if shared:
                assert m._pool is not None and m._pool.nbytes() > 0
                with pytest.raises(RuntimeError, match="share_pass_buffers"):
                    m(a, v, mae_loss_weight=1, contrast_loss_weight=0.01)               # combined loss with gradients
                lc = m(a, v, mae_loss_weight=0, contrast_loss_weight=0.01, mask_plan=pc)[0]
                m(a, v, mae_loss_weight=1, contrast_loss_weight=0, mask_plan=pm)        # the MAE pass takes the memory over
                with pytest.raises(RuntimeError, match="shared activation pool"):
                    lc.backward()
                with torch.no_grad():                                                   # without gradients the combined forward is fine
                    both = m(a, v, mae_loss_weight=1, contrast_loss_weight=0.01, mask_plan={"mae": pm, "contrastive": pc})
                assert abs(both[1].item() - out[-1][0]) <= 2e-3 * abs(out[-1][0]), (both[1].item(), out[-1][0])
                assert abs(both[4].item() - 0.01 * out[-1][3]) <= 2e-3 * 0.01 * abs(out[-1][3]) + 1e-7, (both[4].item(), out[-1][3])      # (train_step's pass 1 has weight 1)
            del m
        g0, o0, mem0 = grads[False]
        g1, o1, mem1 = grads[True]
        # (fp8: the first step calibrates, later steps run on delayed scales - step by step the two models still do the same thing)
        for s0, s1 in zip(o0, o1):
            for x, y in zip(s0, s1):
                assert abs(x - y) <= 1e-3 * abs(x) + 1e-7, (o0, o1)
        rel = float((g0 - g1).double().norm() / g0.double().norm())
        assert rel < (2e-2 if mode == "3" else 1e-4), rel          # bf16: the order of the fp32 atomics; fp8: amax atomics may move a scale by an ulp
        assert mem1 < 0.8 * mem0, (mem0, mem1)                    # the passes' buffers: the larger pass (+ chunk slack) instead of the sum


@pytest.mark.parametrize("mode", ["0", "3"])
def test_contrastive_training_learns_an_audio_visual_correspondence(mode):
    """End-to-end learning evidence for the contrastive branch (with i.i.d. Gaussian clips loss_c sits at ln B whatever the
    step does - the token mean erases what tells them apart).  Clips with a REAL correspondence (tests.helpers.correlated_av_batch: a shared
    latent drives a spectral envelope and a frame texture), a FRESH batch of 16 every step, 40 reference steps
    (/root/reference/src/traintest_cavmae_base.py:131-152) in bf16 and in fp8 mode 3: the InfoNCE loss must fall below 0.8 ln B and the
    retrieval accuracy must reach 4 / B at least (mean of the last 5 steps; the CPU oracle driven by torch.optim.Adam on the same data
    reaches loss_c ~0.2 / c_acc ~0.9, tools/train_sanity.py --oracle).  The control - same marginals, frames paired with ANOTHER clip's audio
    latent - must NOT: a step that ignored its inputs, or a broken gradient, fails one of the two."""
    import math
    from avsiam_amd import engine
    from avsiam_amd.models import CAVMAE_BASE
    from avsiam_amd.traintest_cavmae_base import train_step
    from tests.helpers import correlated_av_batch, record_margin
    cfg = AVSiamConfig(audio_tokens=128, frames=1)
    B, steps = 16, 40
    if True:
        res = {}
        for shuffled in (False, True):
            m = CAVMAE_BASE(cfg=cfg, init_seed=0, init_mode="init", verbose=False, plan_seed=3, fp8_mode=mode).cuda()
            m.publish_grads = False
            hist = []
            for step in range(steps):
                a, v = correlated_av_batch(cfg, B, seed=step, shuffle_pairs=shuffled)
                out = train_step(m, a.cuda(), v.cuda(), 2e-4)
                hist.append([float(x.item()) for x in out])
            assert all(x == x and abs(x) < 1e4 for h in hist for x in h), hist
            res[shuffled] = (sum(h[3] for h in hist[-5:]) / 5, sum(h[4] for h in hist[-5:]) / 5, hist[0][3], hist[0][0], hist[-1][0])
            if mode == "3":
                assert m.fp8_saturation_events() <= 4
            del m
        (lc, acc, lc0, lm0, lm1), (lc_s, acc_s, _, _, _) = res[False], res[True]
        record_margin(f"train_sanity_correlated_{'bf16' if mode == '0' else 'fp8m3'}", loss_c_first=lc0, loss_c_last5=lc, c_acc_last5=acc,
                      loss_c_last5_shuffled_control=lc_s, c_acc_last5_shuffled_control=acc_s, loss_mae_first=lm0, loss_mae_last=lm1, ln_B=math.log(B))
        assert lc < 0.8 * math.log(B) and acc >= 4.0 / B, (lc, acc)
        # (the MAE loss is recorded, not asserted: on fresh batches it follows each batch's latent energy - the reconstruction side of
        #  training is pinned on a fixed batch by test_fp8_forward_with_recompute_trains_at_the_14x14_geometry and the oracle-Adam step test)
        assert not (lc_s < 0.8 * math.log(B) and acc_s >= 4.0 / B), ("the control learned a correspondence that is not there", lc_s, acc_s)


@pytest.mark.parametrize("fp8", ["0", "3"])
def test_graphed_step_equals_the_eager_step(fp8):
    """(fp8 "3": the same in fp8 mode 3 - the delayed-scaling records live on the device and are advanced by nodes of the graph; the calibration
    step is one of the warm-up steps.)
    graph_step.GraphedTrainStep at the reference's launch geometry (batch 4, one frame: run_pretrain_base.sh:30-31): the step replayed
    from one captured hipGraph draws the same plans (device-resident Philox key, host part in front of the replay), applies the same
    Adam updates (device-resident step count) and returns the same losses as the eager step - two models from the same seeds, one
    stepped eagerly, one replayed; then eager and replayed steps mixed on one model."""
    from avsiam_amd.graph_step import GraphedTrainStep
    from avsiam_amd.models import CAVMAE_BASE
    from avsiam_amd.param_spec import P1, P2
    from avsiam_amd.traintest_cavmae_base import train_step
    from avsiam_amd import engine
    cfg = AVSiamConfig(audio_tokens=128, frames=1)
    B = 4
    a, v = synth_inputs(cfg, B, 11)
    a, v = a.cuda(), v.cuda()

    def fresh():
        m = CAVMAE_BASE(cfg=cfg, init_seed=2, init_mode="random", verbose=False, plan_seed=21, fp8_mode=fp8).cuda()
        m.publish_grads = False
        return m

    def plans_equal(x, y):
        px, py = x.last_plans(B), y.last_plans(B)
        assert torch.equal(px["mae"].ids_keep_a, py["mae"].ids_keep_a) and torch.equal(px["mae"].ids_keep_v, py["mae"].ids_keep_v)
        assert torch.equal(px["contrastive"].a_group, py["contrastive"].a_group) and torch.equal(px["contrastive"].v_group, py["contrastive"].v_group)
        assert all(torch.equal(p, q) for p, q in zip(px["contrastive"].a_keep, py["contrastive"].a_keep))

    def close(oe, og):
        # the reconstruction losses agree to the order of the fp32 atomics (2e-3 with Adam's sign sensitivity, docs/rounds/r04.md item 1); the
        # InfoNCE loss of a batch of 4 the model is memorising is 1e-4 .. 5e-2 and moves by a few percent with that same noise (tau = 0.05)
        mae = all(abs(x - y) <= 2e-3 * abs(x) + 1e-6 for x, y in zip(oe[:3], og[:3]))
        return mae and abs(oe[3] - og[3]) <= 0.05 * abs(oe[3]) + 5e-3 and oe[4] == og[4]

    try:
        me, mg = fresh(), fresh()
        for _ in range(2):
            train_step(me, a, v, 2e-4)
        gs = GraphedTrainStep(mg, a, v, 2e-4, warmup=2)
        assert gs.kernel_nodes > 300
        for i in range(3):
            oe = [float(x.item()) for x in train_step(me, a, v, 2e-4)]
            og = [float(x.item()) for x in gs.step()]
            plans_equal(me, mg)
            assert close(oe, og), (i, oe, og)
        assert me._opt_state[P1]["step"] == mg._opt_state[P1]["step"] == 5 and me._opt_state[P2]["step"] == mg._opt_state[P2]["step"] == 5
        rel = float((me.arena.p - mg.arena.p).double().norm() / me.arena.p.double().norm())
        assert rel < (2e-3 if fp8 == "0" else 6e-3), rel      # measured 3.6e-4 (bf16) / 3e-4 .. 2.9e-3 (fp8 mode 3, box to box) after five updates: Adam's
                                      # first steps move every weight by +-lr by the SIGN of its gradient element, and the order of the fp32 atomics
                                      # decides the sign of the smallest ones (docs/rounds/r04.md item 1); e5m2 gradient operands leave more of them near zero
        # mixed: an eager step on the graphed model, then a replay - the counters are re-written from the host state in front of every replay
        train_step(me, a, v, 2e-4); train_step(mg, a, v, 2e-4)
        oe = [float(x.item()) for x in train_step(me, a, v, 2e-4)]
        og = [float(x.item()) for x in gs.step()]
        plans_equal(me, mg)
        assert close(oe, og), (oe, og)
        # a new batch is a copy into the fixed buffers
        a2, v2 = synth_inputs(cfg, B, 12)
        a.copy_(a2.cuda()); v.copy_(v2.cuda())
        oe = [float(x.item()) for x in train_step(me, a, v, 2e-4)]
        og = [float(x.item()) for x in gs.step()]
        assert close(oe, og), (oe, og)
    finally:
        pass


@pytest.mark.parametrize("fp8", ["0", "3"])
def test_deterministic_mode_gives_bit_identical_steps(fp8):
    """EngineOptions.deterministic (what torch.use_deterministic_algorithms is to the reference's loop): every reduction into a
    parameter gradient has one writer per element and a fixed order - weight-gradient GEMMs without a split of their token rows, bias column sums /
    the LayerNorm slab reduce / the qkv-bias vector-matrix product as one block per column group, atomics-free positional scatter and un-shuffle token
    sums, the fc1 bias gradient by the column-sum kernel instead of the GEMM epilogue's atomics, everything on one stream.  Two models from the same
    seeds, three eager training steps each (device-drawn plans): losses, every weight, both Adam moments BITWISE equal - in bf16 and in fp8 mode 3
    (whose amax atomics are max-reductions, order-free).  The default mode is what makes two runs differ (atomics order x Adam's sign sensitivity,
    docs/rounds/r04.md item 1): it must agree with the deterministic result to the noise it is known to have, which also checks that the one-writer
    kernels compute the same sums."""
    from avsiam_amd import _lib
    from avsiam_amd.config import EngineOptions
    from avsiam_amd.models import CAVMAE_BASE
    from avsiam_amd.param_spec import P1, P2
    from avsiam_amd.traintest_cavmae_base import train_step
    from avsiam_amd.weights import synth_inputs
    cfg = AVSiamConfig(audio_tokens=128, frames=2)
    B = 5
    a, v = synth_inputs(cfg, B, 17)
    a, v = a.cuda(), v.cuda()

    def run(det, steps=3, lr=2e-4):
        m = CAVMAE_BASE(cfg=cfg, init_seed=4, init_mode="random", verbose=False, plan_seed=33, options=EngineOptions(fp8=fp8, deterministic=det)).cuda()
        m.publish_grads = False
        outs = [[float(x.item()) for x in train_step(m, a, v, lr)] for _ in range(steps)]
        torch.cuda.synchronize()
        return m, outs

    m1, o1 = run(True)
    m2, o2 = run(True)
    assert _lib.tuning_get("det") == 0                                  # the knob is set only while a deterministic model's backward is queued
    assert o1 == o2, (o1, o2)
    assert torch.equal(m1.arena.p, m2.arena.p) and torch.equal(m1.arena.g, m2.arena.g)
    for w in (P1, P2):
        assert torch.equal(m1._opt_state[w]["m"], m2._opt_state[w]["m"]) and torch.equal(m1._opt_state[w]["v"], m2._opt_state[w]["v"])
    # one step of the default (atomics) mode from the same start at lr = 0 (identical weights in both passes): the same gradients up to summation order
    md, od = run(False, steps=1, lr=0.0)
    m3, o3 = run(True, steps=1, lr=0.0)
    for x, y in zip(od[0][:4], o3[0][:4]):
        assert abs(x - y) <= 1e-6 * abs(x), (od, o3)                    # losses: the forward has no atomics in either mode
    g0, g1 = md.arena.g.double(), m3.arena.g.double()
    rel = float((g0 - g1).norm() / g1.norm())
    assert rel < (1e-5 if fp8 == "0" else 2e-2), rel                    # (fp8: an amax shard that lands an ulp apart moves a whole tensor's grid)
