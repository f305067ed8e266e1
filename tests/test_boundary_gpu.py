"""The drop-in boundary beyond the two single-pass calls of the training step (SURVEY.md section 8(b), (f) rows 1-2, P13, P16
and the entry point): the COMBINED forward validate() makes, validate()/train() bookkeeping, the pretrained-ViT
initialisation, an external torch optimizer, the command-line entry and the process-group setup on the real backend.
Everything runs through the C ABI on the GPU; tolerances as in tests/test_parity_gpu.py."""
import argparse
import os
import subprocess
import sys

import numpy as np
import pytest
import torch

from avsiam_amd.config import AVSiamConfig
from avsiam_amd.param_spec import P1, P2
from avsiam_amd.weights import synth_inputs, synth_state
from tests.helpers import ROOT, golden_plan, gpu_grads_vs_golden, load_golden, record_margin

pytestmark = pytest.mark.gpu

LOSS_RTOL = 3e-3            # measured worst case ~6e-4 (profiles/r02/parity_margins.json)


def _model(cfg, seed=1234, mode="random", **kw):
    from avsiam_amd.models import CAVMAE_BASE
    return CAVMAE_BASE(cfg=cfg, init_seed=seed, init_mode=mode, verbose=False, **kw).cuda()


def test_combined_forward_matches_reference_golden():
    """forward(mae_loss_weight=3.0, contrast_loss_weight=0.01) - what validate() calls (traintest_cavmae_base.py:401) - against
    the unmodified reference's outputs AND gradients for the same call (tests/golden/mc_w1_b4.npz)."""
    d = load_golden("mc_w1_b4")
    cfg = AVSiamConfig()
    B = int(d["batch"])
    wm, wc = (float(x) for x in d["loss_weights"])
    a, v = synth_inputs(cfg, B, int(d["input_seed"]))
    plan = golden_plan(d)
    m = _model(cfg, int(d["weight_seed"]))
    with torch.no_grad():
        out0 = m(a.cuda(), v.cuda(), 0.75, 0.75, mae_loss_weight=wm, contrast_loss_weight=wc, mask_plan=plan)
    out = m(a.cuda(), v.cuda(), 0.75, 0.75, mae_loss_weight=wm, contrast_loss_weight=wc, mask_plan=plan)
    got0 = np.array([out0[i].item() for i in (0, 1, 2, 3, 4, 7)])
    got = np.array([out[i].item() for i in (0, 1, 2, 3, 4, 7)])
    np.testing.assert_array_equal(got0, got)                     # no-grad and grad forwards are the same kernels
    np.testing.assert_allclose(got[:5], d["out_scalars"][:5], rtol=LOSS_RTOL, atol=1e-6)
    assert abs(got[5] - d["out_scalars"][5]) <= 1.0 / B + 1e-6
    record_margin("combined_golden", loss_rel=float(np.max(np.abs(got[:5] - d["out_scalars"][:5]) / np.abs(d["out_scalars"][:5]))))
    assert out[5] is None and out[6] is None                     # the mixed encoder's None masks win (cav_mae_base.py:722)
    assert abs(got[0] - (got[1] + got[4])) < 1e-6                # loss = loss_c + loss_mae (:739), loss_mae NOT weighted
    np.testing.assert_allclose(m._engine("contrastive", B).total.cpu().numpy(), d["logits"], atol=0.01)
    out[0].backward()
    gpu_grads_vs_golden(d, lambda n: m._params[n].grad, "combined_golden", l2_rel=0.01, samp_rel=0.3, sum_rel=2.0)


def test_combined_forward_matches_oracle():
    """Same call against the oracle at configs[0]'s shape with full per-tensor gradient comparison (both passes live in ONE
    backward: the shared range of the arena accumulates both)."""
    import random
    from avsiam_amd.maskplan import make_contrastive_plan, make_mae_plan
    from oracle import ref_cpu
    from tests.test_parity_gpu import _compare_grads
    cfg = AVSiamConfig(audio_tokens=128)
    B, wm, wc = 5, 3.0, 0.01
    a, v = synth_inputs(cfg, B, 31)
    gen = torch.Generator().manual_seed(4)
    plan = {"mae": make_mae_plan(cfg, B, gen), "contrastive": make_contrastive_plan(cfg, B, gen, random.Random(4))}
    m = _model(cfg, 555)
    out = m(a.cuda(), v.cuda(), mae_loss_weight=wm, contrast_loss_weight=wc, mask_plan=plan)
    out[0].backward()
    torch.set_num_threads(16)
    P = {k: t.clone().requires_grad_(True) for k, t in synth_state(cfg, 555, "random", include_dead=False).items()}
    ref = ref_cpu.forward(P, cfg, a, v, plan, mae_loss_weight=wm, contrast_loss_weight=wc)
    ref[0].backward()
    for i in (0, 1, 2, 3, 4):
        assert abs(out[i].item() - ref[i].item()) <= LOSS_RTOL * abs(ref[i].item()) + 1e-6, (i, out[i].item(), ref[i].item())
    _compare_grads(m, {k: p.grad for k, p in P.items()}, tag="combined_oracle")


def _args(tmp, **kw):
    d = dict(n_epochs=1, batch_size=4, lr=1e-3, lrscheduler_start=10, lrscheduler_step=5, lrscheduler_decay=0.5, n_print_steps=50,
             exp_dir=str(tmp), save_model=False, rank=0, gpu=0, world_size=1, steps_per_epoch=3, masking_ratio=0.75, masking_ratio_a=0.75,
             mask_mode="unstructured", mae_loss_weight=3.0, contrast_loss_weight=0.01)
    d.update(kw)
    return argparse.Namespace(**d)


def test_validate_uses_the_run_loss_weights_and_matches_oracle(tmp_path):
    """validate(audio_model, val_loader, val_sampler, args) (reference signature, :381): the forward gets
    args.mae_loss_weight / args.contrast_loss_weight (:401).  One batch: read the plans the device drew and feed the oracle."""
    from avsiam_amd.traintest_cavmae_base import SyntheticAVLoader, validate
    from oracle import ref_cpu
    cfg = AVSiamConfig(audio_tokens=128)
    B = 4
    m = _model(cfg, 77, plan_seed=11)
    val = SyntheticAVLoader(cfg, B, 1, "cuda", seed=5)
    args = _args(tmp_path)
    ev = validate(m, val, None, args)
    assert len(ev) == 6 and all(np.isfinite(ev))
    plans = m.last_plans(B)
    torch.set_num_threads(16)
    P = {k: t.clone() for k, t in synth_state(cfg, 77, "random", include_dead=False).items()}
    with torch.no_grad():
        ref = ref_cpu.forward(P, cfg, val.a.cpu(), val.v.cpu(), plans, mae_loss_weight=3.0, contrast_loss_weight=0.01)
    want = [ref[i].item() for i in (0, 1, 2, 3, 4)]
    for g, w in zip(ev[:5], want):
        assert abs(g - w) <= LOSS_RTOL * abs(w) + 1e-6, (ev, want)
    assert abs(ev[5] - ref[7].item()) <= 1.0 / B + 1e-6
    # the contrastive term carries its weight (x 0.01), the MAE term does not carry 3.0 (:735,739)
    args1 = _args(tmp_path, contrast_loss_weight=1.0)
    m2 = _model(cfg, 77, plan_seed=11)
    ev1 = validate(m2, val, None, args1)
    assert abs(ev1[4] * 0.01 - ev[4]) <= 1e-5 * abs(ev[4]) + 1e-9 and abs(ev1[1] - ev[1]) <= 1e-6 * abs(ev[1])
    # several batches: the mean over batches (:417-422)
    m3 = _model(cfg, 77, plan_seed=11)
    val3 = SyntheticAVLoader(cfg, B, 3, "cuda", seed=5)
    ev3 = validate(m3, val3, None, args)
    assert all(np.isfinite(ev3)) and abs(ev3[0] - ev[0]) < 0.2 * abs(ev[0])          # same data, other masks


def test_train_loop_bookkeeping(tmp_path):
    """train(): epoch means come from EVERY step even when no step of the epoch is a print step (device-side accumulation),
    best_audio_model.pth / best_optim_state.pth are written when the evaluation loss improves (:221-230), the optimizer state
    has torch.optim.Adam's format, and without a validation loader no 'best' model is invented."""
    from avsiam_amd.models import CAVMAE_BASE
    from avsiam_amd.traintest_cavmae_base import SyntheticAVLoader, train
    cfg = AVSiamConfig(audio_tokens=128)
    m = CAVMAE_BASE(cfg=cfg, verbose=False, plan_seed=1)
    args = _args(tmp_path, n_epochs=2, save_model=True, n_print_steps=50, steps_per_epoch=3)
    val = SyntheticAVLoader(cfg, 4, 1, "cuda", seed=5)
    train(m, None, [val, None], [None, None], None, args, None)
    res = np.loadtxt(tmp_path / "result.csv", delimiter=",")
    assert res.shape == (2, 10)
    # epoch 2 (global steps 3..5) holds no print step (50): its train-loss columns must still be real means
    assert (res[:, :4] > 0).all() and (res[:, 4:9] >= 0).all() and np.isfinite(res).all(), res
    assert abs(res[1, 3] - (res[1, 0] + res[1, 1])) < 1e-4        # pass-2 total = MAE a + v (contrastive weight 0 in pass 2)
    assert res[0, 9] == 1e-3
    models = tmp_path / "models"
    sd = torch.load(models / "best_audio_model.pth")
    assert len(sd) == 963 and all(k.startswith("module.") for k in sd)
    assert (models / "audio_model.1.pth").exists() and (models / "audio_model.2.pth").exists()
    osd = torch.load(models / "best_optim_state.pth", weights_only=False)
    params = list(m.parameters())
    opt = torch.optim.Adam(params, 1e-3, weight_decay=5e-7, betas=(0.95, 0.999))
    opt.load_state_dict(osd)                                       # torch's own format
    live1 = sum(1 for n, p in m._params.items() if m.arena.info[n].live & P1)
    assert len(osd["state"]) == live1
    some = next(iter(osd["state"].values()))
    assert float(some["step"]) >= 3 and some["exp_avg"].abs().sum() > 0
    # round trip through the model's own loader
    m.load_optimizer_state_dict(P1, osd)
    assert m._opt_state[P1]["step"] == int(float(some["step"]))
    # no validation loader: no best model
    m2 = CAVMAE_BASE(cfg=cfg, verbose=False, plan_seed=1)
    d2 = tmp_path / "noval"
    train(m2, None, [None, None], [None, None], None, _args(d2, steps_per_epoch=2), None)
    assert not (d2 / "models" / "best_audio_model.pth").exists()
    assert np.loadtxt(d2 / "result.csv", delimiter=",").reshape(1, 10)[0, 3] > 0


def test_train_loop_with_graph_step_equals_the_eager_loop(tmp_path):
    """train() with args.graph_step (`--graph-step`, round 5): the loop captures the step on its first batch - whose one warm-up step IS that
    batch's training step - and again when the learning rate changes (epoch 2 here), copies every other batch into the captured buffers and
    replays.  Same number of optimizer steps, same plans, the same result.csv (losses to the order of the fp32 atomics) and the same weights as
    the eager loop from the same seeds; validation between the epochs runs eagerly on the same model."""
    from avsiam_amd.models import CAVMAE_BASE
    from avsiam_amd.param_spec import P1, P2
    from avsiam_amd.traintest_cavmae_base import SyntheticAVLoader, train
    cfg = AVSiamConfig(audio_tokens=128)
    runs = {}
    for name, flag in (("eager", False), ("graph", True)):
        m = CAVMAE_BASE(cfg=cfg, init_seed=3, init_mode="random", verbose=False, plan_seed=7)
        d = tmp_path / name
        args = _args(d, n_epochs=2, steps_per_epoch=3, lrscheduler_start=1, lrscheduler_step=1, graph_step=flag)      # lr halves after epoch 1
        val = SyntheticAVLoader(cfg, 4, 1, "cuda", seed=5)
        train(m, None, [val, None], [None, None], None, args, None)
        runs[name] = (m, np.loadtxt(d / "result.csv", delimiter=","))
    (me, re_), (mg, rg) = runs["eager"], runs["graph"]
    assert me._opt_state[P1]["step"] == mg._opt_state[P1]["step"] == 6 and me._opt_state[P2]["step"] == mg._opt_state[P2]["step"] == 6
    assert re_[0, 9] == rg[0, 9] == 1e-3 and re_[1, 9] == rg[1, 9] == 5e-4
    # Two EAGER runs of this loop differ by up to 3.6e-3 in the reconstruction losses, 1.9e-2 in the InfoNCE terms of a batch of 4 and 1.7e-3 in the
    # weights after six Adam updates (order of the fp32 atomics x Adam's sign sensitivity: docs/rounds/r04.md item 1); graph vs eager measured the same
    for col in (0, 1, 3, 4, 5):
        assert np.allclose(re_[:, col], rg[:, col], rtol=1e-2, atol=1e-6), (col, re_[:, col], rg[:, col])
    for col in (2, 6, 7):
        assert np.allclose(re_[:, col], rg[:, col], rtol=0.08, atol=5e-3), (col, re_[:, col], rg[:, col])
    rel = float((me.arena.p - mg.arena.p).double().norm() / me.arena.p.double().norm())
    assert rel < 5e-3, rel


def test_pretrained_vit_checkpoint_forward_matches_oracle():
    """SURVEY 8(f) row 1 on the device: a timm-shaped checkpoint loaded the reference constructor's way
    (tests/test_oracle_golden.py pins that derivation bit-exactly to the reference) runs through the HIP path and agrees with
    the oracle on the same state - the bf16 shadows and transposed copies are rebuilt from the loaded masters."""
    import random
    from avsiam_amd.maskplan import make_contrastive_plan, make_mae_plan
    from avsiam_amd.weights import state_from_vit, synth_vit_checkpoint
    from oracle import ref_cpu
    cfg = AVSiamConfig(audio_tokens=128)
    B = 4
    ckpt = synth_vit_checkpoint(AVSiamConfig(), 4242)
    ckpt = {k: v for k, v in ckpt.items()}
    m = _model(cfg, 3, mode="init")
    a, v = synth_inputs(cfg, B, 8)
    gen = torch.Generator().manual_seed(2)
    pm, pc = make_mae_plan(cfg, B, gen), make_contrastive_plan(cfg, B, gen, random.Random(2))
    before = m(a.cuda(), v.cuda(), mae_loss_weight=1, contrast_loss_weight=0, mask_plan=pm)[0].item()
    m.load_vit_pretrained(ckpt, seed=3)
    st = state_from_vit(ckpt, cfg, seed=3)
    assert torch.equal(m._params["ast_base.blocks.4.mlp.fc1.weight"].cpu(), ckpt["blocks.4.mlp.fc1.weight"])
    torch.set_num_threads(16)
    P = {k: t.clone().requires_grad_(True) for k, t in st.items()}
    for mae, plan in ((True, pm), (False, pc)):
        out = m(a.cuda(), v.cuda(), mae_loss_weight=1 if mae else 0, contrast_loss_weight=0 if mae else 1, mask_plan=plan)
        ref = ref_cpu.forward(P, cfg, a, v, plan, mae_loss_weight=1 if mae else 0, contrast_loss_weight=0 if mae else 1)
        assert abs(out[0].item() - ref[0].item()) <= LOSS_RTOL * abs(ref[0].item()), (mae, out[0].item(), ref[0].item())
        if mae:
            assert abs(out[0].item() - before) > 1e-3              # the load changed the weights the kernels see


def test_external_torch_optimizer_without_mark_weights_changed():
    """The reference loop's recipe - optimizer.zero_grad(); loss.backward(); optimizer.step() with torch.optim.Adam on
    model.parameters() (:64-66,137-139) - must see its own updates at the next forward WITHOUT any extra call: the bf16 weight
    shadows are refreshed when a parameter's version counter moved."""
    from oracle import ref_cpu
    import random
    from avsiam_amd.maskplan import make_mae_plan
    cfg = AVSiamConfig(audio_tokens=128)
    B = 4
    m = _model(cfg, 9)
    opt = torch.optim.Adam([p for p in m.parameters() if p.requires_grad], 1e-3, weight_decay=5e-7, betas=(0.95, 0.999))
    a, v = synth_inputs(cfg, B, 12)
    plan = make_mae_plan(cfg, B, torch.Generator().manual_seed(6))
    out = m(a.cuda(), v.cuda(), mae_loss_weight=1, contrast_loss_weight=0, mask_plan=plan)
    l0 = out[0].item()
    opt.zero_grad()
    out[0].backward()
    opt.step()
    l1 = m(a.cuda(), v.cuda(), mae_loss_weight=1, contrast_loss_weight=0, mask_plan=plan)[0].item()
    assert abs(l1 - l0) > 1e-3, (l0, l1)                           # the step was seen by the kernels (the oracle check below says: correctly)
    torch.set_num_threads(16)
    P = {k: p.detach().cpu().clone() for k, p in m._params.items() if m.arena.info[k].live}
    with torch.no_grad():
        ref = ref_cpu.forward(P, cfg, a, v, plan, mae_loss_weight=1, contrast_loss_weight=0)
    assert abs(l1 - ref[0].item()) <= LOSS_RTOL * abs(ref[0].item()), (l1, ref[0].item())


def test_entry_point_main_runs_two_steps(tmp_path, capsys):
    """run_cavmae_pretrain_base.main(argv) with the reference's flags (synthetic data): trains, validates, writes the
    reference's artefacts.  --raw-input routes un-normalised fbank / uint8 frames through the device-side normalisation."""
    from avsiam_amd import run_cavmae_pretrain_base as entry
    exp = tmp_path / "exp"
    env_keep = {k: os.environ.pop(k, None) for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK")}
    try:
        model = entry.main(["--model", "cav-mae", "--dataset", "audioset", "--target_length", "256", "--batch-size", "4", "--lr", "2e-4",
                            "--n-epochs", "1", "--steps-per-epoch", "2", "--n-print-steps", "1", "--exp-dir", str(exp), "--save_model", "True",
                            "--data-val", "synthetic", "--val-steps", "1", "--mae_loss_weight", "3.0", "--contrast_loss_weight", "0.01",
                            "--raw-input", "--noise", "True", "--norm_pix_loss", "True", "--tr_pos", "False", "--masking_ratio", "0.75"])
    finally:
        for k, val in env_keep.items():
            if val is not None:
                os.environ[k] = val
    out = capsys.readouterr().out
    assert "Not using distributed mode" in out and "Epoch: [1][0/2]" in out and "Eval Total Loss" in out and "training diverged" not in out
    for f in ("args.json", "args.pkl", "result.csv", "progress.pkl", "models/audio_model.1.pth", "models/best_audio_model.pth",
              "models/best_optim_state.pth", "models/best_optim_state_2.pth"):
        assert (exp / f).exists(), f
    res = np.loadtxt(exp / "result.csv", delimiter=",").reshape(1, 10)
    assert np.isfinite(res).all() and res[0, 3] > 0 and res[0, 7] > 0
    assert model.cfg.audio_tokens == 128
    # --pretrain_path: the checkpoint this run wrote resumes WITH the first optimizer's state that sits beside it (moments and
    # step count: bias correction does not restart); a checkpoint of another model is refused instead of training from the random start
    from avsiam_amd.param_spec import P1
    env_keep = {k: os.environ.pop(k, None) for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK")}
    common = ["--model", "cav-mae", "--dataset", "audioset", "--target_length", "256", "--batch-size", "4", "--lr", "2e-4", "--n-epochs", "1",
              "--steps-per-epoch", "1", "--n-print-steps", "1", "--exp-dir", str(tmp_path / "exp2"), "--save_model", "False"]
    try:
        resumed = entry.main(common + ["--pretrain_path", str(exp / "models" / "best_audio_model.pth")])
        out = capsys.readouterr().out
        assert "restored the state of optimizer 1" in out and "missing keys: 0, unexpected keys: 0" in out
        assert resumed._opt_state[P1]["step"] == model._opt_state[P1]["step"] + 1 == 3          # 2 steps of the first run + 1
        assert resumed._opt_state[P2]["step"] == 3 and "optimizer 2 too" in out                 # (this loop saves the second optimizer beside the first)
        foreign = tmp_path / "foreign.pth"
        torch.save({"encoder.layer.0.weight": torch.zeros(3)}, foreign)
        with pytest.raises(SystemExit, match="not a checkpoint of this model"):
            entry.main(common + ["--pretrain_path", str(foreign)])
    finally:
        for k, val in env_keep.items():
            if val is not None:
                os.environ[k] = val


def test_entry_point_runs_vit_huge14_in_fp8_mode3(tmp_path, capsys):
    """BASELINE.json configs[4] through the reference's own entry point: `--model cav-mae-huge14 --fp8 3` builds
    models.CAVMAE_HUGE (the name /root/reference/src/models/__init__.py:13 exports; selection by args.model as src/run_cavmae_pretrain_base.py:171-175)
    with the fp8 MFMA path as a property of THAT model, trains two steps at depth 2 (the calibration step and one on delayed scales), validates,
    and writes the checkpoint with its '.fp8' delayed-scaling state - which a second invocation resumes from.  The same process then builds the
    default bf16 model through the same entry point: no process-wide precision switch is left behind."""
    from avsiam_amd import run_cavmae_pretrain_base as entry
    from avsiam_amd.models import CAVMAE_BASE, CAVMAE_HUGE
    exp = tmp_path / "exp_h14"
    env_keep = {k: os.environ.pop(k, None) for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "AVSIAM_FP8")}
    common = ["--dataset", "audioset", "--target_length", "1024", "--batch-size", "2", "--lr", "1e-4", "--n-epochs", "1", "--n-print-steps", "1",
              "--save_model", "True", "--mae_loss_weight", "3.0", "--contrast_loss_weight", "0.01"]
    try:
        model = entry.main(["--model", "cav-mae-huge14", "--fp8", "3", "--depth", "2", "--steps-per-epoch", "2", "--exp-dir", str(exp),
                            "--data-val", "synthetic", "--val-steps", "1"] + common)
        out = capsys.readouterr().out
        assert isinstance(model, CAVMAE_HUGE) and model.options.fp8 == "3"
        assert (model.cfg.embed_dim, model.cfg.head_dim, model.cfg.st, model.cfg.audio_tokens, model.cfg.video_tokens) == (1280, 80, 14, 657, 256)
        assert "Epoch: [1][1/2]" in out and "Eval Total Loss" in out and "training diverged" not in out
        res = np.loadtxt(exp / "result.csv", delimiter=",").reshape(1, 10)
        assert np.isfinite(res).all() and res[0, 3] > 0 and res[0, 7] > 0
        # every stack of both passes ran on fp8 operands (forward, input gradients, weight gradients) and is calibrated; nothing saturated
        stacks = [st for (which, _), eng in model._engines.items() for _, st in model._fp8_stacks(which, eng)]
        assert len(stacks) >= 4 and all(st.fp8_wgrad and len(st.f8_seen) == 4 * st.nblocks for st in stacks)
        assert model.fp8_saturation_events() <= 4
        assert (exp / "models" / "best_audio_model.pth.fp8").exists()
        resumed = entry.main(["--model", "cav-mae-huge14", "--fp8", "3", "--depth", "2", "--steps-per-epoch", "1", "--exp-dir", str(tmp_path / "exp_h14b"),
                              "--pretrain_path", str(exp / "models" / "best_audio_model.pth")] + common)
        out = capsys.readouterr().out
        assert "restored the fp8 delayed-scaling state" in out and "missing keys: 0, unexpected keys: 0" in out
        assert resumed.options.fp8 == "3"
        # the default model of the same process is the bf16 ViT-B (the precision travelled with the model object, not with the process)
        base = entry.main(["--model", "cav-mae", "--steps-per-epoch", "1", "--exp-dir", str(tmp_path / "exp_b"), "--target_length", "256",
                           "--dataset", "audioset", "--batch-size", "2", "--lr", "1e-4", "--n-epochs", "1", "--save_model", "False"])
        capsys.readouterr()
        assert type(base) is CAVMAE_BASE and base.options.fp8 == "0" and base.cfg.embed_dim == 768
        assert not any(getattr(st, "fp8", False) for eng in base._engines.values() for st in vars(eng).values() if hasattr(st, "nblocks"))
        with pytest.raises(SystemExit):                                    # argparse: a name outside the reference's export list
            entry.main(["--model", "cav-mae-giant"])
    finally:
        for k, val in env_keep.items():
            if val is not None:
                os.environ[k] = val


def test_torchrun_entry_forms_the_rccl_group_and_runs(tmp_path):
    """The way the reference is launched (torchrun, one process per GPU): utils.init_distributed_mode reads RANK / WORLD_SIZE /
    LOCAL_RANK, forms the "nccl" (= RCCL) process group - at world size 1 too, like the reference (utils.py:288) - and the
    entry point trains.  A second child process then runs the data-parallel code path itself on a one-rank RCCL group: the
    embedding all-gather and the chunked gradient all-reduce are ISSUED - through torch.distributed (comm.TorchDistComm(always=True))
    and through the C ABI's own RCCL communicator (comm.RcclComm) - and must leave losses and gradients unchanged."""
    env = dict(os.environ, PYTHONPATH=ROOT, MASTER_ADDR="127.0.0.1")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "1", "--master-addr", "127.0.0.1",
           "--master-port", "29577", "-m", "avsiam_amd.run_cavmae_pretrain_base", "--target_length", "256", "--batch-size", "4",
           "--n-epochs", "1", "--steps-per-epoch", "2", "--n-print-steps", "1", "--exp-dir", str(tmp_path / "e1")]
    r = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=600, cwd=ROOT)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]
    assert "| distributed init (rank 0): env://, gpu 0" in r.stdout and "Epoch: [1][1/2]" in r.stdout
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "1", "--master-addr", "127.0.0.1",
           "--master-port", "29579", os.path.join(ROOT, "tools", "rccl_selfcheck.py"), "--engine"]
    r = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=600, cwd=ROOT)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]
    assert "native communicator ok" in r.stdout, r.stdout[-1500:]          # avs_comm_* / avs_allreduce / avs_allgather / avs_reducescatter
    assert "engine path on rccl ok" in r.stdout, r.stdout[-1500:]


@pytest.mark.parametrize("env", [{}, {"AVSIAM_COMM": "rccl"}, {"AVSIAM_DP_DEFER": "1", "AVSIAM_DP_WIRE": "bf16"}])
def test_bench_force_dp_issues_every_collective_on_a_one_rank_rccl_group(env):
    """`bench.py --force-dp` under torchrun with ONE rank: the distributed branch of the bench and of the engine executes on the hardware -
    the "nccl" (= RCCL) group is formed, the packed embedding all-gather and the chunked, overlapped gradient all-reduce are issued
    (torch.distributed, or the C ABI's communicator; fp32 or bf16 wire; with the deferred MAE-only update), the barriers and the MAX
    over ranks run - and stdout carries exactly one line, the JSON (small shape: the full-size figures are in profiles/r03/dp_*.json)."""
    import json
    e = dict(os.environ, PYTHONPATH=ROOT, MASTER_ADDR="127.0.0.1", **env)
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "1", "--master-addr", "127.0.0.1",
           "--master-port", "29583", os.path.join(ROOT, "bench.py"), "--gpus", "1", "--steps", "2", "--warmup", "1", "--batch", "8", "--frames", "2",
           "--no-cpu-baseline", "--force-dp", "--roofline-steps", "0"]
    r = subprocess.run(cmd, env=e, capture_output=True, text=True, timeout=600, cwd=ROOT)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.strip()]
    assert len(lines) == 1, r.stdout[-1500:]
    d = json.loads(lines[0])
    f = d["config"]["force_dp"]
    assert d["n_gpus"] == 1 and d["value"] > 0 and f["allreduce_messages_last_backward"] >= 2, d["config"]
    assert f["comm"] == env.get("AVSIAM_COMM", "torch") and f["wire"] == env.get("AVSIAM_DP_WIRE", "fp32")
    assert all(v == v and abs(v) < 1e4 for v in d["final_losses"].values()), d["final_losses"]


def test_bench_gpus_2_launches_its_own_ranks():
    """`python bench.py --gpus 2` the way the driver's scaling run would start it - no launcher around it: the parent (no GPU call)
    starts two ranks as a child torch.distributed.run, the ranks run the WORLD_SIZE = 2 branches of the script and of the engine (rank
    environment, per-rank batches and seeds, the packed embedding all-gather, the chunked gradient all-reduce inside backward, barrier
    + max-over-ranks timing, the collectives' own report of the group size) and rank 0's ONE JSON line comes back through the
    parent.  A rehearsal: the box has one GPU, RCCL refuses two ranks on it, so AVSIAM_BENCH_SHARE_GPU=1 puts both ranks on device 0
    with gloo + host-staged collectives - everything but RCCL itself, whose calls the --force-dp test above issues at one rank."""
    import json
    e = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK")}
    e.update(PYTHONPATH=ROOT, AVSIAM_BENCH_SHARE_GPU="1")
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "2", "--warmup", "1", "--batch", "6", "--frames", "2",
           "--no-cpu-baseline", "--roofline-steps", "0"]
    r = subprocess.run(cmd, env=e, capture_output=True, text=True, timeout=900, cwd=ROOT)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-3000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.strip().startswith("{")]
    assert len(lines) == 1, r.stdout[-1500:]
    d = json.loads(lines[0])
    c = d["config"]
    assert d["n_gpus"] == 2 and c["global_batch"] == 12 and c["parallelism"] == "dp2" and d["scaling"] == "weak" and d["value"] > 0
    assert c["collectives"]["group_world_size"] == 2 and c["collectives"]["allreduce_messages_last_backward"] >= 2 and "rehearsal" in c
    assert all(v == v and abs(v) < 1e4 for v in d["final_losses"].values()), d["final_losses"]
    assert "starting 2 ranks as a child torch.distributed.run" in r.stderr


@pytest.mark.parametrize("noise", [False, True])
def test_raw_inputs_fused_into_the_input_reads(noise):
    """un-normalised fbank + uint8 frames handed to forward() with their transforms (input_xf): the patch
    gather and the loss-target gather apply the dataset arithmetic (/root/reference/src/dataloader.py:505-513, 461-462, 152-155)
    on the fly.  Must equal (a) the two-pass path - normalise on the device first (avsiam_amd.preprocess), then the plain
    forward - exactly (same arithmetic, same Philox stream), and (b) without the noise, the oracle fed with inputs normalised
    on the CPU by the dataloader's formulas."""
    import random
    from avsiam_amd import preprocess
    from avsiam_amd.maskplan import make_contrastive_plan, make_mae_plan
    from avsiam_amd.ops import InputXf
    from oracle import ref_cpu
    cfg = AVSiamConfig(audio_tokens=128, frames=2)
    B, mean, std = 3, -5.081, 4.4849
    g = torch.Generator().manual_seed(5)
    a_raw = torch.randn(B, cfg.audio_len, cfg.n_mels, generator=g) * std + mean
    v_u8 = torch.randint(0, 256, (B, cfg.frames, 3, cfg.img_size, cfg.img_size), generator=g, dtype=torch.uint8)
    gen = torch.Generator().manual_seed(2)
    pm, pc = make_mae_plan(cfg, B, gen), make_contrastive_plan(cfg, B, gen, random.Random(2))
    shift = amp = None
    if noise:
        shift = torch.tensor([5, -300, 0], dtype=torch.int32, device="cuda")
        amp = torch.tensor([0.05, 0.0, 0.09], dtype=torch.float32, device="cuda")
    xf = (InputXf.audio(mean, std, shift, amp, seed=77), InputXf.frames())
    a_dev = preprocess.normalize_fbank(a_raw.cuda(), mean, std, shift=shift, amp=amp, seed=77)
    v_dev = preprocess.normalize_frames(v_u8.cuda())
    m = _model(cfg, 41)
    for mae, plan in ((True, pm), (False, pc)):
        kw = dict(mae_loss_weight=1 if mae else 0, contrast_loss_weight=0 if mae else 1, mask_plan=plan)
        two = m(a_dev, v_dev, **kw)
        two[0].backward()
        g_two = m.arena.g.clone()
        one = m(a_raw.cuda(), v_u8.cuda(), input_xf=xf, **kw)
        one[0].backward()
        for i in (0, 1, 2, 3, 4):
            assert one[i].item() == two[i].item(), (mae, i, one[i].item(), two[i].item())
        lo, hi = m.arena.range[P2 if mae else P1]
        ga, gb = m.arena.g[lo:hi].double(), g_two[lo:hi].double()
        assert float(torch.dot(ga, gb) / (ga.norm() * gb.norm())) > 0.999999      # the same kernels on the same operands (fp32 atomics order)
        if not noise:
            torch.set_num_threads(16)
            an = (a_raw - mean) / std
            vn = (v_u8.float() / 255 - torch.tensor(preprocess.IMAGENET_DEFAULT_MEAN).view(1, 1, 3, 1, 1)) / torch.tensor(preprocess.IMAGENET_DEFAULT_STD).view(1, 1, 3, 1, 1)
            P = {k: t.clone() for k, t in synth_state(cfg, 41, "random", include_dead=False).items()}
            with torch.no_grad():
                ref = ref_cpu.forward(P, cfg, an, vn, plan, mae_loss_weight=1 if mae else 0, contrast_loss_weight=0 if mae else 1)
            assert abs(one[0].item() - ref[0].item()) <= LOSS_RTOL * abs(ref[0].item()), (mae, one[0].item(), ref[0].item())
    with pytest.raises(ValueError):
        m(a_raw.cuda(), v_u8.cuda().float(), input_xf=xf, mae_loss_weight=1, contrast_loss_weight=0, mask_plan=pm)


def test_library_rebuilds_from_source_on_this_box(tmp_path):
    """The shipped libavsiam_hip.so is built in the container; here the same sources are compiled afresh ON the GPU box
    (python -m avsiam_amd.build --out: a build of its own that leaves the product library alone) and the rebuilt library must
    give bitwise the results of the shipped one."""
    out = tmp_path / "libavsiam_rebuilt.so"
    env = dict(os.environ, PYTHONPATH=ROOT)
    env.pop("AVSIAM_HIPCC_EXTRA", None)
    r = subprocess.run([sys.executable, "-m", "avsiam_amd.build", "--out", str(out)], env=env, capture_output=True, text=True, timeout=1200, cwd=ROOT)
    assert r.returncode == 0 and out.exists(), r.stdout[-1500:] + r.stderr[-1500:]
    code = ("import torch, hashlib\n"
            "from avsiam_amd import ops\n"
            "torch.manual_seed(3)\n"
            "M, N, K = 1000, 768, 512\n"
            "A = torch.zeros(ops.pad_rows(M, 256), K, device='cuda', dtype=torch.bfloat16); A[:M] = torch.randn(M, K, device='cuda').bfloat16()\n"
            "W = (torch.randn(N, K, device='cuda') * 0.05).bfloat16(); b = torch.randn(N, device='cuda')\n"
            "o = torch.zeros(ops.pad_rows(M, 256), N, device='cuda', dtype=torch.bfloat16); o2 = torch.zeros_like(o)\n"
            "ops.gemm_nt(A, W, o, M, bias=b, out2=o2, act=1)\n"
            "x = torch.randn(M, N, device='cuda'); y = torch.zeros(M, N, device='cuda', dtype=torch.bfloat16)\n"
            "mean, rstd = torch.zeros(M, device='cuda'), torch.zeros(M, device='cuda')\n"
            "ops.layernorm_fwd(x, b, b, y, mean, rstd, M, 1e-5)\n"
            "torch.cuda.synchronize()\n"
            "h = hashlib.sha1()\n"
            "for t in (o[:M], o2[:M], y, mean, rstd): h.update(t.cpu().view(torch.uint8).numpy().tobytes())\n"
            "print('DIGEST', h.hexdigest())\n")
    digests = []
    for lib in (str(out), None):
        e = dict(env)
        if lib:
            e["AVSIAM_HIP_LIB"] = lib
        else:
            e.pop("AVSIAM_HIP_LIB", None)
        r = subprocess.run([sys.executable, "-c", code], env=e, capture_output=True, text=True, timeout=600, cwd=ROOT)
        assert r.returncode == 0, r.stdout[-1500:] + r.stderr[-1500:]
        digests.append([ln for ln in r.stdout.splitlines() if ln.startswith("DIGEST")][-1])
    assert digests[0] == digests[1], digests
