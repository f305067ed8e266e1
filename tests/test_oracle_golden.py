"""The oracle (oracle/ref_cpu.py) against golden vectors produced by the unmodified reference
(oracle/gen_golden.py, run in the build container).  fp32 vs fp32, identical mask plan:
losses rel 1e-5, gradients rel 1e-4 (summation order only) - SURVEY.md section 8(c)."""
import numpy as np
import pytest
import torch

from avsiam_amd.config import AVSiamConfig
from avsiam_amd.weights import synth_inputs, synth_state
from oracle import ref_cpu
from tests.helpers import check_grads_against_golden, golden_plan, load_golden, sample_positions


def _run(name):
    d = load_golden(name)
    cfg = AVSiamConfig()
    B = int(d["batch"])
    const = float(d["constant"])
    a, v = synth_inputs(cfg, B, int(d["input_seed"]), None if np.isnan(const) else const)
    P = {k: t.clone().requires_grad_(True) for k, t in synth_state(cfg, int(d["weight_seed"]), "random").items()}
    plan = golden_plan(d)
    extras = {}
    mae = str(d["which"]) == "mae"
    out = ref_cpu.forward(P, cfg, a, v, plan, mae_loss_weight=1 if mae else 0,
                          contrast_loss_weight=0 if mae else 1, extras=extras)
    out[0].backward()
    return d, out, extras, {k: p.grad for k, p in P.items()}


@pytest.mark.parametrize("name", ["c_w1_b4", "m_w1_b4", "m_w1_b2_const", "c_w1_b5", "c_w1_b10"])
def test_oracle_matches_reference(name):
    torch.set_num_threads(8)
    d, out, extras, grads = _run(name)
    got = np.array([out[i].item() for i in (0, 1, 2, 3, 4, 7)])
    np.testing.assert_allclose(got, d["out_scalars"], rtol=1e-5, atol=1e-6)
    if str(d["which"]) == "mae":
        np.testing.assert_array_equal(out[5].numpy(), d["mask_a"])
        np.testing.assert_array_equal(out[6].numpy(), d["mask_v"])
        for k in ("pred_a", "pred_v"):
            p = extras[k].detach().double().reshape(-1)
            assert abs(p.norm().item() - d[k + "_l2"]) <= 1e-5 * d[k + "_l2"]
            s = np.array([p[i].item() for i in sample_positions(k, p.numel(), 64)])
            np.testing.assert_allclose(s, d[k + "_samples"], rtol=1e-4, atol=1e-5)
    else:
        assert out[5] is None and out[6] is None
        np.testing.assert_allclose(extras["logits"].detach().numpy(), d["logits"], rtol=1e-4, atol=2e-4)
    check_grads_against_golden(d, grads, rel_l2=1e-4)


def test_oracle_matches_reference_combined_forward():
    """The call validate() makes (traintest_cavmae_base.py:401): both branches in one forward at the run's loss weights
    (mae 3.0 / contrastive 0.01).  The MAE branch only switches on its weight (:694,739); the contrastive loss is scaled (:735)."""
    torch.set_num_threads(8)
    d = load_golden("mc_w1_b4")
    cfg = AVSiamConfig()
    B = int(d["batch"])
    a, v = synth_inputs(cfg, B, int(d["input_seed"]))
    P = {k: t.clone().requires_grad_(True) for k, t in synth_state(cfg, int(d["weight_seed"]), "random").items()}
    plan = golden_plan(d)
    assert set(plan) == {"mae", "contrastive"}
    wm, wc = (float(x) for x in d["loss_weights"])
    extras = {}
    out = ref_cpu.forward(P, cfg, a, v, plan, mae_loss_weight=wm, contrast_loss_weight=wc, extras=extras)
    out[0].backward()
    got = np.array([out[i].item() for i in (0, 1, 2, 3, 4, 7)])
    np.testing.assert_allclose(got, d["out_scalars"], rtol=1e-5, atol=1e-6)
    assert out[5] is None and out[6] is None                                   # :722 - the mixed encoder's None masks win
    np.testing.assert_allclose(extras["logits"].detach().numpy(), d["logits"], rtol=1e-4, atol=2e-4)
    for k in ("pred_a", "pred_v"):
        p = extras[k].detach().double().reshape(-1)
        assert abs(p.norm().item() - d[k + "_l2"]) <= 1e-5 * d[k + "_l2"]
    check_grads_against_golden(d, {k: p.grad for k, p in P.items()}, rel_l2=1e-4)


def test_pretrained_init_matches_reference_constructor(golden_dir):
    """SURVEY 8(f) row 1, pinned to the reference: oracle/gen_golden.py ran the reference CONSTRUCTOR
    (/root/reference/src/models/cav_mae_base.py:236-307) with its checkpoint load answered by
    weights.synth_vit_checkpoint(cfg, seed) and stored a CRC-32 of every tensor it derived.  weights.state_from_vit on the
    same checkpoint must reproduce all of them BIT-exactly (copies, the RGB-mean audio kernel, the nearest-interpolated
    position table, the deep-copied towers and joint layers, the zero decoder tokens)."""
    import json
    import os
    import zlib
    from avsiam_amd.param_spec import alias_of
    from avsiam_amd.weights import state_from_vit, synth_vit_checkpoint
    with open(os.path.join(golden_dir, "pretrained_init.json")) as f:
        g = json.load(f)
    cfg = AVSiamConfig()
    ckpt = synth_vit_checkpoint(cfg, g["vit_seed"])
    assert len(ckpt) == g["n_checkpoint_keys"] == 152
    st = state_from_vit(ckpt, cfg, seed=0)
    assert len(g["tensors"]) >= 790
    bad = []
    for k, rec in g["tensors"].items():
        t = st[alias_of(k)].contiguous()
        assert list(t.shape) == rec["shape"], k
        if zlib.crc32(t.numpy().tobytes()) != rec["crc32"]:
            bad.append((k, float(t.double().sum()), rec["sum"]))
    assert not bad, bad[:5]


def test_schema_matches_reference(golden_dir):
    import json
    import os
    from avsiam_amd.param_spec import build_spec, state_dict_keys, alias_of
    with open(os.path.join(golden_dir, "schema.json")) as f:
        sch = json.load(f)
    cfg = AVSiamConfig()
    keys = state_dict_keys(cfg)
    assert len(keys) == sch["n_keys"] == 963
    assert sorted(keys) == sorted(sch["shapes"].keys())
    spec = {s.name: s for s in build_spec(cfg)}
    assert len(spec) == 723
    assert sum(int(np.prod(s.shape)) for s in spec.values()) == sch["n_params"] == 248035494
    for k in keys:
        assert list(spec[alias_of(k)].shape) == sch["shapes"][k], k
    # LayerNorm eps: 1e-5 everywhere except vit_base.norm / norm_a (and their ast_base copies) = 1e-6
    for n, e in sch["ln_eps"].items():
        want = 1e-6 if n in ("vit_base.norm", "vit_base.norm_a", "ast_base.norm", "ast_base.norm_a") else 1e-5
        assert e == want, (n, e)
    # live sets: 86,426,880 (contrastive pass) and 212,123,392 (MAE pass)
    assert sum(int(np.prod(s.shape)) for s in spec.values() if s.live & 1) == 86426880
    assert sum(int(np.prod(s.shape)) for s in spec.values() if s.live & 2) == 212123392


def test_pretrained_vit_init_reproduces_constructor_identities():
    """a timm ViT-B/16 checkpoint loaded the way the reference constructor does it
    (/root/reference/src/models/cav_mae_base.py:236-307) - checked on a random timm-shaped state dict."""
    import torch.nn.functional as F
    from avsiam_amd.config import AVSiamConfig
    from avsiam_amd.models import CAVMAE_BASE
    cfg = AVSiamConfig()
    g = torch.Generator().manual_seed(3)
    D, depth = cfg.embed_dim, cfg.depth
    vit = {"cls_token": torch.randn(1, 1, D, generator=g), "pos_embed": torch.randn(1, 197, D, generator=g),
           "patch_embed.proj.weight": torch.randn(D, 3, 16, 16, generator=g), "patch_embed.proj.bias": torch.randn(D, generator=g),
           "norm.weight": torch.randn(D, generator=g), "norm.bias": torch.randn(D, generator=g),
           "head.weight": torch.randn(21843, D, generator=g), "head.bias": torch.randn(21843, generator=g)}
    for i in range(depth):
        b = f"blocks.{i}."
        for n, shp in (("norm1.weight", (D,)), ("norm1.bias", (D,)), ("attn.qkv.weight", (3 * D, D)), ("attn.qkv.bias", (3 * D,)),
                       ("attn.proj.weight", (D, D)), ("attn.proj.bias", (D,)), ("norm2.weight", (D,)), ("norm2.bias", (D,)),
                       ("mlp.fc1.weight", (4 * D, D)), ("mlp.fc1.bias", (4 * D,)), ("mlp.fc2.weight", (D, 4 * D)), ("mlp.fc2.bias", (D,))):
            vit[b + n] = torch.randn(*shp, generator=g)
    m = CAVMAE_BASE(cfg=cfg, verbose=False)
    m.load_vit_pretrained(vit)
    sd = m.state_dict()
    assert len(sd) == 963
    for k, v in vit.items():                                        # the checkpoint itself, under both block names
        assert torch.equal(sd["vit_base." + k], v), k
        if k.startswith("blocks."):
            assert torch.equal(sd["my_" + k], v), k                 # my_blocks aliases vit_base.blocks (:248,278)
    for i in (0, depth - 1):
        for n in ("norm1", "norm2"):
            for suf in ("_a", "_v"):
                assert torch.equal(sd[f"vit_base.blocks.{i}.{n}{suf}.weight"], vit[f"blocks.{i}.{n}.weight"])      # :262-267
    assert torch.equal(sd["vit_base.norm_a.bias"], vit["norm.bias"])                                               # :299
    assert torch.equal(sd["vit_base.patch_embed_a.proj.weight"], vit["patch_embed.proj.weight"].mean(dim=1, keepdim=True))   # :291-294
    want_pos = F.interpolate(vit["pos_embed"][:, 1:].permute(0, 2, 1), size=[cfg.audio_tokens]).permute(0, 2, 1)
    assert torch.equal(sd["vit_base.pos_embed_a"], want_pos)                                                        # :298
    for k in sd:
        if k.startswith("ast_base."):                                                                               # deepcopy (:303)
            assert torch.equal(sd[k], sd["vit_base." + k[len("ast_base."):]]), k
        if k.startswith("mm_layer_1."):                                                                             # :306-307
            assert torch.equal(sd[k], sd[f"vit_base.blocks.{depth - 1}." + k[len("mm_layer_1."):]]), k
    with pytest.raises(ValueError):
        m.load_vit_pretrained({"pos_embed": torch.zeros(1, 577, D)})


def test_patch14_checkpoint_kernel_is_zero_padded_into_the_16x16_storage():
    """config.stride: a timm patch14 kernel [D, 3, 14, 14] loads into the [D, 3, 16, 16] storage zero-padded; the audio kernel derived from
    it (RGB mean) is padded the same way, and the oracle's convolution (14 x 14 corner, stride 14) sees exactly the checkpoint's kernel."""
    from avsiam_amd.config import vit_huge14
    from avsiam_amd.weights import state_from_vit, synth_vit_checkpoint
    from avsiam_amd.config import AVSiamConfig
    import dataclasses
    from oracle import ref_cpu
    cfg = vit_huge14(depth=1)
    sd = synth_vit_checkpoint(dataclasses.replace(cfg, stride=0), 11)             # a timm-shaped checkpoint at this width (16 x 16 kernel, 197 positions)
    w14 = torch.randn(cfg.embed_dim, 3, 14, 14)
    sd["patch_embed.proj.weight"] = w14
    sd["pos_embed"] = torch.randn(1, cfg.video_tokens + 1, cfg.embed_dim)
    st = state_from_vit(sd, cfg)
    w = st["vit_base.patch_embed.proj.weight"]
    assert tuple(w.shape) == (cfg.embed_dim, 3, 16, 16) and torch.equal(w[..., :14, :14], w14)
    assert float(w[..., 14:, :].abs().max()) == 0 and float(w[..., :, 14:].abs().max()) == 0
    wa = st["vit_base.patch_embed_a.proj.weight"]
    assert torch.equal(wa[..., :14, :14], w14.mean(dim=1, keepdim=True)) and float(wa[..., 14:, :].abs().max()) == 0
    img = torch.randn(2, 3, 224, 224)
    got = ref_cpu.patch_embed(img, w, None, cfg.stride)
    want = torch.nn.functional.conv2d(img, w14, None, stride=14).flatten(2).transpose(1, 2)
    assert got.shape == (2, 256, cfg.embed_dim) and torch.allclose(got, want, atol=1e-5)


def test_golden_gradient_checker_discriminates():
    """tests.helpers.gpu_grads_vs_golden is what the GPU suite holds the HIP gradients against the reference's per-tensor statistics with
    (L2 norm, 8 sampled elements, element sum).  Each of its three tolerances must be able to FAIL (a bound nothing can
    exceed is not a check): on a synthetic golden record, with the tolerances the GPU tests use,
      - the exact tensor and a tensor with bf16-sized independent noise pass;
      - a zero tensor, a 2 % scaled tensor and a tensor with 10 % noise fail (norm / samples);
      - a tensor whose elements all carry an offset of 2e-3 of an element's rms - invisible to the norm and the samples - fails on the sum."""
    import json
    import zlib
    from tests.helpers import gpu_grads_vs_golden, sample_positions
    from tests.test_parity_gpu import GOLD_L2, GOLD_L2_C, GOLD_SAMP, GOLD_SAMP_C, GOLD_SUM, GOLD_SUM_C
    gen = torch.Generator().manual_seed(0)
    name = "mm_layer_2.mlp.fc2.weight"
    ref = torch.randn(768 * 3072, generator=gen, dtype=torch.float64) * 1e-3
    d = {"grad_names": np.array(json.dumps([name])), "grad_none": np.array(json.dumps([])), "grad_sum": np.array([ref.sum().item()]),
         "grad_l2": np.array([ref.norm().item()]), "grad_samples": np.array([[ref[i].item() for i in sample_positions(name, ref.numel())]])}
    rms = ref.norm().item() / ref.numel() ** 0.5
    noise = torch.randn(ref.numel(), generator=gen, dtype=torch.float64)
    for l2, samp, sm in ((GOLD_L2, GOLD_SAMP, GOLD_SUM), (GOLD_L2_C, GOLD_SAMP_C, GOLD_SUM_C)):
        check = lambda g: gpu_grads_vs_golden(d, lambda n: g, "selftest", l2, samp, sm)
        check(ref.clone())
        check(ref + 2.0 ** -9 * rms * noise)                        # independent errors of bf16 size: inside every bound
        for bad in (torch.zeros_like(ref), ref * 1.02, ref + 0.1 * rms * noise * 5, ref + 2e-3 * rms):
            with pytest.raises(AssertionError):
                check(bad)
    # the offset case is rejected by the SUM alone: norm and samples cannot see it
    off = ref + 2e-3 * rms
    with pytest.raises(AssertionError) as e:
        gpu_grads_vs_golden(d, lambda n: off, "selftest", GOLD_L2, GOLD_SAMP, GOLD_SUM)
    assert "sum" in str(e.value)
