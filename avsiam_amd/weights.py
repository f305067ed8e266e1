"""Deterministic name-keyed weight synthesiser.

There is no network for ``timm.create_model(..., pretrained=True)`` (cav_mae_base.py:236) or the
``jx_vit_base_patch16_224_in21k`` file (:240), so weights are synthesised.  Every tensor is generated
from a Philox stream keyed by (seed, crc32(canonical name)); the same call reproduces the same
248 M parameters on the build container and on the GPU box without shipping them.

mode='init'   : reference-like initial state - LayerNorm 1/0, zero decoder pos/mask/modality tokens
                (:312-314,336-337), N(0, 0.02) weights, and the structural identities of the ctor:
                norm*_a/_v == norm* (:265-269), patch_embed_a = RGB-mean of patch_embed (:292-293),
                pos_embed_a = nearest interpolation of pos_embed[:,1:] (:298), ast_base == vit_base (:303),
                mm_layer_1/2 == vit_base.blocks[-1] (:306-307).
mode='random' : every tensor independent and non-trivial (used by parity tests so that selecting the
                wrong LayerNorm set / tower / token is visible).
"""
import os
import zlib

import numpy as np
import torch

from .config import AVSiamConfig
from .param_spec import alias_of, build_spec, build_spec_ft


def _stream(seed: int, name: str):
    key = zlib.crc32(alias_of(name).encode())
    return np.random.Generator(np.random.Philox(key=[seed & 0xFFFFFFFF, key]))


def _random_tensor(info, seed):
    g = _stream(seed, info.name)
    n = int(np.prod(info.shape))
    x = g.standard_normal(n, dtype=np.float32)
    if info.kind == "ln_w":
        x = 1.0 + 0.1 * x
    elif info.kind == "ln_b":
        x = 0.05 * x
    elif info.kind in ("pos", "token"):
        x = 0.02 * x
    elif info.kind == "bias":
        x = 0.02 * x
    else:  # linear_w / conv_w
        x = 0.02 * x
    return torch.from_numpy(x.reshape(info.shape))


def synth_state(cfg: AVSiamConfig, seed: int = 0, mode: str = "init", include_dead: bool = True, spec=None):
    """Returns {name: fp32 tensor} for the unique tensors of the schema (`spec`: default CAVMAE_BASE's)."""
    spec = build_spec(cfg) if spec is None else spec
    out, todo = {}, []
    for info in spec:
        if not include_dead and info.live == 0:
            continue
        if mode == "random":
            out[info.name] = None
            todo.append(info)
            continue
        if info.kind == "ln_w":
            out[info.name] = torch.ones(info.shape)
        elif info.kind == "ln_b":
            out[info.name] = torch.zeros(info.shape)
        elif info.zero_init:
            out[info.name] = torch.zeros(info.shape)
        else:
            out[info.name] = None
            todo.append(info)
    # every tensor has its own Philox stream (keyed by seed and name), so the draws are independent of the order they are made in: the big
    # matrices are drawn side by side (numpy releases the GIL while it fills an array) - 5 s -> ~1 s for ViT-B's 212 M values on 8 cores
    if len(todo) > 8:
        from concurrent.futures import ThreadPoolExecutor
        with ThreadPoolExecutor(max_workers=max(1, min(16, os.cpu_count() or 1))) as pool:
            for info, t in zip(todo, pool.map(lambda i: _random_tensor(i, seed), todo)):
                out[info.name] = t
    else:
        for info in todo:
            out[info.name] = _random_tensor(info, seed)
    if mode == "init":
        _tie_init(out, cfg)
    return out


def synth_state_ft(cfg: AVSiamConfig, label_dim: int, seed: int = 0, mode: str = "init"):
    """State of ``CAVMAEFT_BASE`` (553 keys / 313 unique tensors at ViT-B).  Tensors are keyed by name, so the shared ViT
    equals ``synth_state``'s for the same seed; mode 'init' also applies the constructor's identities, where the fusion
    blocks are copies of blocks 10 AND 11 (cav_mae_base.py:821-822), not twice the last one as in the pre-training model."""
    st = synth_state(cfg, seed, mode, spec=build_spec_ft(cfg, label_dim))
    if mode == "init":
        pre = f"vit_base.blocks.{cfg.depth - 2}."
        for k in list(st.keys()):
            if k.startswith(pre):
                st["mm_layer_1." + k[len(pre):]] = st[k].clone()
    return st


def _tie_init(st, cfg):
    """Structural identities of the reference constructor (see module docstring)."""
    v = "vit_base."
    if v + "patch_embed_a.proj.weight" in st:
        st[v + "patch_embed_a.proj.weight"] = st[v + "patch_embed.proj.weight"].mean(dim=1, keepdim=True).clone()
        st[v + "patch_embed_a.proj.bias"] = st[v + "patch_embed.proj.bias"].clone()
        pe = st[v + "pos_embed"][:, 1:].permute(0, 2, 1)
        st[v + "pos_embed_a"] = torch.nn.functional.interpolate(pe, size=[cfg.audio_tokens]).permute(0, 2, 1).contiguous()
    for k in list(st.keys()):
        if k.startswith(v):
            a = "ast_base." + k[len(v):]
            if a in st:
                st[a] = st[k].clone()
    last = f"vit_base.blocks.{cfg.depth - 1}."
    for k in list(st.keys()):
        if k.startswith(last):
            for mm in ("mm_layer_1.", "mm_layer_2."):
                if mm + k[len(last):] in st:
                    st[mm + k[len(last):]] = st[k].clone()
    for k in ("my_patch_embed.proj.weight", "my_patch_embed.proj.bias"):
        if k in st:
            st[k] = st["vit_base." + k[len("my_"):]].clone()
    for k in ("my_patch_embed_a.proj.weight", "my_patch_embed_a.proj.bias"):
        if k in st:
            st[k] = st["vit_base." + k[len("my_"):]].clone()


def synth_vit_checkpoint(cfg: AVSiamConfig, seed: int = 0):
    """A timm-shaped ViT state dict (the keys of ``jx_vit_base_patch16_224_in21k``: cls_token, pos_embed,
    patch_embed.proj.*, blocks.N.{norm1,attn.qkv,attn.proj,norm2,mlp.fc1,mlp.fc2}.*, norm.*, head.*) with name-keyed
    synthetic values - a stand-in for the checkpoint the reference constructor loads (cav_mae_base.py:236-240) on machines
    without network; also what the pretrained-init golden vector was generated from (oracle/gen_golden.py)."""
    out = {}
    for k, t in synth_state(cfg, seed, "random").items():
        if not k.startswith("vit_base."):
            continue
        kk = k[len("vit_base."):]
        if "_a." in kk or "_v." in kk or kk.endswith("_a"):         # derived by the constructor, not part of the checkpoint
            continue
        out[kk] = t.clone()
    return out


def state_from_vit(vit_sd, cfg: AVSiamConfig, seed: int = 0):
    """Initial state of CAVMAE_BASE from a timm ViT checkpoint (``jx_vit_base_patch16_224_in21k``: keys ``cls_token``,
    ``pos_embed``, ``patch_embed.proj.*``, ``blocks.N.*``, ``norm.*``, ``head.*``), the way the reference constructor
    derives it (/root/reference/src/models/cav_mae_base.py:236-307): the checkpoint goes into ``vit_base`` (non-strict:
    :240), every block's ``norm1/norm2`` is copied to ``norm1_a/_v`` / ``norm2_a/_v`` (:262-267), ``norm`` to ``norm_a``
    (:299), the audio patch embedding is the RGB mean of the visual kernel (:291-294), ``pos_embed_a`` the nearest
    interpolation of the visual table to the audio token count (:298), ``ast_base`` a deep copy of ``vit_base`` (:303),
    ``mm_layer_1/2`` copies of the last block (:306-307).  Everything the checkpoint does not cover (decoder, mask token,
    modality embeddings) keeps the synthetic initial state of ``synth_state(cfg, seed)``.
    Returns {name: fp32 tensor} over the unique tensors of the schema; unknown / mis-shaped checkpoint keys raise."""
    st = synth_state(cfg, seed, mode="init")
    for k, val in vit_sd.items():
        name = "vit_base." + k
        if name not in st:
            if k.startswith(("pre_logits", "fc_norm", "head_dist")):     # timm variants carry these; the reference drops them (strict=False)
                continue
            raise KeyError(f"checkpoint key '{k}' has no counterpart in the CAVMAE_BASE schema")
        if k == "patch_embed.proj.weight" and cfg.st != cfg.patch and tuple(val.shape[-2:]) == (cfg.st, cfg.st):
            # a stride x stride kernel (timm's patch14) into patch x patch storage (config.stride): zero-padded; the padded positions are
            # never read (the im2col columns there are zero) and never receive a gradient
            val = torch.nn.functional.pad(val, (0, cfg.patch - cfg.st, 0, cfg.patch - cfg.st))
        if tuple(val.shape) != tuple(st[name].shape):
            raise ValueError(f"checkpoint key '{k}': shape {tuple(val.shape)} != {tuple(st[name].shape)}")
        st[name] = val.detach().to(torch.float32).clone()
    for i in range(cfg.depth):
        for n in ("norm1", "norm2"):
            for suf in ("_a", "_v"):
                for wb in ("weight", "bias"):
                    st[f"vit_base.blocks.{i}.{n}{suf}.{wb}"] = st[f"vit_base.blocks.{i}.{n}.{wb}"].clone()
    for wb in ("weight", "bias"):
        st[f"vit_base.norm_a.{wb}"] = st[f"vit_base.norm.{wb}"].clone()
    _tie_init(st, cfg)
    return st


def synth_inputs(cfg: AVSiamConfig, batch: int, seed: int = 87, constant: float = None):
    """AudioSet-shaped synthetic pair: a ~ N(0,1) [B, target_length, 128] (fbank after normalisation,
    /root/reference/src/dataloader.py:505-506), v ~ N(0,1) [B,(T,)3,224,224] (:152-155).
    ``constant`` reproduces the loader's degenerate fallback tensors (:385,424)."""
    g = np.random.Generator(np.random.Philox(key=[seed & 0xFFFFFFFF, 0xA5A5]))
    a_shape = (batch, cfg.audio_len, cfg.n_mels)
    v_shape = (batch, cfg.in_chans, cfg.img_size, cfg.img_size) if cfg.frames == 1 else \
        (batch, cfg.frames, cfg.in_chans, cfg.img_size, cfg.img_size)
    if constant is not None:
        return torch.full(a_shape, constant), torch.full(v_shape, constant)
    a = torch.from_numpy(g.standard_normal(int(np.prod(a_shape)), dtype=np.float32).reshape(a_shape))
    v = torch.from_numpy(g.standard_normal(int(np.prod(v_shape)), dtype=np.float32).reshape(v_shape))
    return a, v
