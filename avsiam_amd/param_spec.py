"""Parameter schema of ``CAVMAE_BASE`` (963 state-dict keys / 723 tensors at ViT-B).

Restates what ``CAVMAE_BASE.__init__`` builds (/root/reference/src/models/cav_mae_base.py:216-337):
``vit_base`` (timm ViT skeleton + per-modality LayerNorm copies :264-269, audio patch embed :291-297,
``pos_embed_a`` :298, ``norm_a`` :299), ``ast_base = deepcopy(vit_base)`` :303, ``mm_layer_1/2`` :306-307,
the decoder :311-337, the ``my_patch_embed*`` modules :285-286 and the ``my_blocks`` alias of
``vit_base.blocks`` :248,278.

Each entry carries its liveness in the two passes of the training step
(/root/reference/src/traintest_cavmae_base.py:131-152): pass 1 = contrastive only,
pass 2 = MAE only.  The flat arena (arena.py) orders tensors [pass-1 only | both | pass-2 only | dead]
so each pass's live gradient set is one contiguous range (one RCCL all-reduce, one Adam launch).
"""
from collections import OrderedDict
from dataclasses import dataclass
from typing import Tuple

from .config import AVSiamConfig

P1, P2 = 1, 2          # liveness bits


@dataclass(frozen=True)
class ParamInfo:
    name: str
    shape: Tuple[int, ...]
    kind: str          # 'linear_w' | 'bias' | 'ln_w' | 'ln_b' | 'conv_w' | 'pos' | 'token'
    live: int          # bitmask of P1 / P2
    zero_init: bool = False   # zero at reference init (decoder pos/mask/modality tokens :312-314,336-337)


def _block(prefix, dim, hidden, live_plain, live_a, live_v, live_core):
    out = []
    for n, lv in (("norm1", live_plain), ("norm1_a", live_a), ("norm1_v", live_v)):
        out.append(ParamInfo(f"{prefix}.{n}.weight", (dim,), "ln_w", lv))
        out.append(ParamInfo(f"{prefix}.{n}.bias", (dim,), "ln_b", lv))
    out.append(ParamInfo(f"{prefix}.attn.qkv.weight", (3 * dim, dim), "linear_w", live_core))
    out.append(ParamInfo(f"{prefix}.attn.qkv.bias", (3 * dim,), "bias", live_core))
    out.append(ParamInfo(f"{prefix}.attn.proj.weight", (dim, dim), "linear_w", live_core))
    out.append(ParamInfo(f"{prefix}.attn.proj.bias", (dim,), "bias", live_core))
    for n, lv in (("norm2", live_plain), ("norm2_a", live_a), ("norm2_v", live_v)):
        out.append(ParamInfo(f"{prefix}.{n}.weight", (dim,), "ln_w", lv))
        out.append(ParamInfo(f"{prefix}.{n}.bias", (dim,), "ln_b", lv))
    out.append(ParamInfo(f"{prefix}.mlp.fc1.weight", (hidden, dim), "linear_w", live_core))
    out.append(ParamInfo(f"{prefix}.mlp.fc1.bias", (hidden,), "bias", live_core))
    out.append(ParamInfo(f"{prefix}.mlp.fc2.weight", (dim, hidden), "linear_w", live_core))
    out.append(ParamInfo(f"{prefix}.mlp.fc2.bias", (dim,), "bias", live_core))
    return out


def _tower(prefix, cfg: AVSiamConfig, role):
    """role 'vit': the Siamese tower (pass 1 with _a/_v norms; pass 2 video with _v norms,
    cav_mae_base.py:487,557-558).  role 'ast': the deep-copied audio tower of pass 2, plain norms (:489)."""
    D, p = cfg.embed_dim, cfg.patch
    vit = role == "vit"
    emb = (P1 | P2) if vit else 0       # both passes embed through vit_base (:448,453,515,520)
    out = [
        ParamInfo(f"{prefix}.cls_token", (1, 1, D), "token", 0),
        ParamInfo(f"{prefix}.pos_embed", (1, cfg.video_tokens + 1, D), "pos", emb),
        ParamInfo(f"{prefix}.pos_embed_a", (1, cfg.audio_tokens, D), "pos", emb),
        ParamInfo(f"{prefix}.patch_embed.proj.weight", (D, cfg.in_chans, p, p), "conv_w", emb),
        ParamInfo(f"{prefix}.patch_embed.proj.bias", (D,), "bias", emb),
    ]
    for i in range(cfg.depth):
        if vit:
            out += _block(f"{prefix}.blocks.{i}", D, D * cfg.mlp_ratio, 0, P1, P1 | P2, P1 | P2)
        else:
            out += _block(f"{prefix}.blocks.{i}", D, D * cfg.mlp_ratio, P2, 0, 0, P2)
    out += [
        ParamInfo(f"{prefix}.norm.weight", (D,), "ln_w", (P1 | P2) if vit else 0),      # :492,563
        ParamInfo(f"{prefix}.norm.bias", (D,), "ln_b", (P1 | P2) if vit else 0),
        ParamInfo(f"{prefix}.head.weight", (cfg.n_classes, D), "linear_w", 0),
        ParamInfo(f"{prefix}.head.bias", (cfg.n_classes,), "bias", 0),
        ParamInfo(f"{prefix}.patch_embed_a.proj.weight", (D, 1, p, p), "conv_w", emb),
        ParamInfo(f"{prefix}.patch_embed_a.proj.bias", (D,), "bias", emb),
        ParamInfo(f"{prefix}.norm_a.weight", (D,), "ln_w", P1 if vit else P2),          # :566 / :495
        ParamInfo(f"{prefix}.norm_a.bias", (D,), "ln_b", P1 if vit else P2),
    ]
    return out


def build_spec(cfg: AVSiamConfig):
    """Unique tensors in the registration order of the reference module (723 at ViT-B)."""
    D, Dd, p = cfg.embed_dim, cfg.dec_dim, cfg.patch
    spec = []
    spec += _tower("vit_base", cfg, "vit")
    spec += [
        ParamInfo("my_patch_embed.proj.weight", (D, cfg.in_chans, p, p), "conv_w", 0),
        ParamInfo("my_patch_embed.proj.bias", (D,), "bias", 0),
        ParamInfo("my_patch_embed_a.proj.weight", (D, 1, p, p), "conv_w", 0),
        ParamInfo("my_patch_embed_a.proj.bias", (D,), "bias", 0),
    ]
    spec += _tower("ast_base", cfg, "ast")
    for n in ("mm_layer_1", "mm_layer_2"):                                   # applied with 'a' norms :699-700
        spec += _block(n, D, D * cfg.mlp_ratio, 0, P2, 0, P2)
    spec += [
        ParamInfo("decoder_embed.weight", (Dd, D), "linear_w", P2),
        ParamInfo("decoder_embed.bias", (Dd,), "bias", P2),
        ParamInfo("decoder_pos_embed_a", (1, cfg.audio_tokens, Dd), "pos", P2, True),
        ParamInfo("decoder_pos_embed_v", (1, cfg.video_tokens, Dd), "pos", P2, True),
        ParamInfo("mask_token", (1, 1, Dd), "token", P2, True),
    ]
    for i in range(cfg.dec_depth):                                           # plain norms :630
        spec += _block(f"decoder_blocks.{i}", Dd, Dd * cfg.mlp_ratio, P2, 0, 0, P2)
    spec += [
        ParamInfo("decoder_norm.weight", (Dd,), "ln_w", P2),
        ParamInfo("decoder_norm.bias", (Dd,), "ln_b", P2),
        ParamInfo("decoder_pred_a.weight", (p * p, Dd), "linear_w", P2),
        ParamInfo("decoder_pred_a.bias", (p * p,), "bias", P2),
        ParamInfo("decoder_pred_v.weight", (p * p * cfg.in_chans, Dd), "linear_w", P2),
        ParamInfo("decoder_pred_v.bias", (p * p * cfg.in_chans,), "bias", P2),
        ParamInfo("decoder_modality_a", (1, 1, Dd), "token", P2, True),
        ParamInfo("decoder_modality_v", (1, 1, Dd), "token", P2, True),
    ]
    names = [s.name for s in spec]
    assert len(set(names)) == len(names)
    return spec


def build_spec_ft(cfg: AVSiamConfig, label_dim: int):
    """Unique tensors of ``CAVMAEFT_BASE`` (/root/reference/src/models/cav_mae_base.py:744-825): the Siamese ViT with its
    per-modality norms and audio embedding (:747-804), four classification heads LayerNorm+Linear (:809-815) and the two
    fusion blocks, copies of blocks 10 / 11 (:821-822).  There is one pass (inference), so every tensor the forward modes
    (:827-1035) read is tagged P1 and the rest 0."""
    D, p = cfg.embed_dim, cfg.patch
    spec = []
    for s in _tower("vit_base", cfg, "vit"):
        spec.append(ParamInfo(s.name, s.shape, s.kind, P1 if s.live else 0))
    spec += [
        ParamInfo("my_patch_embed.proj.weight", (D, cfg.in_chans, p, p), "conv_w", 0),
        ParamInfo("my_patch_embed.proj.bias", (D,), "bias", 0),
        ParamInfo("my_patch_embed_a.proj.weight", (D, 1, p, p), "conv_w", 0),
        ParamInfo("my_patch_embed_a.proj.bias", (D,), "bias", 0),
    ]
    for n, width in (("mlp_head", D), ("mlp_head_a", D), ("mlp_head_mm", 2 * D), ("mlp_head_mm_v2", D)):
        lv = 0 if n == "mlp_head_mm_v2" else P1                              # mm_v2 is constructed (:815) but no mode reads it
        spec += [
            ParamInfo(f"{n}.0.weight", (width,), "ln_w", lv),
            ParamInfo(f"{n}.0.bias", (width,), "ln_b", lv),
            ParamInfo(f"{n}.1.weight", (label_dim, width), "linear_w", lv),
            ParamInfo(f"{n}.1.bias", (label_dim,), "bias", lv),
        ]
    for n in ("mm_layer_1", "mm_layer_2"):                                   # applied with 'a' norms :946-947,1023-1024
        spec += _block(n, D, D * cfg.mlp_ratio, 0, P1, 0, P1)
    names = [s.name for s in spec]
    assert len(set(names)) == len(names)
    return spec


def state_dict_keys_ft(cfg: AVSiamConfig, label_dim: int):
    """Keys of ``CAVMAEFT_BASE.state_dict()`` in module-registration order (553 at ViT-B): vit_base, my_blocks (alias),
    my_patch_embed, my_patch_embed_a, the four heads, mm_layer_1/2."""
    spec = build_spec_ft(cfg, label_dim)
    vit = [s.name for s in spec if s.name.startswith("vit_base.")]
    keys = list(vit)
    keys += ["my_blocks." + n[len("vit_base.blocks."):] for n in vit if n.startswith("vit_base.blocks.")]
    for pre in ("my_patch_embed.", "my_patch_embed_a.", "mlp_head.", "mlp_head_a.", "mlp_head_mm.", "mlp_head_mm_v2.",
                "mm_layer_1.", "mm_layer_2."):
        keys += [s.name for s in spec if s.name.startswith(pre)]
    return keys


def alias_of(name: str) -> str:
    """``my_blocks.*`` is the same module object as ``vit_base.blocks.*`` (cav_mae_base.py:248,278)."""
    if name.startswith("my_blocks."):
        return "vit_base.blocks." + name[len("my_blocks."):]
    return name


def state_dict_keys(cfg: AVSiamConfig):
    """All keys of ``CAVMAE_BASE.state_dict()`` in module-registration order: opt is no module;
    vit_base, my_blocks (alias), my_patch_embed, my_patch_embed_a, ast_base, mm_layer_1/2, decoder_*"""
    spec = build_spec(cfg)
    by_prefix = OrderedDict()
    keys = []
    # direct parameters of the top-level module come first in nn.Module.state_dict()
    top = [s.name for s in spec if "." not in s.name]
    keys += top
    vit = [s.name for s in spec if s.name.startswith("vit_base.")]
    keys += vit
    keys += ["my_blocks." + n[len("vit_base.blocks."):] for n in vit if n.startswith("vit_base.blocks.")]
    for pre in ("my_patch_embed.", "my_patch_embed_a.", "ast_base.", "mm_layer_1.", "mm_layer_2.",
                "decoder_embed.", "decoder_blocks.", "decoder_norm.", "decoder_pred_a.", "decoder_pred_v."):
        keys += [s.name for s in spec if s.name.startswith(pre)]
    del by_prefix
    return keys


def live_names(cfg: AVSiamConfig, which: int):
    return [s.name for s in build_spec(cfg) if s.live & which]
