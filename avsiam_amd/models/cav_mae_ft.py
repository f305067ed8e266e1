"""``CAVMAEFT_BASE`` - the fine-tuned classifier's INFERENCE modes on the hand-written gfx950 kernels.

Boundary kept: the constructor and ``forward(a, v, mode, is_eval=False)`` of
/root/reference/src/models/cav_mae_base.py:744-746,827 and the 553-key ``state_dict()`` schema (so the checkpoints the
reference's fine-tuning writes, traintest_ft_base.py:255-264, load here, ``module.`` prefix or not), with the return
shapes of every mode (:847,866,892,961,1035).

Scope: forward only.  Parameters are registered with ``requires_grad=False`` and outputs
carry no autograd graph; fine-tuning (the backward of these modes, traintest_ft_base.py:143-175) is out of scope.
There is no CPU/eager fallback: ``forward`` without a GPU and libavsiam_hip.so raises.
"""
import torch
import torch.nn as nn

from .. import _lib
from ..arena import ParamArena
from ..config import AVSiamConfig
from ..param_spec import build_spec_ft
from ..weights import synth_state_ft
from .cav_mae_base import _attach

MODES = ("audioonly", "videoonly", "retrieval", "mm_grad")
MAX_ENGINES = 4          # (batch, frames) shapes whose activation buffers are kept resident


class CAVMAEFT_BASE(nn.Module):
    def __init__(self, label_dim, img_size=224, audio_length=1024, patch_size=16, in_chans=3, embed_dim=768,
                 modality_specific_depth=23, num_heads=16, mlp_ratio=4., norm_layer=nn.LayerNorm, norm_pix_loss=False,
                 tr_pos=True, *, cfg: AVSiamConfig = None, init_seed=0, init_mode="init"):
        """As in the reference every argument but ``label_dim`` is accepted and ignored (the dimensions are fixed by the
        ViT-B skeleton, :749-766); keyword-only ``cfg=`` selects other shapes."""
        super().__init__()
        self.cfg = cfg if cfg is not None else AVSiamConfig()
        self.label_dim = int(label_dim)
        self._spec = build_spec_ft(self.cfg, self.label_dim)
        self.arena = ParamArena(self.cfg, self._spec, transposed=False, grads=False)
        self.arena.load_state(synth_state_ft(self.cfg, self.label_dim, init_seed, init_mode))
        self._params = {}
        for info in self._spec:
            p = nn.Parameter(self.arena.view(info.name), requires_grad=False)
            self._params[info.name] = p
            _attach(self, info.name, p)
        self.my_blocks = self.vit_base.blocks                      # same module object, two names (:749,782)
        first = ("vit_base", "my_blocks")                          # registration order of the reference => same state_dict key order
        mods = dict(self._modules)
        self._modules.clear()
        for k in first + tuple(k for k in mods if k not in first):
            self._modules[k] = mods[k]
        self._engines = {}
        self._shadow_dirty = True

    def __create_fusion__(self):
        """mm_layer_1/2 <- copies of blocks 10 and 11 (:824-826; the fine-tune CLI calls it after loading a pre-trained
        checkpoint that lacks them)."""
        with torch.no_grad():
            for dst, src in (("mm_layer_1", self.cfg.depth - 2), ("mm_layer_2", self.cfg.depth - 1)):
                pre = f"vit_base.blocks.{src}."
                for name, p in self._params.items():
                    if name.startswith(pre):
                        self._params[dst + "." + name[len(pre):]].copy_(p)
        self._shadow_dirty = True

    def _apply(self, fn, recurse=True):
        probe = fn(torch.empty(0, dtype=torch.float32, device=self.arena.p.device))
        if probe.dtype != torch.float32:
            raise TypeError("CAVMAEFT_BASE keeps fp32 master weights; bf16 shadows are managed internally")
        if probe.device != self.arena.p.device:
            self.arena.to(probe.device)
            for name, p in self._params.items():
                p.data = self.arena.view(name)
            self._engines.clear()
            self._shadow_dirty = True
        return self

    def load_state_dict(self, state_dict, strict=True, assign=False):
        if any(k.startswith("module.") for k in state_dict):       # DDP-wrapped checkpoints (traintest_ft_base.py:255)
            state_dict = {(k[7:] if k.startswith("module.") else k): v for k, v in state_dict.items()}
        out = super().load_state_dict(state_dict, strict=strict)
        self._shadow_dirty = True
        return out

    def mark_weights_changed(self):
        self._shadow_dirty = True

    def _engine(self, batch, frames):
        key = (batch, frames)
        if key not in self._engines:
            from ..ft_engine import FtForward
            while len(self._engines) >= MAX_ENGINES:               # serving with many batch shapes: drop the oldest buffers
                self._engines.pop(next(iter(self._engines)))
            self._engines[key] = FtForward(self.arena, self.cfg, self.label_dim, batch, frames, self.arena.p.device)
        else:
            self._engines[key] = self._engines.pop(key)            # most recently used last
        return self._engines[key]

    def forward(self, a, v, mode, is_eval=False):
        """a: [B, 1024, 128] fbank; v: [B, T, 3, 224, 224] frames (either may be None when the mode ignores it).
        Returns what the reference returns for the mode; any other mode returns None as there (no else branch)."""
        if mode not in MODES:
            return None
        if not self.arena.p.is_cuda:
            raise _lib.AvsiamHipError("CAVMAEFT_BASE.forward needs a GPU: the path runs only on libavsiam_hip.so "
                                      "(no CPU/eager fallback). Move the model with .cuda() first.")
        _lib.load()
        cfg, dev = self.cfg, self.arena.p.device
        need_a, need_v = mode != "videoonly", mode != "audioonly"
        B = (a if need_a else v).shape[0]
        T = 1
        if need_a:
            if tuple(a.shape[1:]) != (cfg.audio_len, cfg.n_mels):
                raise ValueError(f"a must be [B,{cfg.audio_len},{cfg.n_mels}], got {tuple(a.shape)}")
            a = a.to(dev, torch.float32).contiguous()
        if need_v:
            if v.dim() != 5 or tuple(v.shape[2:]) != (cfg.in_chans, cfg.img_size, cfg.img_size) or v.shape[0] != B:
                raise ValueError(f"v must be [B,T,{cfg.in_chans},{cfg.img_size},{cfg.img_size}], got {tuple(v.shape)}")
            T = v.shape[1]
            v = v.to(dev, torch.float32).contiguous().view(B * T, cfg.in_chans, cfg.img_size, cfg.img_size)
        eng = self._engine(B, T)
        if self._shadow_dirty:
            self.arena.refresh_shadows(None)
            for e in self._engines.values():
                e.refresh_heads()
            self._shadow_dirty = False
        if mode == "audioonly":
            out = eng.audioonly(a).clone()
            return out.unsqueeze(1) if is_eval else out                                    # :845-847
        if mode == "videoonly":
            return eng.videoonly(v).clone().squeeze(1)                                     # :865
        if mode == "retrieval":
            if T <= 5:
                raise IndexError(f"retrieval returns frame 5 of each clip (cav_mae_base.py:892); got {T} frames")
            ta, tv = eng.retrieval(a, v)
            return ta.clone(), tv.clone()
        res = eng.mm_grad(a, v, bool(is_eval))
        if is_eval:
            return res.clone()
        return tuple(r.clone() for r in res)


CAVMAEFT = CAVMAEFT_BASE      # (/root/reference/src/models/__init__.py:8 exports the name; same signature family)


class CAVMAEFT_LARGE(CAVMAEFT_BASE):
    """``models.CAVMAEFT_LARGE`` (/root/reference/src/models/__init__.py:9; source file absent from the snapshot): the same inference modes on
    the ViT-L/16 skeleton (``config.vit_large()``); oracle-only parity (oracle/ref_cpu.py::ft_forward is shape-generic)."""

    def __init__(self, label_dim, *args, cfg: AVSiamConfig = None, **kw):
        from ..config import vit_large
        if cfg is not None and (cfg.embed_dim, cfg.num_heads) != (1024, 16):
            raise ValueError("CAVMAEFT_LARGE: cfg must be a ViT-L shape (config.vit_large(...))")
        super().__init__(label_dim, *args, cfg=cfg if cfg is not None else vit_large(), **kw)


class CAVMAEFT_HUGE(CAVMAEFT_BASE):
    """``models.CAVMAEFT_HUGE`` (/root/reference/src/models/__init__.py:13; source file absent): ViT-H/14 skeleton (``config.vit_huge14()``)."""

    def __init__(self, label_dim, *args, cfg: AVSiamConfig = None, **kw):
        from ..config import vit_huge14
        if cfg is not None and (cfg.embed_dim, cfg.num_heads) != (1280, 16):
            raise ValueError("CAVMAEFT_HUGE: cfg must be a ViT-H shape (config.vit_huge14(...) / config.vit_huge(...))")
        super().__init__(label_dim, *args, cfg=cfg if cfg is not None else vit_huge14(), **kw)
