"""Shape configuration of the AVSiam pre-training hot path.

The reference hard-codes every dimension inside ``CAVMAE_BASE.__init__``
(/root/reference/src/models/cav_mae_base.py:248-261 encoder 768/12 heads/12 layers,
:316-329 decoder 512/16 heads/8 layers, :298 512 audio tokens).  This dataclass makes
them explicit so the same engine runs the reference-native shape, the small parity
shape of BASELINE.json configs[0] (128 audio tokens) and the T-frame extension of
configs[1] (multi-frame note).
"""
from dataclasses import dataclass


@dataclass(frozen=True)
class AVSiamConfig:
    embed_dim: int = 768
    depth: int = 12
    num_heads: int = 12
    dec_dim: int = 512
    dec_depth: int = 8
    dec_heads: int = 16
    mlp_ratio: int = 4
    patch: int = 16
    img_size: int = 224
    in_chans: int = 3
    audio_tokens: int = 512      # La: pos_embed_a length (cav_mae_base.py:298)
    audio_f: int = 8             # frequency patches: 128 mel bins / 16
    frames: int = 1              # T: frames per sample (reference pre-training: 1)
    n_classes: int = 21843       # dead in21k head kept for checkpoint-key compatibility
    stride: int = 0              # patch stride = the patch extent actually used; 0 = `patch`.  14 with patch 16: the 14 x 14 patch grid
                                 # (ViT-H/14) on 16 x 16 STORAGE - patch-embedding kernels and prediction rows keep 256 positions per
                                 # channel, of which the 14 x 14 corner is read / scored and the rest is dead weight
    n_groups: int = 5            # multi-ratio groups of the contrastive pass (:534)
    mae_mask_ratio: float = 0.75  # hard-coded at :696
    temperature: float = 0.05    # :647

    @property
    def st(self):                # patch stride (= used extent)
        return self.stride or self.patch

    @property
    def audio_t(self):           # time patches (64 for target_length 1024)
        return self.audio_tokens // self.audio_f

    @property
    def video_tokens(self):      # Lv per frame
        return (self.img_size // self.st) ** 2

    @property
    def grid(self):
        return self.img_size // self.st

    @property
    def head_dim(self):
        return self.embed_dim // self.num_heads

    @property
    def dec_head_dim(self):
        return self.dec_dim // self.dec_heads

    @property
    def keep_a(self):            # tokens kept by the MAE pass (int(L*(1-r)), :372)
        return int(self.audio_tokens * (1 - self.mae_mask_ratio))

    @property
    def keep_v(self):
        return int(self.video_tokens * (1 - self.mae_mask_ratio))

    @property
    def audio_len(self):         # spectrogram frames (target_length)
        return self.audio_t * self.st

    @property
    def n_mels(self):
        return self.audio_f * self.st


def vit_base(**kw):
    return AVSiamConfig(**kw)


def vit_large(**kw):
    """ViT-L/16 re-parameterisation (BASELINE.json configs[3]); no reference source exists
    for it so parity for this shape is pinned by the oracle only."""
    kw.setdefault("depth", 24)
    return AVSiamConfig(embed_dim=1024, num_heads=16, **kw)


def vit_huge(**kw):
    """ViT-H width (timm vit_huge: 1280 wide, 32 layers, 16 heads of 80, MLP 5120) on this path's 16 x 16 patch grid - the encoder
    of BASELINE.json configs[4] in bf16.  Not the /14 patch grid (588 = 14*14*3 is not a multiple of the GEMMs' 64-wide K step) and
    not fp8; like ViT-L there is no reference source for it, parity is pinned by the oracle only.  ``depth`` may be overridden for
    small test shapes."""
    kw.setdefault("depth", 32)
    return AVSiamConfig(embed_dim=1280, num_heads=16, **kw)


def vit_huge14(**kw):
    """ViT-H/14 geometry (BASELINE.json configs[4]): 14 x 14 patches - 256 tokens per 224 x 224 frame, 9 x 73 = 657 tokens for a
    1022 x 126 corner of the 1024 x 128 spectrogram - on 16 x 16 patch storage (``stride``), so the GEMMs keep their 64-aligned K."""
    kw.setdefault("depth", 32)
    kw.setdefault("audio_tokens", 657)
    kw.setdefault("audio_f", 9)
    return AVSiamConfig(embed_dim=1280, num_heads=16, stride=14, **kw)


# =====================================================================================================================
@dataclass
class EngineOptions:
    """How ONE model runs the hot path - precision, activation memory, backward schedule.  A property of the model (round 6): two models with
    different precisions live side by side in one process, and the entry point chooses per run (`--fp8`, `--recompute`).  The `AVSIAM_*`
    environment variables only SEED the defaults (`from_env`, read when a model is constructed without the keyword); nothing in the engine
    reads the environment or a module global any more.

    STRUCTURAL fields are read when a pass engine is built (buffers, fp8 records and kernels selected then): change them through
    `CAVMAE_BASE.set_options`, which drops the engines.  RUNTIME fields are read on every call.

    fp8           "0" bf16 operands (the headline metric) | "1" e4m3 forward GEMMs | "2" + e5m2 x e4m3 input gradients | "3" + fp8 weight
                  gradients (BASELINE.json configs[4]'s "fp8 MFMA path"; engine.Stack)                                           [structural]
    fp8_lean      mode 3: producers whose bf16 output has no reader left write the 8-bit copy only                                [structural]
    fp8_gelu8     modes 2 / 3: gelu'(x) travels between the fc1 forward and fc2 input-gradient epilogues as 8-bit codes           [structural]
    gelu8         the same 8-bit gelu'(x) codes in the BF16 path (round 6, default ON: every golden / oracle assertion holds at unchanged
                  tolerances with margins within 5 % of the bf16 operand's, +0.7 % throughput, -7 GiB; profiles/r06/ab_gelu8.txt)              [structural]
    recompute     "0" | "1" | fraction: leading blocks of every stack that keep no activations and re-run their forward           [structural]
    grad_stream   "bf16" | "fp32": the residual-GRADIENT stream between the blocks of a stack                                     [structural]
    attn_tile     0 automatic | 64 | 128: rows per attention workgroup (A/B)                                                      [structural]
    attn_fused    sequences of at most 128 tokens take the single-workgroup attention backward                                    [structural]
    attn_fused224 ... and those of 129 .. 224 tokens at head dim 64 the 7-wave form (round 6: bitwise-tested, measured 2 % SLOWER than the
                  two-kernel form on the contrastive mix - one 142-KB workgroup per CU serialises load / phase 1 / phase 2 - so OFF by default;
                  profiles/r06/attn_fused224_ab.log)                                                                              [structural]
    group_towers  the MAE pass's audio and visual towers as one packed stack with two weight sets per launch                      [structural]
    prune_dead    pass 2 skips the rows whose loss and gradient are identically zero: prediction heads and their backward on MASKED
                  rows only; in the last decoder block query / proj / LN2 / MLP / decoder_norm on masked rows only (K, V for all)     [structural]
    wgrad_stream  "2" weight gradients on a second stream beside attention / LayerNorm backward | "1" beside everything | "0" one
                  stream | "auto" (default): "2" for stacks of at least 32 768 packed rows (the headline shape: every stack), "1" below - the
                  reference's one-frame shapes, whose forward / input-gradient GEMMs leave partial rounds for them to fill (+0.9 % at batch 64)      [runtime]
    wgrad_group   a block's fc2 / fc1 / proj weight gradients in one launch                                                       [runtime]
    batch_reduce  single GPU, no gradient accumulation: the small per-block reductions of a stack's backward - the parameter-gradient
                  reduce of every LayerNorm backward, the value third of every qkv bias gradient - in ONE launch each at the end of the
                  stack's backward instead of ~100 launches of ~10 us inside it (each backward keeps a slab workspace of its own)          [runtime]
    deterministic weight gradients with ONE writer per output tile and an ordered contraction (no split over the token rows, no
                  cross-workgroup atomics), everything on one stream: two runs of a step are bit-identical.  For debugging (the
                  first multi-GPU session); slower                                                                                [runtime]
    """
    fp8: str = "0"
    fp8_lean: bool = True
    fp8_gelu8: bool = True
    gelu8: bool = True
    recompute: str = "0"
    grad_stream: str = "bf16"
    attn_tile: int = 0
    attn_fused: bool = True
    attn_fused224: bool = False
    group_towers: bool = True
    prune_dead: bool = True
    wgrad_stream: str = "auto"
    wgrad_group: bool = True
    batch_reduce: bool = True
    deterministic: bool = False

    STRUCTURAL = ("fp8", "fp8_lean", "fp8_gelu8", "gelu8", "recompute", "grad_stream", "attn_tile", "attn_fused", "attn_fused224", "group_towers", "prune_dead")

    _ENV = {"fp8": ("AVSIAM_FP8", str), "fp8_lean": ("AVSIAM_FP8_LEAN", "flag"), "fp8_gelu8": ("AVSIAM_FP8_GELU8", "flag"), "gelu8": ("AVSIAM_GELU8", "flag"),
            "recompute": ("AVSIAM_RECOMPUTE", str), "grad_stream": ("AVSIAM_GRAD_STREAM", str), "attn_tile": ("AVSIAM_ATTN_TILE", int),
            "attn_fused": ("AVSIAM_ATTN_FUSED", "flag"), "attn_fused224": ("AVSIAM_ATTN_FUSED224", "flag"), "group_towers": ("AVSIAM_GROUP_TOWERS", "flag"), "prune_dead": ("AVSIAM_PRUNE_DEAD", "flag"),
            "wgrad_stream": ("AVSIAM_WGRAD_STREAM", str), "wgrad_group": ("AVSIAM_WGRAD_GROUP", "flag"), "batch_reduce": ("AVSIAM_BATCH_REDUCE", "flag"), "deterministic": ("AVSIAM_DETERMINISTIC", "flag")}

    @classmethod
    def from_env(cls, **over):
        """defaults <- AVSIAM_* environment (unset or empty: the default) <- keyword overrides that are not None"""
        import os
        kw = {}
        for field, (env, kind) in cls._ENV.items():
            v = os.environ.get(env)
            if v in (None, ""):
                continue
            kw[field] = (v != "0") if kind == "flag" else kind(v)
        kw.update({k: v for k, v in over.items() if v is not None})
        return cls(**kw).validated()

    def validated(self):
        self.fp8, self.recompute, self.wgrad_stream = str(self.fp8), str(self.recompute), str(self.wgrad_stream)
        if self.fp8 not in ("0", "1", "2", "3"):
            raise ValueError(f"fp8 mode must be 0, 1, 2 or 3, not {self.fp8!r}")
        f = float(self.recompute)
        if not 0.0 <= f <= 1.0:
            raise ValueError(f"recompute must be 0, 1 or a fraction between them, not {self.recompute!r}")
        if self.grad_stream not in ("bf16", "fp32"):
            raise ValueError(f"grad_stream must be bf16 or fp32, not {self.grad_stream!r}")
        if self.wgrad_stream not in ("0", "1", "2", "auto"):
            raise ValueError(f"wgrad_stream must be 0, 1, 2 or auto, not {self.wgrad_stream!r}")
        if self.attn_tile not in (0, 64, 128):
            raise ValueError(f"attn_tile must be 0, 64 or 128, not {self.attn_tile!r}")
        return self

    def describe(self):
        import dataclasses
        return {k: v for k, v in dataclasses.asdict(self).items()}
