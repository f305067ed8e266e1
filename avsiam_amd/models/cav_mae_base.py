"""``CAVMAE_BASE`` - drop-in for the reference pre-training model, running on hand-written gfx950 kernels.

Boundary kept: the constructor and ``forward`` signatures of
/root/reference/src/models/cav_mae_base.py:219-222,685,741, an ``nn.Module`` whose ``parameters()`` /
``state_dict()`` expose the reference's 963-key schema (incl. the ``my_blocks`` alias), and a differentiable
``loss`` so ``loss.backward()`` fills ``.grad`` exactly as the reference loop expects
(/root/reference/src/traintest_cavmae_base.py:131-152).

Differences that are deliberate and documented:
* weights: no network here, so instead of timm's pretrained ViT (:236,240) the constructor synthesises a
  reference-like initial state (weights.py); ``load_state_dict`` accepts reference checkpoints;
* like the reference, most constructor arguments are accepted and ignored (dims are fixed by the model family,
  :248-261,316-329); a keyword-only ``cfg=`` selects other shapes (T frames, 128 audio tokens, ViT-L);
* ``mask_plan=`` (keyword-only) injects the token selection; without it the plan is drawn ON THE DEVICE by the
  mask-plan kernel (csrc/maskplan.hip, Philox streams keyed by the model's plan seed) with the reference's
  distribution; ``draw_plans`` is the equivalent host generator (maskplan.py);
* the forward/backward of a pass is ONE autograd node (hand-scheduled backward); gradients are delivered through
  ``.grad`` views of a flat arena, so ``torch.autograd.grad`` on individual parameters and gradient
  accumulation across several backward calls are not supported (the reference loop does neither).
There is no CPU/eager fallback: calling ``forward`` without a GPU and libavsiam_hip.so raises.
"""
import random as _pyrandom

import torch
import torch.nn as nn

from .. import _lib
from ..arena import ParamArena
from ..config import AVSiamConfig
from ..maskplan import ContrastivePlan, MaePlan, make_contrastive_plan, make_mae_plan
from ..param_spec import P1, P2, build_spec
from ..weights import synth_state


class _Holder(nn.Module):
    """Attribute container reproducing the reference's module tree (parameters only)."""


def _attach(root, dotted, param):
    parts = dotted.split(".")
    mod = root
    for p in parts[:-1]:
        if p not in mod._modules:
            mod.add_module(p, _Holder())
        mod = mod._modules[p]
    mod.register_parameter(parts[-1], param)


class _HotPath(torch.autograd.Function):
    """One node for a whole pass: forward launches the kernel schedule, backward the hand-written reverse."""

    @staticmethod
    def forward(ctx, anchor, model, audio, imgs, plan_m, plan_c, contrast_w):
        """plan_*: False = branch off, None = draw on the device, else an explicit MaePlan / ContrastivePlan."""
        ctx.set_materialize_grads(False)
        ctx.model, ctx.contrast_w = model, contrast_w
        ctx.has_m, ctx.has_c = plan_m is not False, plan_c is not False
        dev = audio.device
        zero = torch.zeros(1, device=dev)
        B = audio.shape[0]
        out = {}
        if plan_m is not False:
            eng = model._engine("mae", B)
            if plan_m is None:
                eng.draw_device(model._next_seed(), model._np_rng())
            lm, la, lv, ma, mv = eng.forward(audio, imgs, plan_m)
            out.update(loss_mae=lm.clone(), la=la.clone(), lv=lv.clone(), mask_a=ma.clone(), mask_v=mv.clone())
        else:
            out.update(loss_mae=zero.clone(), la=zero.clone(), lv=zero.clone(), mask_a=None, mask_v=None)
        if plan_c is not False:
            eng = model._engine("contrastive", B)
            if plan_c is None:
                eng.draw_device(model._next_seed(), model._np_rng())
            lc, acc = eng.forward(audio, imgs, plan_c, contrast_w)        # lc = contrast_loss_weight * nce (:735), from the kernel
            out.update(loss_c=lc.clone(), c_acc=acc.clone())
        else:
            out.update(loss_c=zero.clone(), c_acc=zero.clone())
        ctx.batch = B
        masks = [m for m in (out["mask_a"], out["mask_v"]) if m is not None]
        ctx.mark_non_differentiable(out["c_acc"], out["la"], out["lv"], *masks)
        ctx.n_masks = len(masks)
        return (out["loss_mae"], out["loss_c"], out["la"], out["lv"], out["c_acc"], *masks)

    @staticmethod
    def backward(ctx, g_mae, g_c, *unused):
        model = ctx.model
        arena = model.arena
        B = ctx.batch
        live = 0
        if ctx.has_m and g_mae is not None:
            live |= P2
        if ctx.has_c and g_c is not None:
            live |= P1
        if live & P1:
            arena.zero_grad_range(P1)
        if live & P2:
            lo, hi = arena.range[P2]
            lo = max(lo, arena.range[P1][1]) if live & P1 else lo
            arena.g[lo:hi].zero_()
        if live & P2:
            model._engine("mae", B).backward(g_mae.reshape(1).float().contiguous())
        if live & P1:
            model._engine("contrastive", B).backward(g_c.reshape(1).float().contiguous(), ctx.contrast_w)
        if model.publish_grads:
            model._publish(live)
        return (None,) * 7


class CAVMAE_BASE(nn.Module):
    """CAV-MAE / AVSiam pre-training model (reference: cav_mae_base.py:216-741)."""

    def __init__(self, img_size=224, audio_length=1024, patch_size=16, in_chans=3, embed_dim=768,
                 modality_specific_depth=23, num_heads=16, decoder_embed_dim=512, decoder_depth=8, decoder_num_heads=16,
                 mlp_ratio=4., norm_layer=nn.LayerNorm, norm_pix_loss=False, tr_pos=False, opt=None, *,
                 cfg: AVSiamConfig = None, init_seed=0, init_mode="init", plan_seed=None, verbose=True):
        super().__init__()
        if verbose:
            print('A CAV-MAE Model')                              # reference prints (:224-226)
            print('Use norm_pix_loss: ', norm_pix_loss)
            print('Learnable Positional Embedding: ', tr_pos)
        self.opt = opt
        self.cfg = cfg if cfg is not None else AVSiamConfig()
        self.arena = ParamArena(self.cfg)
        self.arena.load_state(synth_state(self.cfg, init_seed, init_mode))
        self._params = {}
        for info in build_spec(self.cfg):
            p = nn.Parameter(self.arena.view(info.name), requires_grad=True)
            self._params[info.name] = p
            _attach(self, info.name, p)
        self.my_blocks = self.vit_base.blocks                      # same module object, two names (:248,278)
        self.publish_grads = True
        self._engines = {}
        self._opt_state = {}
        self._world, self._rank = 1, 0
        self._gen = None
        self._pyrng = None
        self._plan_seed = plan_seed
        self._shadow_dirty = True

    # ---- device management: parameters are views of one flat buffer, so move the buffer and re-point them -------
    def _apply(self, fn, recurse=True):
        probe = fn(torch.empty(0, dtype=torch.float32, device=self.arena.p.device))
        if probe.dtype != torch.float32:
            raise TypeError("CAVMAE_BASE keeps fp32 master weights; bf16 shadows are managed internally")
        if probe.device != self.arena.p.device:
            self.arena.to(probe.device)
            for name, p in self._params.items():
                p.data = self.arena.view(name)
                p.grad = None
            self._engines.clear()
            self._opt_state.clear()
            self._shadow_dirty = True
        return self

    def load_state_dict(self, state_dict, strict=True, assign=False):
        out = super().load_state_dict(state_dict, strict=strict)
        self._shadow_dirty = True
        return out

    def load_vit_pretrained(self, vit_state_dict, seed=0):
        """Initialise from a timm ViT-B/16 checkpoint (the reference hard-codes jx_vit_base_patch16_224_in21k,
        cav_mae_base.py:236-240) - see weights.state_from_vit for the derivation of the audio / per-modality copies."""
        from ..weights import state_from_vit
        self.arena.load_state(state_from_vit(vit_state_dict, self.cfg, seed))
        self._shadow_dirty = True
        self._opt_state.clear()

    def mark_weights_changed(self):
        """Call after modifying parameters outside adam_step() (e.g. an external optimizer)."""
        self._shadow_dirty = True

    def set_distributed(self, world, rank):
        self._world, self._rank = world, rank
        self._engines.clear()

    # ---- engines ---------------------------------------------------------------------------------------------
    def _require_gpu(self):
        if not self.arena.p.is_cuda:
            raise _lib.AvsiamHipError("CAVMAE_BASE.forward needs a GPU: the hot path runs only on libavsiam_hip.so "
                                      "(no CPU/eager fallback). Move the model with .cuda() first.")
        _lib.load()

    def _engine(self, which, batch):
        key = (which, batch)
        if key not in self._engines:
            from ..engine import ContrastivePass, MaePass
            dev = self.arena.p.device
            if which == "mae":
                self._engines[key] = MaePass(self.arena, self.cfg, batch, dev)
            else:
                self._engines[key] = ContrastivePass(self.arena, self.cfg, batch, dev, self._world, self._rank)
        return self._engines[key]

    def _sync_shadows(self):
        if self._shadow_dirty:
            self.arena.refresh_shadows(None)
            self._shadow_dirty = False

    def _publish(self, live):
        for info in build_spec(self.cfg):
            if info.live & live:
                p = self._params[info.name]
                gv = self.arena.gview(info.name)
                if p.grad is None or p.grad.data_ptr() == gv.data_ptr():
                    p.grad = gv
                else:
                    p.grad.add_(gv)

    # ---- plans -------------------------------------------------------------------------------------------------
    def _rngs(self):
        if self._gen is None:
            seed = self._plan_seed if self._plan_seed is not None else int(torch.initial_seed() % (2 ** 31))
            self._gen = torch.Generator().manual_seed(seed)
            self._pyrng = _pyrandom.Random(seed)
            import numpy as np
            self._nprng = np.random.default_rng(seed)
            self._seed_base = seed
        return self._gen, self._pyrng

    def _next_seed(self):
        """64-bit Philox key of the next device-side plan: (base seed, draw counter)."""
        self._rngs()
        self._draws = getattr(self, "_draws", 0) + 1
        return ((self._seed_base & 0xFFFFFFFF) << 32) | (self._draws & 0xFFFFFFFF)

    def _np_rng(self):
        self._rngs()
        return self._nprng

    def last_plans(self, batch):
        """The plans the device drew in the latest forward (rebuilt from the device buffers; synchronises)."""
        out = {}
        for which in ("mae", "contrastive"):
            eng = self._engines.get((which, batch))
            if eng is not None:
                out[which] = eng.last_plan()
        return out

    def draw_plans(self, batch, mae=True, contrastive=True):
        gen, pyrng = self._rngs()
        pm = make_mae_plan(self.cfg, batch, gen) if mae else None
        pc = make_contrastive_plan(self.cfg, batch, gen, pyrng) if contrastive else None
        return pm, pc

    # ---- forward (reference signature, :685) --------------------------------------------------------------------
    def forward(self, audio, imgs, mask_ratio_a=0.75, mask_ratio_v=0.75, mae_loss_weight=1., contrast_loss_weight=0.01,
                mask_mode='unstructured', *, mask_plan=None):
        """Returns (loss, loss_mae, loss_mae_a, loss_mae_v, loss_c, mask_a, mask_v, c_acc) like the reference (:741).
        As in the reference, mask_ratio_* and mask_mode are ignored (ratios are fixed at :696 / :546-549) and
        mae_loss_weight only switches the MAE branch on (:694,739)."""
        self._require_gpu()
        cfg = self.cfg
        B = audio.shape[0]
        if audio.shape[1:] != (cfg.audio_len, cfg.n_mels):
            raise ValueError(f"audio must be [B,{cfg.audio_len},{cfg.n_mels}], got {tuple(audio.shape)}")
        want_v = (B, cfg.in_chans, cfg.img_size, cfg.img_size) if imgs.dim() == 4 else \
            (B, cfg.frames, cfg.in_chans, cfg.img_size, cfg.img_size)
        if tuple(imgs.shape) != want_v or (imgs.dim() == 4 and cfg.frames != 1):
            raise ValueError(f"imgs must be {want_v} for frames={cfg.frames}, got {tuple(imgs.shape)}")
        audio = audio.to(self.arena.p.device, torch.float32).contiguous()
        imgs = imgs.to(self.arena.p.device, torch.float32).contiguous()
        do_m, do_c = mae_loss_weight != 0, contrast_loss_weight != 0
        plan_m = plan_c = None
        if isinstance(mask_plan, dict):
            plan_m, plan_c = mask_plan.get("mae"), mask_plan.get("contrastive")
        elif isinstance(mask_plan, MaePlan):
            plan_m = mask_plan
        elif isinstance(mask_plan, ContrastivePlan):
            plan_c = mask_plan
        if not do_m:
            plan_m = False                                   # False: branch off; None: draw the plan on the device
        if not do_c:
            plan_c = False
        self._sync_shadows()
        anchor = self._params["vit_base.norm.weight"]
        if torch.is_grad_enabled():
            res = _HotPath.apply(anchor, self, audio, imgs, plan_m, plan_c, float(contrast_loss_weight))
        else:
            res = _HotPath.forward(_NoCtx(), anchor, self, audio, imgs, plan_m, plan_c, float(contrast_loss_weight))
        loss_mae, loss_c, la, lv, c_acc = (r.reshape(()) for r in res[:5])
        masks = res[5:]
        mask_a, mask_v = (masks[0], masks[1]) if (do_m and not do_c) else (None, None)     # :594 - the mixed encoder returns None
        loss = loss_c + loss_mae                                                          # :739
        return loss, loss_mae, la, lv, loss_c, mask_a, mask_v, c_acc

    # ---- fused data-parallel + optimizer step over the flat arena -------------------------------------------------
    def allreduce_grads(self, which):
        """c1: ONE all-reduce(SUM) over exactly the live gradient range of the pass (RCCL over xGMI); the 1/W of DDP's
        mean is folded into adam_step's grad_scale."""
        if self._world > 1:
            import torch.distributed as dist
            g = self.arena.live_slice(self.arena.ensure_grads(), which)
            if g.is_cuda and dist.get_backend() == "gloo":    # test path: stage through the host
                h = g.cpu()
                dist.all_reduce(h)
                g.copy_(h)
            else:
                dist.all_reduce(g)

    def adam_step(self, which, lr, beta1=0.95, beta2=0.999, eps=1e-8, weight_decay=5e-7):
        """torch.optim.Adam(lr, weight_decay=5e-7, betas=(0.95, 0.999)) of the reference loop (:64-66) on the pass's
        live range; each pass has its own moments and step count, like the reference's two optimizers."""
        from .. import ops
        a = self.arena
        lo, hi = a.range[which]
        st = self._opt_state.get(which)
        if st is None:
            st = {"m": torch.zeros(hi - lo, device=a.p.device), "v": torch.zeros(hi - lo, device=a.p.device), "step": 0}
            self._opt_state[which] = st
        st["step"] += 1
        ops.adam(a.p[lo:hi], a.g[lo:hi], st["m"], st["v"], a.pb[lo:hi], hi - lo, lr, st["step"], beta1, beta2, eps,
                 weight_decay, 1.0 / self._world)
        a.refresh_shadows(which, cast=False)


class _NoCtx:
    """Stand-in ctx for no-grad forwards (validate(), traintest_cavmae_base.py:381-424)."""

    def set_materialize_grads(self, v):
        pass

    def mark_non_differentiable(self, *a):
        pass


CAVMAE = CAVMAE_BASE          # north-star wording "CAVMAE": same signature family
