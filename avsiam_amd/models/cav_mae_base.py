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
import os
import random as _pyrandom

import torch
import torch.nn as nn

from .. import _lib
from ..arena import ParamArena
from ..config import AVSiamConfig, EngineOptions
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
    def forward(ctx, anchor, model, audio, imgs, plan_m, plan_c, contrast_w, xf=(None, None)):
        """plan_*: False = branch off, None = draw on the device, else an explicit MaePlan / ContrastivePlan.
        xf: (audio, frames) transforms of RAW inputs (ops.InputXf) or (None, None)."""
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
                model._draw(eng)
            lm, la, lv, ma, mv = eng.forward(audio, imgs, plan_m, xf)
            out.update(loss_mae=lm.clone(), la=la.clone(), lv=lv.clone(), mask_a=ma.clone(), mask_v=mv.clone())
        else:
            out.update(loss_mae=zero.clone(), la=zero.clone(), lv=zero.clone(), mask_a=None, mask_v=None)
        if plan_c is not False:
            eng = model._engine("contrastive", B)
            if plan_c is None:
                model._draw(eng)
            lc, acc = eng.forward(audio, imgs, plan_c, contrast_w, xf)    # lc = contrast_loss_weight * nce (:735), from the kernel
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
        if not model.options.deterministic:
            return _HotPath._backward(ctx, g_mae, g_c)
        # EngineOptions.deterministic: the library's reductions into parameter gradients take their one-writer forms while THIS model's backward is
        # queued (the "det" knob is read by the launchers at enqueue time, host-side: set / restored around the schedule - another model is not affected)
        old = _lib.tuning_get("det")
        _lib.tuning_set("det", 1)
        try:
            return _HotPath._backward(ctx, g_mae, g_c)
        finally:
            _lib.tuning_set("det", old)

    @staticmethod
    def _backward(ctx, g_mae, g_c):
        model = ctx.model
        arena = model.arena
        B = ctx.batch
        live = 0
        if ctx.has_m and g_mae is not None:
            live |= P2
        if ctx.has_c and g_c is not None:
            live |= P1
        for w in (P1, P2):
            if live & w:
                model._reduced[w] = False                              # fresh local gradients
        from .. import ops as _ops
        # (every zero-fill opens a new gradient epoch - arena.zero_grad_range; engine.Stack.backward: one backward per block and epoch
        #  unless accumulate=True)
        if live & P1:
            _ops.timed("torch_zero_grads", lambda: arena.zero_grad_range(P1))
        if live & P2:
            lo, hi = arena.range[P2]
            lo = max(lo, arena.range[P1][1]) if live & P1 else lo
            _ops.timed("torch_zero_grads", lambda: arena.zero_grad_range(P2, lo, hi))
        # data parallel: the gradient all-reduce belongs to backward, as under DDP (traintest_cavmae_base.py:58-59) - chunks of
        # the flat arena are reduced as soon as the schedule declares them final (comm.GradReducer).  A backward with BOTH
        # passes live accumulates two passes into the shared range, so it is reduced once, at the end.
        red = {}
        defer = False
        if model._dp and model.reduce_in_backward:
            if live == (P1 | P2):
                red[0] = model._make_reducer(arena.range[P1][0], arena.range[P2][1], overlap=False)
            else:
                # AVSIAM_DP_DEFER=1: the MAE pass's all-reduce is split at the end of the shared parameters; the messages of the
                # MAE-ONLY parameters (audio tower, joint layers, decoder: 126 M of the 212 M) stay in flight after backward and
                # their Adam update is postponed (adam_step) - both run under the NEXT step's contrastive pass, which reads none of them
                defer = live == P2 and model.defer_p2 and not model.publish_grads
                red[live] = model._make_reducer(*arena.range[live], boundary=arena.range[P1][1] if defer else None)
        both = live == (P1 | P2)                   # the second pass adds to gradients the first has written (shared blocks)
        if live & P2:
            model._engine("mae", B).backward(g_mae.reshape(1).float().contiguous(), reducer=red.get(P2), accumulate=both)
        if live & P1:
            model._engine("contrastive", B).backward(g_c.reshape(1).float().contiguous(), ctx.contrast_w, reducer=red.get(P1), accumulate=both)
        for r in red.values():
            r.finish(defer_from=r.boundary if defer else None)
            model.last_reduce_messages = r.messages
            if defer and r.boundary is not None:
                model._deferred = {"reducer": r}
        if red:
            for w in (P1, P2):
                if live & w:
                    model._owe(w, 1.0 / model._world)                  # the SUM is in the arena; DDP's mean is still owed
                    model._reduced[w] = True
            if model.publish_grads:                                    # an external optimizer reads .grad: deliver the mean now
                for w in (P1, P2):
                    if live & w:
                        model._average(w, live)
        if model.publish_grads:
            model._publish(live)
        return (None,) * 8


class CAVMAE_BASE(nn.Module):
    """CAV-MAE / AVSiam pre-training model (reference: cav_mae_base.py:216-741)."""

    def __init__(self, img_size=224, audio_length=1024, patch_size=16, in_chans=3, embed_dim=768,
                 modality_specific_depth=23, num_heads=16, decoder_embed_dim=512, decoder_depth=8, decoder_num_heads=16,
                 mlp_ratio=4., norm_layer=nn.LayerNorm, norm_pix_loss=False, tr_pos=False, opt=None, *,
                 cfg: AVSiamConfig = None, init_seed=0, init_mode="init", plan_seed=None, verbose=True, share_pass_buffers=None,
                 fp8_mode=None, recompute=None, grad_stream=None, deterministic=None, options: EngineOptions = None):
        """Positional arguments: the reference's (cav_mae_base.py:219-222), accepted and - like there - mostly ignored.  Keyword-only
        extensions: `cfg` (shape), `fp8_mode` ("0".."3"), `recompute` ("0" | "1" | fraction), `grad_stream` ("bf16" | "fp32"),
        `deterministic` (bit-reproducible steps: one-writer reductions, one stream), `share_pass_buffers`, or a whole `options` object (config.EngineOptions) - precision and memory policy belong to THIS model;
        the AVSIAM_* environment only seeds what is not given."""
        super().__init__()
        if options is not None and any(x is not None for x in (fp8_mode, recompute, grad_stream, deterministic)):
            raise ValueError("pass either `options` or the single keywords (fp8_mode / recompute / grad_stream / deterministic), not both")
        import dataclasses as _dc
        # (a COPY of a given options object: the engines hold it by reference and set_options() edits it in place - two models built from one object stay apart)
        self.options = _dc.replace(options).validated() if options is not None else EngineOptions.from_env(fp8=fp8_mode, recompute=recompute, grad_stream=grad_stream,
                                                                                              deterministic=deterministic)
        # engine.BufferPool: the two passes of the training step (run one after the other) take their activation buffers from the SAME
        # memory - the card holds the larger pass, not the sum.  Opt-in (None: AVSIAM_SHARE_PASS_BUFFERS=1); a combined-loss forward WITH
        # gradients (both passes alive until one backward) is then refused.
        self.share_pass_buffers = (os.environ.get("AVSIAM_SHARE_PASS_BUFFERS", "0") == "1") if share_pass_buffers is None else bool(share_pass_buffers)
        self._pool = None
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
        self._comm, self._dp = None, False
        self.reduce_in_backward = True             # data parallel: loss.backward() all-reduces, like DDP (False: call allreduce_grads)
        self.defer_p2 = os.environ.get("AVSIAM_DP_DEFER", "0") == "1"      # see _HotPath.backward
        self.dp_wire = os.environ.get("AVSIAM_DP_WIRE", "fp32")     # gradient all-reduce wire format (comm.GradReducer): "fp32" | "bf16"
        self._deferred = None                      # {"reducer": in-flight all-reduce of the MAE-only gradients, "adam": its postponed update}
        self._reduced = {P1: False, P2: False}     # this pass's gradients in the arena are already summed over the ranks
        # factor the arena's gradients still owe (1/W after a SUM all-reduce), per SEGMENT of the arena: the live ranges of the two
        # passes overlap in the shared parameters, and a factor applied per pass would hit that middle segment twice
        self._grad_scale = {"p1": 1.0, "shared": 1.0, "p2": 1.0}
        self._versions = None
        self.last_reduce_messages = 0
        self._gen = None
        self._pyrng = None
        self._plan_seed = plan_seed
        self._shadow_dirty = True
        self._graph = None                         # graph_step.GraphedTrainStep: {"seed": int64 [1], "steps": {P1: int32 [1], P2: int32 [1]}} on the device

    def set_options(self, **kw):
        """Change options of this model (config.EngineOptions).  A STRUCTURAL change (precision, recompute, gradient stream, ...) drops the
        pass engines - activation buffers and fp8 records are rebuilt by the next forward; save `fp8_state()` first if the delayed scales are
        to survive.  Runtime fields (wgrad_stream, wgrad_group, deterministic) take effect on the next backward."""
        import dataclasses
        new = dataclasses.replace(self.options, **kw).validated()
        structural = any(getattr(new, f) != getattr(self.options, f) for f in EngineOptions.STRUCTURAL)
        if structural:
            self.release_buffers()
        for f in dataclasses.fields(EngineOptions):                 # in place: the engines hold this object by reference
            setattr(self.options, f.name, getattr(new, f.name))
        return self.options

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
        self.flush_deferred()
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

    def set_distributed(self, world, rank, comm=None):
        """comm: the collectives to use (comm.TorchDistComm = RCCL by default; tests inject their own)."""
        from ..comm import default_comm
        self._world, self._rank = world, rank
        self._comm = comm if comm is not None else default_comm(world)
        assert self._comm.world == world and self._comm.rank == rank, "comm does not match (world, rank)"
        self._dp = getattr(self._comm, "active", world > 1)        # collectives on the path (always at world > 1)
        self._engines.clear()
        # The gradient all-reduce overlaps the backward (comm.GradReducer, AVSIAM_DP_OVERLAP): RCCL's kernels need compute units
        # WHILE a persistent GEMM holds the chip, so every persistent kernel leaves `cu_reserve` CUs free (include/avsiam_hip.h,
        # avs_tuning_set) - 8 by default when collectives are on the path (one per XCD; AVSIAM_CU_RESERVE overrides, 0 = none).
        # One blocking message after the backward (AVSIAM_DP_OVERLAP=0) needs no reservation.  Cost at one rank: docs/rounds/r05.md.
        # (the knob is process-wide, like the device: the last model to call set_distributed decides)
        if self.arena.p.is_cuda and _lib.env_value("AVSIAM_CU_RESERVE") is None:
            overlap = os.environ.get("AVSIAM_DP_OVERLAP", "1") != "0"
            _lib.tuning_set("cu_reserve", 8 if (self._dp and overlap) else 0)

    def release_buffers(self):
        """Drop every pass engine (activation buffers, saved plans, fp8 records: `fp8_state()` first if they are to survive) and the
        shared activation pool's memory.  Engines are keyed by (pass, batch): with `share_pass_buffers` an engine of ANOTHER batch size
        (validation between training steps) rewinds the same pool, so a pending forward of the first one must not be backpropagated
        afterwards (the pool's owner check raises) - call this between phases that use different batch sizes when memory is tight."""
        self.flush_deferred()
        self._engines.clear()
        if self._pool is not None:
            self._pool.release()
        if self.arena.p.is_cuda:
            torch.cuda.empty_cache()

    def _make_reducer(self, lo, hi, overlap=None, boundary=None):
        from ..comm import GradReducer
        # the bf16 wire buffer (AVSIAM_DP_WIRE=bf16) is kept across steps and shared by the passes' reducers: it spans the whole live
        # arena and is indexed by absolute offset, so the MAE-only messages a deferred reducer leaves in flight never alias what the
        # next contrastive backward stages
        g = self.arena.ensure_grads()
        st = getattr(self, "_wire_staging", None)
        if st is None and self.dp_wire == "bf16" and self._dp:
            st = self._wire_staging = torch.empty(self.arena.live_end, dtype=torch.bfloat16, device=g.device)
        return GradReducer(self._comm, g, lo, hi, overlap=overlap, wire=self.dp_wire, staging=st, staging_lo=0 if st is not None else None,
                           boundary=boundary)

    def _segments(self, which):
        """(name, lo, hi) of the arena segments pass `which` is live in: [P1 only | shared | P2 only]"""
        b1, b2 = self.arena.range[P1]
        b12, end = self.arena.range[P2]
        return {P1: (("p1", b1, b12), ("shared", b12, b2)), P2: (("shared", b12, b2), ("p2", b2, end))}[which]

    def _owe(self, which, factor):
        for name, _, _ in self._segments(which):
            self._grad_scale[name] = factor

    def _average(self, which, live=None):
        """Apply the factor the gradients of pass `which` still owe (DDP's 1/W).  The factor is tracked per arena segment, so the
        parameters both passes share are scaled once however the calls for the two passes are ordered."""
        if self._deferred is not None and (which == P2 or live == (P1 | P2)):
            self.flush_deferred()                  # never scale a range whose all-reduce is still in flight
        for w in ((P1, P2) if live == (P1 | P2) else (which,)):
            for name, lo, hi in self._segments(w):
                s = self._grad_scale[name]
                if s != 1.0 and hi > lo:
                    self.arena.g[lo:hi].mul_(s)
                self._grad_scale[name] = 1.0

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
            if self.share_pass_buffers and self._pool is None:
                from ..engine import BufferPool
                self._pool = BufferPool(dev)
            if which == "mae":
                self._engines[key] = MaePass(self.arena, self.cfg, batch, dev, pool=self._pool, opts=self.options)
            else:
                self._engines[key] = ContrastivePass(self.arena, self.cfg, batch, dev, self._world, self._rank, self._comm, pool=self._pool, opts=self.options)
            pend = getattr(self, "_fp8_pending", None)
            if pend:                                    # a restored run continues with the quantisation grids it was saved with
                for name, st in self._fp8_stacks(which, self._engines[key]):
                    if f"{which}/{batch}/{name}" in pend:
                        st.load_fp8_state(pend[f"{which}/{batch}/{name}"])
        return self._engines[key]

    # ---- fp8 mode (EngineOptions.fp8): delayed-scaling state travels with the checkpoint ---------------------------------
    @staticmethod
    def _fp8_stacks(which, eng):
        for name in ("stack", "st_t", "st_a", "st_v", "st_mm", "st_dec"):
            st = getattr(eng, name, None)
            if st is not None and getattr(st, "fp8", False):
                yield name, st

    def fp8_state(self):
        """{'<pass>/<batch>/<stack>': scales, amax history, ring position, calibrated GEMMs} of every engine built so far ({} when the
        fp8 mode is off).  Saved beside the weights so that a resumed run quantises on the same grids instead of re-calibrating."""
        out = {}
        for (which, batch), eng in self._engines.items():
            for name, st in self._fp8_stacks(which, eng):
                out[f"{which}/{batch}/{name}"] = st.fp8_state()
        return out

    def load_fp8_state(self, state):
        self._fp8_pending = dict(state or {})
        for (which, batch), eng in self._engines.items():
            for name, st in self._fp8_stacks(which, eng):
                if f"{which}/{batch}/{name}" in self._fp8_pending:
                    st.load_fp8_state(self._fp8_pending[f"{which}/{batch}/{name}"])

    def fp8_saturation_events(self):
        """(tensor, step) pairs whose values exceeded the e4m3 range of the delayed scale they were quantised with (synchronises)"""
        return sum(st.f8.saturation_events() for (which, _), eng in self._engines.items() for _, st in self._fp8_stacks(which, eng))

    def _sync_shadows(self):
        # an optimizer the model does not know about (the reference loop's torch.optim.Adam, p.data edits, EMA) changes the fp32
        # masters in place: every such op bumps the tensor's version counter, so a changed sum marks the bf16 shadows stale.
        # Only checked when gradients are published for an external optimizer; adam_step() refreshes the shadows itself.
        if self.publish_grads:
            v = sum(p._version for p in self._params.values())
            if v != self._versions:
                self._versions = v
                self._shadow_dirty = True
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

    def _draw(self, eng):
        """this step's mask plan for `eng`, drawn on the device.  Inside a captured step (graph_step.GraphedTrainStep) the Philox key is
        read from device memory and advanced by a node of the graph; the host part of the contrastive draw runs in front of each replay."""
        g = self._graph
        if g is None:
            eng.draw_device(self._next_seed(), self._np_rng())
        else:
            eng.draw_device(None, None, seed_dev=g["seed"], host=False)
            g["seed"].add_(1)

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
                mask_mode='unstructured', *, mask_plan=None, input_xf=None):
        """Returns (loss, loss_mae, loss_mae_a, loss_mae_v, loss_c, mask_a, mask_v, c_acc) like the reference (:741).
        As in the reference, mask_ratio_* and mask_mode are ignored (ratios are fixed at :696 / :546-549) and
        mae_loss_weight only switches the MAE branch on (:694,739).
        input_xf (extension): (ops.InputXf.audio(...), ops.InputXf.frames(...)) - `audio` is then the
        UN-normalised fbank and `imgs` the uint8 frames as the reference's dataset holds them before its own arithmetic
        (dataloader.py:505-513, 461-462); the kernels that read the inputs apply it on the fly (either entry may be None)."""
        self._require_gpu()
        cfg = self.cfg
        B = audio.shape[0]
        if audio.shape[1:] != (cfg.audio_len, cfg.n_mels):
            raise ValueError(f"audio must be [B,{cfg.audio_len},{cfg.n_mels}], got {tuple(audio.shape)}")
        want_v = (B, cfg.in_chans, cfg.img_size, cfg.img_size) if imgs.dim() == 4 else \
            (B, cfg.frames, cfg.in_chans, cfg.img_size, cfg.img_size)
        if tuple(imgs.shape) != want_v or (imgs.dim() == 4 and cfg.frames != 1):
            raise ValueError(f"imgs must be {want_v} for frames={cfg.frames}, got {tuple(imgs.shape)}")
        xf = tuple(input_xf) if input_xf is not None else (None, None)
        audio = audio.to(self.arena.p.device, torch.float32).contiguous()
        if xf[1] is not None:
            if imgs.dtype != torch.uint8:
                raise ValueError("input_xf for the frames expects uint8 images")
            imgs = imgs.to(self.arena.p.device).contiguous()
        else:
            imgs = imgs.to(self.arena.p.device, torch.float32).contiguous()
        do_m, do_c = mae_loss_weight != 0, contrast_loss_weight != 0
        if self.share_pass_buffers and do_m and do_c and torch.is_grad_enabled():
            raise RuntimeError("share_pass_buffers: the two passes use the same activation memory, so a combined-loss forward with gradients "
                               "(both alive until one backward) is not available - run the passes one after the other (train_step) or build "
                               "the model without share_pass_buffers")
        if do_m:
            self.flush_deferred()                            # the MAE pass reads the parameters a deferred update still owes
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
            res = _HotPath.apply(anchor, self, audio, imgs, plan_m, plan_c, float(contrast_loss_weight), xf)
        else:
            res = _HotPath.forward(_NoCtx(), anchor, self, audio, imgs, plan_m, plan_c, float(contrast_loss_weight), xf)
        loss_mae, loss_c, la, lv, c_acc = (r.reshape(()) for r in res[:5])
        masks = res[5:]
        mask_a, mask_v = (masks[0], masks[1]) if (do_m and not do_c) else (None, None)     # :594 - the mixed encoder returns None
        loss = loss_c + loss_mae                                                          # :739
        return loss, loss_mae, la, lv, loss_c, mask_a, mask_v, c_acc

    # ---- fused data-parallel + optimizer step over the flat arena -------------------------------------------------
    def allreduce_grads(self, which, average=True, already_reduced=None):
        """c1: all-reduce over exactly the live gradient range of pass `which` (RCCL over xGMI).  ``loss.backward()`` has
        normally done the SUM already (chunked, overlapped with the backward, like DDP's buckets); gradients written into
        the arena by other means (host-side tests) are reduced here in one message - pass already_reduced=False.
        average=True (default) leaves DDP's MEAN in the arena / in ``.grad``, so the reference recipe
        ``loss.backward(); optimizer.step()`` with any torch optimizer sees what it would under DDP.  train_step passes
        average=False: the 1/W is then folded into the fused Adam kernel's grad_scale."""
        if not self._dp:
            return
        if already_reduced is None:
            already_reduced = self._reduced[which]
        if which == P2 and self._deferred is not None and (average or not already_reduced):
            self.flush_deferred()                  # about to touch the MAE-only range while its deferred all-reduce may be in flight: settle it
        if not already_reduced:
            r = self._make_reducer(*self.arena.range[which], overlap=False)
            r.finish()
            self._owe(which, 1.0 / self._world)
            self._reduced[which] = True
        if average:
            self._average(which)

    def adam_step(self, which, lr, beta1=0.95, beta2=0.999, eps=1e-8, weight_decay=5e-7):
        """torch.optim.Adam(lr, weight_decay=5e-7, betas=(0.95, 0.999)) of the reference loop (:64-66) on the pass's
        live range; each pass has its own moments and step count, like the reference's two optimizers.
        With a deferred all-reduce pending (AVSIAM_DP_DEFER) the MAE pass's update covers the shared parameters now and the
        MAE-only parameters in flush_deferred(), with the same step count and hyper-parameters - element for element the same update."""
        from .. import ops
        a = self.arena
        lo, hi = a.range[which]
        st = self._opt_state.get(which)
        if st is None:
            st = {"m": torch.zeros(hi - lo, device=a.p.device), "v": torch.zeros(hi - lo, device=a.p.device), "step": 0}
            self._opt_state[which] = st
        if which == P2 and self._deferred is not None and "adam" in self._deferred:
            self.flush_deferred()                                      # a second update before the first was applied: settle it
        g = self._graph
        if g is None:
            st["step"] += 1
            step_dev = None
        else:                                      # captured step: the count lives in device memory, advanced by this node (the host's copy by GraphedTrainStep.step)
            if self._deferred is not None:
                raise RuntimeError("a captured step does not support the deferred MAE-only update (AVSIAM_DP_DEFER)")
            step_dev = g["steps"][which]
            step_dev.add_(1)
        owed = {self._grad_scale[name] for name, slo, shi in self._segments(which) if shi > slo}
        if len(owed) > 1:                          # the pass's two segments owe different factors (mixed use): settle them first
            self._average(which)
            owed = {1.0}
        scale = owed.pop() if owed else 1.0
        end = hi
        if which == P2 and self._deferred is not None:
            end = a.range[P1][1]                                       # the shared parameters now; [end, hi) when the all-reduce has landed
            self._deferred["adam"] = (end, hi, lr, st["step"], beta1, beta2, eps, weight_decay, scale)
        if end > lo:
            ops.adam(a.p[lo:end], a.g[lo:end], st["m"][:end - lo], st["v"][:end - lo], a.pb[lo:end], end - lo, lr, st["step"], beta1, beta2, eps,
                     weight_decay, scale, step_dev=step_dev)
        for name, _, _ in self._segments(which):
            self._grad_scale[name] = 1.0
        if end == hi:
            a.refresh_shadows(which, cast=False)
        elif end > lo:
            a.refresh_shadows(which, cast=False, span=(lo, end))

    def flush_deferred(self):
        """Complete a deferred MAE-only update (AVSIAM_DP_DEFER): order the stream behind the all-reduce messages still in flight,
        then apply the postponed Adam update and refresh the weight shadows.  Called before anything reads those parameters or
        overwrites their gradients: the next MAE forward, state_dict(), the optimizer state, another update."""
        d, self._deferred = self._deferred, None
        if d is None:
            return
        from .. import ops
        d["reducer"].wait_deferred()
        if "adam" in d:
            a = self.arena
            lo = a.range[P2][0]
            b, hi, lr, step, beta1, beta2, eps, wd, scale = d["adam"]
            st = self._opt_state[P2]
            ops.adam(a.p[b:hi], a.g[b:hi], st["m"][b - lo:hi - lo], st["v"][b - lo:hi - lo], a.pb[b:hi], hi - b, lr, step, beta1, beta2, eps, wd, scale)
            a.refresh_shadows(P2, cast=False, span=(b, hi))

    def state_dict(self, *args, **kwargs):
        self.flush_deferred()
        return super().state_dict(*args, **kwargs)

    # ---- optimizer state in torch.optim.Adam's format (best_optim_state.pth, traintest_cavmae_base.py:230) -----------
    def optimizer_state_dict(self, which, lr, beta1=0.95, beta2=0.999, eps=1e-8, weight_decay=5e-7):
        """State of the pass's Adam as ``torch.optim.Adam(trainables, ...).state_dict()`` would hold it: parameters indexed in
        ``parameters()`` order; parameters the pass never touched (grad None) have no state, as in torch."""
        self.flush_deferred()
        a = self.arena
        lo, hi = a.range[which]
        st = self._opt_state.get(which)
        state = {}
        params = list(self.parameters())
        index = {id(p): i for i, p in enumerate(params)}
        if st is not None:
            for name, p in self._params.items():
                info = a.info[name]
                if not (info.live & which):
                    continue
                o, n = a.offset[name] - lo, p.numel()
                state[index[id(p)]] = {"step": torch.tensor(float(st["step"])),
                                       "exp_avg": st["m"][o:o + n].view(p.shape).detach().cpu().clone(),
                                       "exp_avg_sq": st["v"][o:o + n].view(p.shape).detach().cpu().clone()}
        group = {"lr": lr, "betas": (beta1, beta2), "eps": eps, "weight_decay": weight_decay, "amsgrad": False, "maximize": False,
                 "foreach": None, "capturable": False, "differentiable": False, "fused": None, "params": list(range(len(params)))}
        return {"state": state, "param_groups": [group]}

    def load_optimizer_state_dict(self, which, sd):
        a = self.arena
        lo, hi = a.range[which]
        dev = a.p.device
        st = {"m": torch.zeros(hi - lo, device=dev), "v": torch.zeros(hi - lo, device=dev), "step": 0}
        params = list(self.parameters())
        name_of = {id(p): n for n, p in self._params.items()}
        for i, s in sd["state"].items():
            name = name_of[id(params[int(i)])]
            o, n = a.offset[name] - lo, params[int(i)].numel()
            st["m"][o:o + n].copy_(s["exp_avg"].reshape(-1))
            st["v"][o:o + n].copy_(s["exp_avg_sq"].reshape(-1))
            st["step"] = int(s["step"])
        self._opt_state[which] = st


class _NoCtx:
    """Stand-in ctx for no-grad forwards (validate(), traintest_cavmae_base.py:381-424)."""

    def set_materialize_grads(self, v):
        pass

    def mark_non_differentiable(self, *a):
        pass


CAVMAE = CAVMAE_BASE          # north-star wording "CAVMAE": same signature family


class CAVMAE_LARGE(CAVMAE_BASE):
    """``models.CAVMAE_LARGE`` (/root/reference/src/models/__init__.py:9; its source file cav_mae_large.py is absent from the snapshot): the same
    model on a ViT-L/16 skeleton - BASELINE.json configs[3].  Same constructor and forward as CAVMAE_BASE; the shape is ``config.vit_large()``
    unless a ``cfg`` is given (frames, audio tokens).  Parity is pinned by the oracle only (no reference source for this width)."""

    def __init__(self, *args, cfg: AVSiamConfig = None, **kw):
        from ..config import vit_large
        if cfg is not None and (cfg.embed_dim, cfg.num_heads) != (1024, 16):
            raise ValueError("CAVMAE_LARGE: cfg must be a ViT-L shape (config.vit_large(...))")
        super().__init__(*args, cfg=cfg if cfg is not None else vit_large(), **kw)


class CAVMAE_HUGE(CAVMAE_BASE):
    """``models.CAVMAE_HUGE`` (/root/reference/src/models/__init__.py:13; source file absent): ViT-H/14 skeleton - 1280 wide, 32 layers, 16 heads
    of 80, 14 x 14 patches (256 tokens per frame, 9 x 73 audio tokens) - BASELINE.json configs[4].  ``fp8_mode="3"`` selects that config's
    fp8 MFMA path for this model alone.  Shape ``config.vit_huge14()`` unless a ``cfg`` is given; oracle-only parity."""

    def __init__(self, *args, cfg: AVSiamConfig = None, **kw):
        from ..config import vit_huge14
        if cfg is not None and (cfg.embed_dim, cfg.num_heads) != (1280, 16):
            raise ValueError("CAVMAE_HUGE: cfg must be a ViT-H shape (config.vit_huge14(...) / config.vit_huge(...))")
        super().__init__(*args, cfg=cfg if cfg is not None else vit_huge14(), **kw)
