"""Flat parameter arena: all 723 tensors of CAVMAE_BASE in ONE fp32 HBM buffer.

Layout (elements, each tensor aligned to 64):  [ pass-1 only | both passes | pass-2 only | dead ]
so the live set of each pass of the training step (/root/reference/src/traintest_cavmae_base.py:131-152) is one
contiguous range.  That buys, per pass: one gradient zero-fill, one RCCL all-reduce over exactly the live
gradients (86.4 M / 212.1 M parameters instead of the 248 M everything-buckets DDP registers),
one fused Adam launch, and one fp32->bf16 shadow refresh.  The reference module tree sees the arena through
nn.Parameter views, so state_dict()/parameters() keep the reference's 963-key schema.

Shadows used by the kernels:  pb   bf16 copy of the live range (GEMM B operands [N,K] as stored by nn.Linear)
                              wt   bf16 transposed copies [K,N] of the Linear weights whose input needs a gradient
                                   (B operand of the dgrad GEMM)
"""
import math

import torch

from .config import AVSiamConfig
from .param_spec import P1, P2, build_spec

ALIGN = 64


def _is_matrix(info):
    return info.kind in ("linear_w", "conv_w")


def _needs_transpose(info):
    # Linear layers whose input gradient is needed; the patch-embed convs take raw pixels (no dgrad)
    return info.kind == "linear_w" and info.live != 0


class ParamArena:
    def __init__(self, cfg: AVSiamConfig, spec=None, transposed=True, grads=True):
        """spec: parameter schema (default: CAVMAE_BASE's); transposed=False skips the [K,N] weight copies, which only
        the backward (dgrad) GEMMs read, and grads=False the flat gradient buffer - inference-only models need neither."""
        self.cfg = cfg
        self.with_grads = grads
        spec = build_spec(cfg) if spec is None else spec
        order = ([s for s in spec if s.live == P1] + [s for s in spec if s.live == (P1 | P2)] +
                 [s for s in spec if s.live == P2] + [s for s in spec if s.live == 0])
        self.info = {s.name: s for s in spec}
        self.offset = {}
        off = 0
        begin = {}                       # first offset of each liveness class present
        for s in order:
            begin.setdefault(s.live, off)
            self.offset[s.name] = off
            off += (math.prod(s.shape) + ALIGN - 1) // ALIGN * ALIGN
        self.total = off
        self.live_end = begin.get(0, off)
        b2 = begin.get(P2, self.live_end)            # classes that are absent collapse to empty ranges
        b12 = begin.get(P1 | P2, b2)
        b1 = begin.get(P1, b12)
        self.range = {P1: (b1, b2), P2: (b12, self.live_end)}
        self.names = [s.name for s in order]
        # transposed-copy arena
        self.t_offset = {}
        toff = 0
        for s in order:
            if transposed and _needs_transpose(s):
                self.t_offset[s.name] = toff
                toff += (math.prod(s.shape) + ALIGN - 1) // ALIGN * ALIGN
        self.t_total = toff
        self.p = torch.zeros(self.total, dtype=torch.float32)
        self._tr_tables = {}
        self.zero_epoch = 0              # bumped whenever a backward zero-fills gradients; gb_epoch: block prefix -> epoch of its last backward
        self.gb_epoch = {}
        self.g = None
        self.pb = None
        self.wt = None

    # ----- host / device management -------------------------------------------------------------
    def load_state(self, state):
        with torch.no_grad():
            for name, t in state.items():
                if name in self.offset:
                    self.view(name).copy_(t.to(self.p.device, torch.float32))

    def to(self, device):
        self.p = self.p.to(device)
        self._tr_tables = {}
        if self.p.is_cuda:
            self.g = torch.zeros(self.live_end, dtype=torch.float32, device=device) if self.with_grads else None
            self.pb = torch.zeros(self.live_end, dtype=torch.bfloat16, device=device)
            self.wt = torch.zeros(self.t_total, dtype=torch.bfloat16, device=device)
        else:
            self.g = self.pb = self.wt = None
        return self

    def ensure_grads(self):
        """Allocate the flat gradient buffer on the arena's device (CPU arenas get one lazily: only the host-side
        DP plumbing tests need it there)."""
        if self.g is None:
            self.g = torch.zeros(self.live_end, dtype=torch.float32, device=self.p.device)
        return self.g

    # ----- views ---------------------------------------------------------------------------------
    def _v(self, flat, name, two_d=False):
        s = self.info[name]
        n = math.prod(s.shape)
        o = self.offset[name]
        t = flat[o:o + n]
        if two_d:
            return t.view(s.shape[0], n // s.shape[0]) if _is_matrix(s) else t
        return t.view(s.shape)

    def view(self, name):
        return self._v(self.p, name)

    def w(self, name):          # fp32 master, flat or [N,K]
        return self._v(self.p, name, True)

    def gw(self, name):         # gradient view, flat or [N,K]
        return self._v(self.g, name, True)

    def gview(self, name):      # gradient in the parameter's own shape
        return self._v(self.g, name)

    def wb(self, name):         # bf16 shadow [N,K]
        return self._v(self.pb, name, True)

    def wtb(self, name):        # bf16 transposed shadow [K,N]
        s = self.info[name]
        n = math.prod(s.shape)
        o = self.t_offset[name]
        return self.wt[o:o + n].view(n // s.shape[0], s.shape[0])

    # ----- shadows / optimizer -------------------------------------------------------------------
    def _transpose_table(self, which, span=None):
        """Descriptor table (built once per device placement) for the batched transpose of every [N,K] -> [K,N] copy
        in the live range of pass `which` (or in the explicit element range `span`)."""
        key = which if span is None else ("span",) + tuple(span)
        if key not in self._tr_tables:
            lo, hi = span if span is not None else ((self.range[P1][0], self.live_end) if which is None else self.range[which])
            rows, tmap, t0 = [], [], 0
            for i, name in enumerate(n for n in self.t_offset if lo <= self.offset[n] < hi):
                src, dst = self.wb(name), self.wtb(name)
                R, C = src.shape
                assert dst.shape == (C, R) and src.is_contiguous() and dst.is_contiguous()
                tpr, ntl = (C + 63) // 64, ((C + 63) // 64) * ((R + 63) // 64)
                rows.append([src.data_ptr(), dst.data_ptr(), R, C, t0, tpr])
                tmap += [i] * ntl
                t0 += ntl
            dev = self.p.device
            self._tr_tables[key] = (torch.tensor(rows, dtype=torch.int64, device=dev), torch.tensor(tmap, dtype=torch.int32, device=dev), t0)
        return self._tr_tables[key]

    def refresh_shadows(self, which=None, cast=True, span=None):
        """fp32 master -> bf16 shadow (+ transposed copies, ONE batched launch) for the live range of pass `which`
        (None: all), or for the element range `span` of the arena (whole tensors)."""
        from . import ops
        lo, hi = span if span is not None else ((self.range[P1][0], self.live_end) if which is None else self.range[which])
        if cast:
            ops.cast_bf16(self.p[lo:hi], self.pb[lo:hi], hi - lo)
        desc, tmap, ntiles = self._transpose_table(which, span)
        if ntiles:
            ops.transpose_batched(desc, tmap, ntiles)

    def zero_grad_range(self, which, lo=None, hi=None):
        """zero-fill the pass's gradient range (or [lo, hi) of the arena) and open a new gradient EPOCH: Stack.backward allows one backward
        per block and epoch unless accumulate=True (the value third of the qkv bias gradient is derived from the accumulated proj bias
        gradient).  Every zero-fill of gradients goes through here, so direct users of the passes (tools, external loops) that zero with
        this method never trip the guard spuriously."""
        a, b = self.range[which]
        lo = a if lo is None else lo
        hi = b if hi is None else hi
        self.g[lo:hi].zero_()
        self.zero_epoch += 1

    def live_slice(self, flat, which):
        lo, hi = self.range[which]
        return flat[lo:hi]
