"""Host-side schedule of the AVSiam pre-training hot path on the HIP kernels.

MI355X-first restructuring of ``CAVMAE_BASE.forward`` (/root/reference/src/models/cav_mae_base.py:685-741):

* every sequence of a pass is PACKED into one [rows, D] token matrix.  The contrastive pass of the reference
  issues 12 layers x (<=5 audio groups + <=5 video groups) Block calls on ragged groups (:554-558); all ops
  but attention are token-wise and the weights are shared (Siamese), so here it is 12 layers of
  {LN(row-selected affine) -> one QKV GEMM -> varlen attention -> proj GEMM(+residual) -> LN -> fc1(+GELU) -> fc2(+residual)}
  over all rows at once;
* only KEPT tokens are embedded (the reference embeds all and drops 75 %, :448-455,476-477);
* the residual stream stays fp32 (as under the reference's autocast, where pos_embed/LayerNorm keep it fp32),
  GEMM/attention operands are bf16 with fp32 accumulation;
* backward is a hand-written schedule (no autograd graph): dgrad GEMMs read pre-transposed bf16 weight copies,
  wgrad GEMMs contract over the token rows with transposing LDS reads and accumulate straight into the flat
  gradient arena.

The mask plan (which tokens each sample keeps) is an explicit input - see maskplan.py.
"""
import math
import os

import numpy as np
import torch

from . import ops
from .arena import ParamArena
from .config import AVSiamConfig, EngineOptions
from .maskplan import ContrastivePlan, MaePlan, group_sizes, group_ratio, len_keep

BF16, F32, I32, U8 = torch.bfloat16, torch.float32, torch.int32, torch.uint8
LN_EPS_BLOCK, LN_EPS_FINAL = 1e-5, 1e-6       # nn.LayerNorm default in Block; timm ViT final norm


# Per-layer activation recompute (AVSIAM_RECOMPUTE=1 / bench.py --recompute): a Stack keeps only the fp32 input of every block and
# re-runs the block's forward (without its last GEMM) in front of the block's backward.  For shapes whose saved activations do not fit
# 288 GB (ViT-H at batch 64 x 10 frames); costs about 8/12 of the forward GEMMs plus the attention forward again.
# A FRACTION (AVSIAM_RECOMPUTE=0.5 / bench.py --recompute 0.5) recomputes only the first ceil(f x nblocks) blocks of every Stack and
# saves the others' activations as usual: the memory between "everything recomputed" (75 GiB at ViT-H/14, batch 64) and the card's
# 288 GB buys back the same share of the recompute time.
# -> EngineOptions.recompute (config.py): a property of the model, seeded by AVSIAM_RECOMPUTE.


def recompute_blocks(nblocks, setting):
    """number of leading blocks of a Stack of `nblocks` that are recomputed under EngineOptions.recompute ("0" | "1" | a fraction in (0, 1))"""
    f = float(setting)
    if not 0.0 <= f <= 1.0:
        raise ValueError(f"recompute must be 0, 1 or a fraction between them, not {f}")
    return min(nblocks, int(math.ceil(f * nblocks - 1e-9)))
# fp8 forward (AVSIAM_FP8=1 / bench.py --fp8; BASELINE configs[4]'s "fp8 MFMA path", never the default): the four forward GEMMs of a
# block (qkv, proj, fc1, fc2) take OCP e4m3 operands with per-tensor DELAYED scaling and accumulate in fp32.  Every quantised tensor has a
# device record {scale, 1/scale, running amax, saturation events} (ops.Fp8Records): the kernel that produces an operand (the LayerNorm
# in front of qkv / fc1, the attention epilogue in front of proj, fc1's GELU epilogue in front of fc2, a quantising pass for the
# weights) reads the scale from the record and folds the |max| it saw into it; once per forward one tiny kernel turns the amax history
# (16 steps) into the next scales, 448 / (2 * max) - no host synchronisation anywhere, saturation is counted, the state is saved with
# the checkpoint (CAVMAE_BASE.fp8_state).  The first time a GEMM runs its operands are calibrated on the spot (absmax -> scale, still
# on the device) and the activation is quantised by a pass.  The MAE pass's two towers run as one stack with two weight sets here too.
# Everything the backward reads is still produced in bf16.  AVSIAM_FP8=2 (bench.py --fp8 --fp8-dgrad) extends the mode into the backward:
# ALL FOUR input-gradient GEMMs of a block - fc2 (with its GELU' epilogue), fc1, proj and (round 4) qkv - run on e5m2 gradient operands,
# written, beside the bf16 gradient the weight-gradient GEMMs and LayerNorm still read, by the kernels that produce them: the LayerNorm
# backward (residual gradient), the fc2 input-gradient epilogue and the three attention backward kernels (dqkv; avs_attn_bwd_q8 /
# avs_attn_bwd_fused_q8); own records, fmax 57344 - against the e4m3 copy of the transposed weight (the forward's weight scale).
# The weight gradients stay bf16.
# -> EngineOptions.fp8 ("0" | "1" | "2" | "3"): a property of the model (CAVMAE_BASE(fp8_mode=...), --fp8 at the entry point), seeded by AVSIAM_FP8.
# AVSIAM_FP8=3 only.  With the weight gradients on fp8 operands four bf16 tensors of a block have no reader left once their consumers'
# records are calibrated: the two LayerNorm outputs and gelu(x) (read by qkv / fc1 / fc2 and their weight gradients - all in e4m3) and
# the fc2 input gradient (read by fc1's input- and weight-gradient GEMMs in e5m2; fc1's bias gradient is the fused column sum).  "1"
# (default): their producers write the 8-bit copy ONLY (NULL bf16 output) and the three activations live in one shared buffer per stack
# instead of one per block - 2 x D + hidden fewer bf16 values written and kept per token and block, and the GEMM epilogues that wrote
# them (bound by the write burst of all CUs at once) shrink to 3/5 (fc1) and 1/3 (fc2 input gradient) of their bytes.  "0": A/B.
# -> EngineOptions.fp8_lean.
# The residual-GRADIENT stream between the blocks of a stack (AVSIAM_GRAD_STREAM=bf16 | fp32).  bf16 (default): a LayerNorm backward
# reads the upstream residual gradient from the bf16 copy the previous LayerNorm backward wrote for the GEMMs anyway and writes
# only its own bf16 copy - 10 instead of 16 bytes per element and call (the kernel is HBM-bound); the sum itself is formed in fp32
# registers and rounded once per LayerNorm (2 x depth roundings of 2^-9 along a stack: ~0.5 % rms at the bottom of 12 blocks,
# inside the bf16 operand noise the gradients already carry).  A stack's OUTPUT gradient (block 0) is still written in fp32.
# fp32: the round-2 behaviour (fp32 stream beside the bf16 copies).
# -> EngineOptions.grad_stream.


def _z(shape, dtype, dev):
    return torch.zeros(shape, dtype=dtype, device=dev)


class BufferPool:
    """One activation pool for ALL passes of a model (opt-in: CAVMAE_BASE(share_pass_buffers=True) / AVSIAM_SHARE_PASS_BUFFERS=1).
    The training step runs its two passes one after the other (traintest_cavmae_base.py:131-152: pass 1 forward / backward / Adam, then
    pass 2), so pass 1's activations are dead when pass 2 starts - yet each pass's stacks own their buffers for the life of the model,
    and at ViT-H/14, batch 64 x 10 frames, it is the SUM of the two that does not fit the card without recomputing blocks.  With a pool
    every pass bump-allocates the buffers of its stacks from offset 0 of the same chunks of memory: the card holds the LARGER pass.
    Consequences, all handled in Stack: a buffer is zero at allocation only - another pass has written over it since - so the pad rows the
    GEMMs read beyond the token rows are re-zeroed (one launch, ops.ZeroTable) at the start of every Stack.forward; nothing in a pooled
    buffer survives into the other pass (results leave a pass as clones); and a backward over BOTH passes of one forward (the combined
    loss with gradients) is refused by the model."""

    CHUNK = 64 << 20          # smallest chunk; a chunk is 8 x the request that opens it (bump allocation: <= 1/8 lost at its end - a chunk
                              # sized request + 1 GiB was tried in round 5 and wasted 45 % at ViT-H, whose requests are all ~1.2 GiB: each
                              # chunk then holds ONE); when the card cannot give 8 x any more, 2 x and then the request itself are tried
                              # before the allocation fails (a late request must not fail while it would fit)

    def __init__(self, dev):
        self.dev = dev
        self.chunks = []
        self.owner = None                      # the pass whose forward ran last (its activations are what the memory holds)
        self.rewind()

    def rewind(self):
        """the next pass starts over at the first byte of every chunk"""
        self.offs = [0] * len(self.chunks)

    def alloc(self, shape, dtype):
        n = 1
        for d in shape:
            n *= int(d)
        nbytes = (n * torch.empty((), dtype=dtype).element_size() + 255) & ~255
        for i, c in enumerate(self.chunks):                # first fit: a later, smaller request may still use the room a larger one left
            if self.offs[i] + nbytes <= c.numel():
                break
        else:
            chunk = None
            for mult in (8, 2, 1):
                try:
                    chunk = torch.zeros((max(self.CHUNK if mult == 8 else 0, mult * nbytes),), dtype=U8, device=self.dev)
                    break
                except torch.OutOfMemoryError:
                    if mult == 1:
                        raise
            self.chunks.append(chunk)
            self.offs.append(0)
            i, c = len(self.chunks) - 1, self.chunks[-1]
        t = c[self.offs[i]:self.offs[i] + nbytes].view(dtype)[:n].view(shape)
        self.offs[i] += nbytes
        return t

    def nbytes(self):
        return sum(c.numel() for c in self.chunks)

    def used(self):
        """bytes handed out since the last rewind (the live requests of the pass that owns the pool now)"""
        return sum(self.offs)

    def release(self):
        """give the memory back (every engine built on this pool must be dropped by the caller: CAVMAE_BASE.release_buffers)"""
        self.chunks, self.offs, self.owner = [], [], None


class Linear:
    """Arena views of one nn.Linear: bf16 weight [N,K], transposed copy [K,N], fp32 bias, gradient views."""

    def __init__(self, arena: ParamArena, wname, bname, need_t=True):
        self.w = arena.wb(wname)
        self.wt = arena.wtb(wname) if need_t and wname in arena.t_offset else None
        self.b = arena.w(bname)
        grads = arena.g is not None
        self.gw = arena.gw(wname) if grads else None
        self.gb = arena.gw(bname) if grads else None


class Norm:
    def __init__(self, arena: ParamArena, prefix):
        self.g, self.b = arena.w(prefix + ".weight"), arena.w(prefix + ".bias")
        grads = arena.g is not None
        self.dg, self.db = (arena.gw(prefix + ".weight"), arena.gw(prefix + ".bias")) if grads else (None, None)


def grad_ranges(arena, prefix, min_elems=1 << 20):
    """Contiguous ranges of the flat gradient arena covered by the live parameters named `prefix`*, smallest first.
    Only ranges of at least `min_elems` elements are returned: the reducer sends what is left (LayerNorm vectors sitting in
    another liveness class of the arena, embeddings, heads) in its final message instead of one tiny message each."""
    spans = sorted((arena.offset[n], arena.offset[n] + (math.prod(s.shape) + 63) // 64 * 64) for n, s in arena.info.items()
                   if n.startswith(prefix) and s.live)
    out = []
    for a, b in spans:
        if out and a == out[-1][1]:
            out[-1][1] = b
        else:
            out.append([a, b])
    return [(a, b) for a, b in out if b - a >= min_elems]


class BlockParams:
    """One transformer Block (cav_mae_base.py:102-211); `sfx0`/`sfx1` name the LayerNorm sets for row_mod 0/1
    ('' plain, '_a', '_v')."""

    def __init__(self, arena, prefix, sfx0, sfx1=None):
        self.arena, self.prefix = arena, prefix
        self.ranges = grad_ranges(arena, prefix + ".")
        self.n1 = [Norm(arena, f"{prefix}.norm1{sfx0}")] + ([Norm(arena, f"{prefix}.norm1{sfx1}")] if sfx1 is not None else [])
        self.n2 = [Norm(arena, f"{prefix}.norm2{sfx0}")] + ([Norm(arena, f"{prefix}.norm2{sfx1}")] if sfx1 is not None else [])
        self.qkv = Linear(arena, f"{prefix}.attn.qkv.weight", f"{prefix}.attn.qkv.bias")
        self.proj = Linear(arena, f"{prefix}.attn.proj.weight", f"{prefix}.attn.proj.bias")
        self.fc1 = Linear(arena, f"{prefix}.mlp.fc1.weight", f"{prefix}.mlp.fc1.bias")
        self.fc2 = Linear(arena, f"{prefix}.mlp.fc2.weight", f"{prefix}.mlp.fc2.bias")


def _ln_fwd(x, norms, y, mean, rstd, rows, eps, row_mod=None, out_map=None, y8=None, q8_dev=None):
    n0 = norms[0]
    n1 = norms[1] if len(norms) > 1 else None
    ops.layernorm_fwd(x, n0.g, n0.b, y, mean, rstd, rows, eps, n1.g if n1 else None, n1.b if n1 else None,
                      row_mod if n1 else None, out_map, y8=y8, q8_dev=q8_dev)


class _LnBatch:
    """The LayerNorm backwards of one stack backward with their parameter-gradient reduces deferred to ONE launch (ops.LnReduceBatch): the first
    backward allocates a slab workspace per call and collects the table, later ones walk the same calls in the same order."""

    def __init__(self, D, dev):
        self.batch, self.dev, self.k, self.keep = ops.LnReduceBatch(D), dev, 0, None

    def slot(self, rows, n0, n1, dcol):
        """the workspace of the next LayerNorm backward of this stack backward"""
        b = self.batch
        if b.desc is None:
            ws = _z((ops.layernorm_bwd_slabs(rows) * 5 * b.D,), F32, self.dev)
            b.add(ws, rows, n0.dg, n0.db, n1.dg if n1 else None, n1.db if n1 else None, dcol)
        else:
            assert self.k < len(b.entries), "a stack's backward issued more LayerNorm backwards than its first one"
        ws = b.keep[self.k][0]
        self.k += 1
        return ws

    def finish(self):
        b = self.batch
        if b.desc is None:
            b.build(self.dev)
        assert self.k == len(b.entries), "a stack's backward issued fewer LayerNorm backwards than its first one"
        self.k = 0
        b.run()


def _ln_bwd(dy, x, mean, rstd, norms, dx, ws, rows, row_mod=None, out_map=None, dres=None, dx_bf16=None, dcol=None, dx8=None, q8=None, batch=None):
    """dcol: bias gradient of the Linear whose output gradient is the dx produced here (column sum fused in-kernel).
    dx8 / q8: e5m2 copy of dx for an fp8 input-gradient GEMM and its device record.
    batch (_LnBatch): the parameter gradients are reduced by the batch's one launch at the end of the stack's backward, not here."""
    n0 = norms[0]
    n1 = norms[1] if len(norms) > 1 else None
    if batch is not None:
        ops.layernorm_bwd(dy, x, mean, rstd, n0.g, dx, None, None, batch.slot(rows, n0, n1, dcol), rows, n1.g if n1 else None, None, None,
                          row_mod if n1 else None, out_map, dres, dx_bf16, None, dx8=dx8, q8=q8, defer=True)
        return
    ops.layernorm_bwd(dy, x, mean, rstd, n0.g, dx, n0.dg, n0.db, ws, rows, n1.g if n1 else None, n1.dg if n1 else None,
                      n1.db if n1 else None, row_mod if n1 else None, out_map, dres, dx_bf16, dcol, dx8=dx8, q8=q8)


def _dx_in(stack):
    """fp32 buffer for the gradient a stack's backward STARTS from: only the fp32 gradient stream reads it"""
    return None if stack.opts.grad_stream == "bf16" else stack.dx[0]


class _SideStream:
    """Weight-gradient GEMMs on a second HIP stream.  A wgrad only needs its two operands to exist and must finish before the
    backward of the NEXT block overwrites the shared scratch it read; nothing downstream waits for its result until the
    optimizer.  AVSIAM_WGRAD_STREAM selects how far that is used:
      2 (default)  wgrads run beside the attention backward, the LayerNorm backward and the column sums only; every
                   forward/dgrad GEMM waits for them (join), so those GEMMs - the kernel bench.py's roofline times - keep
                   the chip to themselves.  -1.0 % step time.
      1            no such restriction (one event per scratch buffer): wgrads also fill the partly empty last rounds of the
                   forward/dgrad GEMMs.  -2.6 % step time, but a GEMM's launch-to-end time then includes the share of the
                   chip the concurrent wgrad holds (per-launch rate reads 0.29 of peak instead of 0.34).
      0            everything on one stream."""

    def __init__(self, dev):
        self.stream = torch.cuda.Stream(device=dev)
        self.done = {}

    def run(self, key, fn):
        ready = torch.cuda.Event()
        ready.record(torch.cuda.current_stream())
        self.stream.wait_event(ready)
        with torch.cuda.stream(self.stream):
            fn()
            d = torch.cuda.Event()
            d.record(self.stream)
        self.done[key] = d

    def before_write(self, key):
        e = self.done.pop(key, None)
        if e is not None:
            torch.cuda.current_stream().wait_event(e)

    def join(self):
        torch.cuda.current_stream().wait_stream(self.stream)
        self.done.clear()


# (EngineOptions.wgrad_stream / wgrad_group / fp8_gelu8: runtime and structural options of the model, config.py)
_side_streams = {}


def _side_stream(dev):
    """One second stream per device for the whole process (a torch.cuda.Stream per backward call would leak pool entries)."""
    key = (dev.type, dev.index)
    if key not in _side_streams:
        _side_streams[key] = _SideStream(dev)
    s = _side_streams[key]
    s.done.clear()
    return s


class _Inline:
    def run(self, key, fn):
        fn()

    def before_write(self, key):
        pass

    def join(self):
        pass


class Stack:
    """`nblocks` transformer blocks over a packed [rows, D] fp32 residual stream with saved activations."""

    def __init__(self, dev, rows, D, H, hidden, seq_lens, nblocks, row_mod=None, inference=False, pool=None, opts=None, q_rows=0):
        """inference=True: forward only - no activation is kept, every block reuses one set of buffers (the residual stream
        ping-pongs between two) and the backward scratch is not allocated.
        pool (BufferPool): the stack's per-token buffers come from the model's shared activation pool instead of torch.zeros.
        opts (config.EngineOptions): the MODEL's options object, shared by reference - structural fields are read here, runtime fields
        (wgrad_stream, wgrad_group, deterministic) on every backward.
        q_rows > 0 (the decoder, EngineOptions.prune_dead): only the first q_rows rows of every (equal-length) sequence leave the LAST block - the rows
        whose prediction is scored; the others are needed there as keys / values only (cav_mae_base.py:629-635,679-682: loss * mask).  The last
        block then computes LayerNorm-1 and qkv for all rows, attention for q_rows queries per sequence against all keys, and proj / LayerNorm-2 /
        MLP on the COMPACT rows [nseq * q_rows] (sequence s owns rows s * q_rows ..): its att / xmid / ln2 / fc1 / act buffers, x[nblocks] and the
        gradient the backward starts from (dxb[0]) are compact.  `self.lq` says whether the form applies (else 0: plain stack)."""
        assert sum(seq_lens) == rows
        assert pool is None or not inference
        self.pool = pool
        self.opts = opts = opts if opts is not None else EngineOptions.from_env()
        FP8, FP8_LEAN, FP8_GELU8 = opts.fp8, opts.fp8_lean, opts.fp8_gelu8
        self._pads = ops.ZeroTable() if pool is not None else None

        def _z(shape, dtype, dev_):          # (shadows the module's _z for every allocation below)
            if pool is None:
                return torch.zeros(shape, dtype=dtype, device=dev_)
            t = pool.alloc(shape, dtype)
            if len(shape) == 2 and shape[0] > rows:                          # [token rows + pad, columns]: the pad rows must read as zeros
                self._pads.add(t[rows:])
            return t
        self.rows, self.D, self.H, self.hidden, self.nblocks = rows, D, H, hidden, nblocks
        self.lq = 0                             # (set below when the pruned last block applies)
        self.row_mod = row_mod
        self.inference = inference
        self.nrecomp = 0 if inference else recompute_blocks(nblocks, opts.recompute)     # blocks [0, nrecomp) keep no activations (one shared set)
        self.recompute = self.nrecomp > 0
        self.fp8 = FP8 in ("1", "2", "3") and D % 256 == 0 and hidden % 256 == 0 and D >= 256     # the fp8 GEMM's tile constraints (N % 256, K % 128)
        self.fp8_bwd = self.fp8 and FP8 in ("2", "3") and not inference
        # FP8 = 3 (round 4): the four weight gradients of a block on fp8 operands too (ops.gemm_tn_fp8_group: e5m2 gradient copies x
        # e4m3 activation copies).  The e4m3 copy of an activation is then KEPT per block (one more byte per element beside the bf16
        # copy the attention / LayerNorm backward still read) instead of living in one buffer per stack.
        self.fp8_wgrad = self.fp8_bwd and FP8 == "3"
        self.fp8_lean = self.fp8_wgrad and FP8_LEAN          # (see FP8_LEAN)
        if self.fp8:
            r8 = ops.pad_rows(rows, 256)
            self.a8 = _z((r8, max(D, hidden)), U8, dev)      # calibration step only: an activation quantised by a pass
            # persistent e4m3 copies of the stack's weights (two sets: the MAE pass's second tower), re-quantised with the step's scales by
            # ONE batched launch per forward once every GEMM is calibrated (ops.Fp8Batch); per block: qkv | proj | fc1 | fc2
            per_blk = 4 * D * D + 2 * D * hidden
            self.w8_flat = torch.zeros((2 * nblocks * per_blk,), dtype=U8, device=dev)
            self.w8_off = {"qkv": 0, "proj": 3 * D * D, "fc1": 4 * D * D, "fc2": 4 * D * D + D * hidden}
            self.w8_per_blk = per_blk
            self.w8_batch = None                                                   # built after the calibration forward
            def blocks8(cols, shared=None):     # per block when the weight gradients read them (recomputed blocks share one, like their bf16 copies)
                if not self.fp8_wgrad:
                    one = shared if shared is not None else _z((r8, cols), U8, dev)
                    return [one] * nblocks
                one = _z((r8, cols), U8, dev) if self.nrecomp else None
                return [one if i < self.nrecomp else _z((r8, cols), U8, dev) for i in range(nblocks)]
            self.ln1_8 = blocks8(D)                                                # e4m3 copy a LayerNorm writes for qkv ...
            self.ln2_8 = blocks8(D, None if self.fp8_wgrad else self.ln1_8[0])     # ... and for fc1 (one buffer serves both unless they are kept)
            self.att8 = blocks8(D)                                                 # ... the attention epilogue for proj
            self.act8 = blocks8(hidden)                                            # ... and fc1's GELU epilogue for fc2
            self.f8 = ops.Fp8Records(nblocks * 12, dev)                            # per block: 4 GEMMs x (activation, weight, second weight set)
            self.f8_seen = set()                                                   # (block, gemm) whose records hold a calibrated scale
        if self.fp8_bwd:
            self.g8 = ops.Fp8Records(nblocks * 4, dev, fmax=ops.BF8_MAX)           # per block: the e5m2 operands dbo (fc2), dfc1 (fc1), dbm (proj)
            self.g8_seen, self.g8_have = set(), set()
            self.dx8 = [_z((r8, D), U8, dev) for _ in range(2)]      # e5m2 copies of dbo / dbm
            self.dfc1_8 = _z((r8, hidden), U8, dev)
            self.dqkv8 = _z((r8, 3 * D), U8, dev)           # e5m2 copy of dqkv, written by the attention backward kernels
            per_blk_t = 4 * D * D + 2 * D * hidden                                # transposed copies the fp8 input-gradient GEMMs read: fc2 | fc1 | proj | qkv
            self.wt8_flat = torch.zeros((2 * nblocks * per_blk_t,), dtype=U8, device=dev)
            self.wt8_off = {"fc2": 0, "fc1": D * hidden, "proj": 2 * D * hidden, "qkv": 2 * D * hidden + D * D}
            self.wt8_per_blk = per_blk_t
            self.wt8_batch = None
        rp = ops.pad_rows(rows, 128)
        self.rp = rp
        # rows per attention workgroup (4 or 2 waves of 32 queries / keys), by the mean sequence length of the stack - measured on the
        # step's mixes (tools/bench_attn.py --step): the backward of the contrastive pass (39-196 video, 102-512 audio tokens) is 12 %
        # faster with 64-row workgroups, its forward 11 % slower; 49/128-token towers: forward 10 % faster; 618 and 2472 tokens: 128
        mean_len = rows / max(1, len(seq_lens))
        force = int(opts.attn_tile)                                           # A/B switch: 64 | 128 for every stack and direction
        if D // H == 80 and not force:
            force = 128                                                       # hd 80 (ViT-H): the 64-row instantiations exist and lose (round 6, tools/bench_attn.py --huge --tile64:
                                                                              # backward +25 % on the contrastive mix - 232-256 registers, the dK/dV kernel spills)
        self.tiles = ops.AttnTiles(seq_lens, dev, tile_rows=force or (64 if mean_len < 64 else 128))
        # backward: sequences that fit one workgroup (<= 64 / <= 128 tokens: the encoder's video sequences, the MAE towers) take the
        # fused kernel (one read of q, k, v, o, dO and one S / exp evaluation for dq, dk and dv; ops.attn_bwd_fused), the longer ones the
        # two-kernel form.  AVSIAM_ATTN_FUSED=0: everything through the two kernels (A/B).
        self.fused_bwd = []
        if inference:
            self.tiles_bwd = self.tiles
        else:
            # (head dim 80 has the 64-row fused form only - a 128-row workgroup would need 81 KB of LDS: the MAE towers' 64-token frame sequences and the
            #  contrastive pass's 51-token ones; -27 % / -2 % on those stacks' backward attention, tools/bench_attn.py --huge)
            cut = 0 if not opts.attn_fused else {32: 128, 64: 128, 80: 64}.get(D // H, 0)
            if cut:
                self.fused_bwd = [sq for sq in (ops.AttnSeqs(seq_lens, dev, 0, 64), ops.AttnSeqs(seq_lens, dev, 64, cut)) if sq.nseq]
                # round 6: sequences of 129 .. 224 tokens at head dim 64 (the encoder's 156 / 196-token frames, 204-token audio) in one 7-wave workgroup
                # per (sequence, head) whose dS image holds the queries in two halves (attn_bwd_fused224_kernel); not with the e5m2 copy of dqkv
                if D // H == 64 and not self.fp8_bwd and opts.attn_fused224:
                    big = ops.AttnSeqs(seq_lens, dev, 128, 224)
                    if big.nseq:
                        self.fused_bwd.append(big)
                        cut = 224
            long_lens = [L for L in seq_lens if L > cut]
            mean_long = sum(long_lens) / max(1, len(long_lens))
            self.tiles_bwd = ops.AttnTiles(seq_lens, dev, tile_rows=force or (64 if mean_long < 256 else 128), min_len=cut)
        self.q_scale = ops.attn_q_scale(D // H)          # q leaves the qkv GEMM ready for the attention kernels

        def per_block(shape, dtype):
            if inference:                                     # one buffer for all blocks
                return [_z(shape, dtype, dev)] * nblocks
            # recomputed blocks share one buffer, refilled in front of each one's backward; the others keep their own
            one = _z(shape, dtype, dev) if self.nrecomp else None
            return [one if i < self.nrecomp else _z(shape, dtype, dev) for i in range(nblocks)]

        if inference:
            ping = [_z((rp, D), F32, dev) for _ in range(2)]
            self.x = [ping[i & 1] for i in range(nblocks + 1)]
        else:
            self.x = [_z((rp, D), F32, dev) for _ in range(nblocks + 1)]     # x[i] = input of block i; x[-1] = output
        def one_for_all(shape, dtype):                        # fp8 mode 3: a bf16 tensor only the calibration step still writes
            return [_z(shape, dtype, dev)] * nblocks
        lean = getattr(self, "fp8_lean", False)
        self.xmid = per_block((rp, D), F32)
        self.ln1 = (one_for_all if lean else per_block)((rp, D), BF16)
        self.ln2 = self.ln1 if (inference or lean) else per_block((rp, D), BF16)
        self.qkv = per_block((rp, 3 * D), BF16)
        self.att = per_block((rp, D), BF16)
        # gelu'(fc1 output): all the backward needs of the pre-activation (gemm act 1 / 2).  With the fp8 backward (modes 2 / 3) it travels as 8-bit
        # fixed-point codes (ops.gemm_nt_fp8: a uint8 `out` / `aux`): half the bytes of the two epilogues that write and read it (AVSIAM_FP8_GELU8=0: A/B)
        # EngineOptions.gelu8 (default on since round 6): the same codes in the bf16 path
        self.fc1 = per_block((rp, hidden), U8 if ((getattr(self, "fp8_bwd", False) and FP8_GELU8) or (opts.gelu8 and not self.fp8 and not inference)) else BF16)
        self.act = (one_for_all if lean else per_block)((rp, hidden), BF16)
        self.lse = per_block((H, rp), F32)
        nshared = nblocks if inference else self.nrecomp
        st = [_z((rp,), F32, dev) for _ in range(4)] if nshared else None
        self.stats = [st if i < nshared else [_z((rp,), F32, dev) for _ in range(4)] for i in range(nblocks)]   # mean1 rstd1 mean2 rstd2
        if inference:
            return
        # backward scratch (shared by all blocks)
        self.dx = [_z((rp, D), F32, dev) for _ in range(2)]
        self.dxb = [_z((rp, D), BF16, dev) for _ in range(2)]
        self.dfc1 = _z((rp, hidden), BF16, dev)
        self.dln = _z((rp, D), BF16, dev)
        self.datt = _z((rp, D), BF16, dev)
        self.dqkv = _z((rp, 3 * D), BF16, dev)
        self.delta = _z((H, rp), F32, dev)
        self.lnws = _z((ops.layernorm_ws(rows, D),), F32, dev)
        # ---- pruned last block (q_rows): applies to bf16 stacks of equal-length sequences with plain norms, the bf16 gradient stream and the
        # two-kernel attention backward (the decoder: 2 472-token sequences)
        self.lq = 0
        if (q_rows and not self.fp8 and row_mod is None and opts.grad_stream == "bf16" and self.tiles.uniform_len and self.tiles_bwd.uniform_len
                and not self.fused_bwd and 0 < q_rows < seq_lens[0] and nblocks >= 1):
            L, nseq = seq_lens[0], len(seq_lens)
            self.lq, self.Mq = int(q_rows), nseq * int(q_rows)
            c = torch.arange(self.Mq)
            self.q_full = ((c // q_rows) * L + c % q_rows).to(I32).to(dev)                       # compact row -> packed row (residual gather of proj)
            r = torch.arange(rows)
            self.q_compact = torch.where(r % L < q_rows, (r // L) * q_rows + r % L, torch.full_like(r, -1)).to(I32).to(dev)     # packed row -> compact row | -1
            # rows [Mq, next multiple of 128) of the compact activations are read by the weight-gradient GEMMs (whole 64-row stages): they must be
            # zero whatever another block (shared recompute buffers) or another pass (pooled memory) left there
            self._qpads = ops.ZeroTable()
            last = nblocks - 1
            padq = ops.pad_rows(self.Mq, 128)
            if padq > self.Mq:
                for buf in (self.att[last], self.ln2[last], self.act[last]):
                    self._qpads.add(buf[self.Mq:padq])
                self._qpads.build(dev)

    @property
    def out(self):
        return self.x[self.nblocks]

    def forward(self, blocks, blocks2=None, split=0):
        """blocks2 / split: rows [split, rows) run through a SECOND set of blocks (the MAE pass's visual tower next to its
        audio tower, cav_mae_base.py:487,489) in the same launches - every GEMM takes both weight sets
        (ops.gemm_nt(dual=...)), the LayerNorm picks the affine per row (row_mod: 0 below split, 1 from it)."""
        if self._pads is not None:            # pooled buffers: another pass has written over them since this stack last ran
            if self._pads.desc is None and self._pads.entries:
                self._pads.build(self.x[0].device)
            self._pads.run()
        if self.fp8:
            self.f8.update()                   # delayed scaling: last forward's amax -> history -> this forward's scales (one launch)
            nsets = 2 if blocks2 is not None else 1
            if self.w8_batch is None and len(self.f8_seen) == 4 * self.nblocks:      # every GEMM calibrated: freeze the weight table
                self.w8_batch = ops.Fp8Batch(self.f8)
                for i, bp in enumerate(blocks):
                    for name in ("qkv", "proj", "fc1", "fc2"):
                        for st_, bl in enumerate((bp, blocks2[i] if blocks2 is not None else None)[:nsets]):
                            self.w8_batch.add(getattr(bl, name).w, self._w8(i, name, st_), (i * 4 + self._G8[name]) * 3 + 1 + st_)
                self.w8_batch.build(self.x[0].device)
            if self.w8_batch is not None:
                self.w8_batch.run()            # all weights of the stack -> e4m3 with this step's scales (one launch)
        for i, bp in enumerate(blocks):
            self._block_forward(i, bp, blocks2[i] if blocks2 is not None else None, split)

    _G8 = {"qkv": 0, "proj": 1, "fc1": 2, "fc2": 3}

    def _rec(self, i, name, operand=0):
        return self.f8.rec((i * 4 + self._G8[name]) * 3 + operand)

    def _w8(self, i, name, which=0):
        """persistent e4m3 copy [N, K] of block i's weight `name` (which: weight set 0 / 1)"""
        D, Hd = self.D, self.hidden
        N, K = {"qkv": (3 * D, D), "proj": (D, D), "fc1": (Hd, D), "fc2": (D, Hd)}[name]
        o = (which * self.nblocks + i) * self.w8_per_blk + self.w8_off[name]
        return self.w8_flat[o:o + N * K].view(N, K)

    def _wt8(self, i, name, which=0):
        """persistent e4m3 copy of the TRANSPOSED weight (B operand [K_in, N_out] of the input-gradient GEMM)"""
        D, Hd = self.D, self.hidden
        N, K = {"fc2": (Hd, D), "fc1": (D, Hd), "proj": (D, D), "qkv": (D, 3 * D)}[name]
        o = (which * self.nblocks + i) * self.wt8_per_blk + self.wt8_off[name]
        return self.wt8_flat[o:o + N * K].view(N, K)

    def _block_forward(self, i, bp, b2, split, last_gemm=True):
        """Block i: x[i] -> x[i + 1] and everything its backward reads.  last_gemm=False (recompute in front of the backward): x[i + 1]
        is not needed again, the fc2 GEMM is skipped."""
        M = self.rows
        x, st = self.x[i], self.stats[i]
        n1 = bp.n1 if b2 is None else [bp.n1[0], b2.n1[0]]
        n2 = bp.n2 if b2 is None else [bp.n2[0], b2.n2[0]]
        if self.fp8:
            return self._block_forward_fp8(i, bp, b2, split, last_gemm, n1, n2)
        if self.lq and i == self.nblocks - 1:
            # pruned last block (see __init__): every row is a key / value, the first lq rows of a sequence are the queries; from the attention
            # output on the block lives on the compact rows
            assert b2 is None
            Mq = self.Mq
            self._qpads.run()
            _ln_fwd(x, n1, self.ln1[i], st[0], st[1], M, LN_EPS_BLOCK, None)
            ops.gemm_nt(self.ln1[i], bp.qkv.w, self.qkv[i], M, bias=bp.qkv.b, scale_cols=self.D, col_scale=self.q_scale)
            ops.attn_fwd(self.qkv[i], self.tiles, self.H, self.att[i], self.lse[i], lq=self.lq)
            ops.gemm_nt(self.att[i], bp.proj.w, self.xmid[i], Mq, bias=bp.proj.b, res=x, res_idx=self.q_full)
            _ln_fwd(self.xmid[i], n2, self.ln2[i], st[2], st[3], Mq, LN_EPS_BLOCK, None)
            ops.gemm_nt(self.ln2[i], bp.fc1.w, self.fc1[i], Mq, bias=bp.fc1.b, out2=self.act[i], act=1)
            if last_gemm:
                ops.gemm_nt(self.act[i], bp.fc2.w, self.x[i + 1], Mq, bias=bp.fc2.b, res=self.xmid[i])
            return
        dq = dp = d1 = d2 = None
        if b2 is not None:
            dq, dp = (split, b2.qkv.w, b2.qkv.b, None), (split, b2.proj.w, b2.proj.b, None)
            d1, d2 = (split, b2.fc1.w, b2.fc1.b, None), (split, b2.fc2.w, b2.fc2.b, None)
        _ln_fwd(x, n1, self.ln1[i], st[0], st[1], M, LN_EPS_BLOCK, self.row_mod)
        ops.gemm_nt(self.ln1[i], bp.qkv.w, self.qkv[i], M, bias=bp.qkv.b, scale_cols=self.D, col_scale=self.q_scale, dual=dq)
        ops.attn_fwd(self.qkv[i], self.tiles, self.H, self.att[i], self.lse[i])
        ops.gemm_nt(self.att[i], bp.proj.w, self.xmid[i], M, bias=bp.proj.b, res=x, dual=dp)
        _ln_fwd(self.xmid[i], n2, self.ln2[i], st[2], st[3], M, LN_EPS_BLOCK, self.row_mod)
        ops.gemm_nt(self.ln2[i], bp.fc1.w, self.fc1[i], M, bias=bp.fc1.b, out2=self.act[i], act=1, dual=d1)
        if last_gemm:
            ops.gemm_nt(self.act[i], bp.fc2.w, self.x[i + 1], M, bias=bp.fc2.b, res=self.xmid[i], dual=d2)

    def _block_forward_fp8(self, i, bp, b2, split, last_gemm, n1, n2):
        """The block's forward with e4m3 GEMM operands (module comment at FP8).  Once a GEMM's records are calibrated its activation
        arrives in e4m3 from the kernel that produces it; before that (first use) it is quantised by a pass."""
        M = self.rows
        x, st = self.x[i], self.stats[i]
        seen = lambda name: (i, name) in self.f8_seen
        r = lambda name: self._rec(i, name)
        l1, l2, at8, ac8 = self.ln1_8[i], self.ln2_8[i], self.att8[i], self.act8[i]
        lean = self.fp8_lean               # a calibrated consumer reads the e4m3 copy only, and so does its weight gradient: no bf16 output
        _ln_fwd(x, n1, None if lean and seen("qkv") else self.ln1[i], st[0], st[1], M, LN_EPS_BLOCK, self.row_mod, y8=l1 if seen("qkv") else None,
                q8_dev=r("qkv") if seen("qkv") else None)
        self._gemm_fp8(i, "qkv", self.ln1[i], l1, bp.qkv, b2.qkv if b2 else None, split, self.qkv[i], scale_cols=self.D, col_scale=self.q_scale)
        ops.attn_fwd(self.qkv[i], self.tiles, self.H, self.att[i], self.lse[i], **({"out8": at8, "q8": r("proj")} if seen("proj") else {}))
        self._gemm_fp8(i, "proj", self.att[i], at8, bp.proj, b2.proj if b2 else None, split, self.xmid[i], res=x)
        _ln_fwd(self.xmid[i], n2, None if lean and seen("fc1") else self.ln2[i], st[2], st[3], M, LN_EPS_BLOCK, self.row_mod, y8=l2 if seen("fc1") else None,
                q8_dev=r("fc1") if seen("fc1") else None)
        # (the fc2 weight gradient reads the e4m3 copy of gelu(x) too: with fp8 weight gradients it is written even when fc2 itself is skipped)
        o8 = {"out8": ac8, "q8": r("fc2")} if (seen("fc2") and (last_gemm or self.fp8_wgrad)) else {}
        self._gemm_fp8(i, "fc1", self.ln2[i], l2, bp.fc1, b2.fc1 if b2 else None, split, self.fc1[i], out2=None if lean and o8 else self.act[i], act=1, **o8)
        if last_gemm:
            self._gemm_fp8(i, "fc2", self.act[i], ac8, bp.fc2, b2.fc2 if b2 else None, split, self.x[i + 1], res=self.xmid[i])
        elif self.fp8_wgrad and not seen("fc2"):
            raise RuntimeError("fp8 weight gradients: a recomputed block met an uncalibrated fc2 record")

    def _gemm_fp8(self, i, name, A, a8, lin, lin2, split, out, **kw):
        """One forward GEMM on e4m3 operands.  a8: where this GEMM's activation lives in e4m3 once its producer writes it."""
        M = self.rows
        W, W2 = lin.w, (lin2.w if lin2 is not None else None)
        N, K = W.shape
        ra, rw, rw2 = self._rec(i, name, 0), self._rec(i, name, 1), self._rec(i, name, 2)
        first = (i, name) not in self.f8_seen
        if first:                                  # calibrate on the spot, on the device: amax -> record -> scale (no host sync)
            ops.absmax_into(A, ra)
            ops.absmax_into(W, rw)
            if W2 is not None:
                ops.absmax_into(W2, rw2)
            self.f8.update(first=(i * 4 + self._G8[name]) * 3, count=3)
            self.f8_seen.add((i, name))
            if self.fp8_wgrad:                     # the block's own e4m3 copy: the weight gradient of this step reads it
                a8 = a8[:A.shape[0]]
            else:
                a8 = self.a8[:A.shape[0], :K] if self.a8.shape[1] == K else self.a8.view(-1)[:A.shape[0] * K].view(A.shape[0], K)
            ops.quantize_fp8(A, 1.0, out=a8, q=ra)
        w8 = self._w8(i, name, 0)
        w8b = self._w8(i, name, 1) if W2 is not None else None
        if self.w8_batch is None:                  # until the table exists (calibration forward): one quantising pass per weight
            ops.quantize_fp8(W, 1.0, out=w8, q=rw)
            if W2 is not None:
                ops.quantize_fp8(W2, 1.0, out=w8b, q=rw2)
        dual = (split, w8b, lin2.b, rw2) if W2 is not None else None
        ops.gemm_nt_fp8(a8, w8, out, M, bias=lin.b, qa=ra, qw=rw, dual=dual, **kw)

    def _dgrad_fp8(self, i, gname, wname, A, a8, lin, lin2, split, out, colsum2=None, **kw):
        """One input-gradient GEMM on an e5m2 gradient operand (A: its bf16 form, a8: where its e5m2 copy lives - written by the
        producer when (i, gname) is in g8_have, else by a pass here) and the e4m3 copy of the transposed weight, quantised with the
        scale of the forward's weight record (the same tensor)."""
        M = self.rows
        G8 = {"dbo": 0, "dfc1": 1, "dbm": 2, "dqkv": 3}
        idx = i * 4 + G8[gname]
        rec = self.g8.rec(idx)
        if (i, gname) not in self.g8_seen:            # first use: calibrate on the device
            ops.absmax_into(A, rec)
            self.g8.update(first=idx, count=1)
            self.g8_seen.add((i, gname))
        if (i, gname) not in self.g8_have:
            ops.quantize_fp8(A, 1.0, out=a8[:A.shape[0]], q=rec, e5m2=True)
        rw, rw2 = self._rec(i, wname, 1), self._rec(i, wname, 2)
        w8 = self._wt8(i, wname, 0)
        w8b = self._wt8(i, wname, 1) if lin2 is not None else None
        if self.wt8_batch is None:                 # first backward: per-weight passes, and the table for the batched launch is collected
            pend = self.__dict__.setdefault("_wt8_pending", [])
            base = (i * 4 + self._G8[wname]) * 3
            ops.quantize_fp8(lin.wt, 1.0, out=w8, q=rw)
            pend.append((lin.wt, w8, base + 1))
            if lin2 is not None:
                ops.quantize_fp8(lin2.wt, 1.0, out=w8b, q=rw2)
                pend.append((lin2.wt, w8b, base + 2))
        dual = (split, w8b, None, rw2, colsum2) if lin2 is not None else None
        ops.gemm_nt_fp8(a8, w8, out, M, qa=rec, qw=rw, grad=True, dual=dual, **kw)

    def fp8_state(self):
        """delayed-scaling state for the checkpoint (None without the fp8 mode)"""
        if not self.fp8:
            return None
        st = {**self.f8.state(), "seen": sorted(self.f8_seen)}
        if self.fp8_bwd:
            st["grad"] = {**self.g8.state(), "seen": sorted(self.g8_seen)}
        return st

    def load_fp8_state(self, st):
        if self.fp8 and st is not None:
            self.f8.load(st)
            self.f8_seen = {tuple(k) for k in st["seen"]}
            if self.fp8_bwd and "grad" in st:
                self.g8.load(st["grad"])
                self.g8_seen = {tuple(k) for k in st["grad"]["seen"]}

    def backward(self, blocks, last_fc2_bias_done=False, blocks2=None, split=0, reducer=None, accumulate=False):
        """In: d(out) in self.dxb[0] (bf16) - and in self.dx[0] (fp32) when GRAD_STREAM is "fp32".  Out: d(x[0]) in self.dx[0] (fp32)
        and self.dxb[0].
        reducer (comm.GradReducer, data parallel): told which ranges of the gradient arena are final as the blocks complete.
        Bias gradients of fc2 / proj are column sums of the residual-stream gradient and come out of the LayerNorm
        backward that produces it (`dcol`); `last_fc2_bias_done` says the caller's LN backward already did that for
        the last block.
        accumulate: the parameter gradients may already hold another pass's contribution (one backward over both passes of a
        combined forward): quantities derived FROM a parameter gradient then use this backward's increment only.
        blocks2 / split (see forward): the input-gradient GEMMs take both weight sets in one launch; everything that
        reduces over rows into a parameter gradient (weight-gradient GEMMs, bias column sums, LayerNorm backward with its
        gamma/beta/bias sums) runs once per row range on row slices of the same buffers."""
        assert not self.inference, "Stack(inference=True) keeps no activations"
        M = self.rows
        dxo, dxm = self.dx
        dbo, dbm = self.dxb
        ranges = [(0, M, blocks)] if blocks2 is None else [(0, split, blocks), (split, M, blocks2)]
        # a second backward over a block since its gradients were zeroed must say accumulate=True (the value third of the qkv bias gradient is
        # a vector-matrix product of the WHOLE accumulated proj bias gradient) - checked HERE, before any kernel is queued, so a refused call
        # leaves the gradient arena untouched
        for lo_, hi_, bl in ranges:
            for i in range(self.nblocks):
                ar = bl[i].arena
                if not accumulate and ar.gb_epoch.get(bl[i].prefix) == ar.zero_epoch:
                    raise RuntimeError(f"{bl[i].prefix}: second backward over this block since its gradients were zeroed - pass accumulate=True "
                                       "(or zero the gradients with arena.zero_grad_range)")
        for lo_, hi_, bl in ranges:
            for i in range(self.nblocks):
                bl[i].arena.gb_epoch[bl[i].prefix] = bl[i].arena.zero_epoch
        one = blocks2 is not None                # a row range has ONE affine set; the packed single-tower case selects by row_mod
        # recompute: the blocks share ONE set of activation buffers, refilled in front of every block's backward - weight-gradient GEMMs
        # still reading them on a second stream would race with the refill, so everything runs on one stream
        # (a partially recomputed stack - engine.RECOMPUTE a fraction - uses the second stream for the blocks that keep their own buffers and
        #  joins it in front of the first recomputed block, see the loop)
        mode = "0" if (self.nrecomp >= self.nblocks or self.opts.deterministic) else self.opts.wgrad_stream
        if mode == "auto":            # small stacks: the weight gradients also fill the partial rounds of the forward / input-gradient GEMMs (config.EngineOptions)
            mode = "2" if self.rows >= 32768 else "1"
        side = _side_stream(dxo.device) if mode in ("1", "2") else _Inline()
        excl = mode == "2"            # 2: wgrads run beside attention / LayerNorm / column sums only - every nt GEMM waits for them
        grp = excl or mode == "0"     # the block's fc2 / fc1 / proj weight gradients are issued together (one grouped launch); mode 1
                                      # issues each as early as its operands exist
        g16 = self.opts.grad_stream == "bf16"   # the residual gradient travels between the LayerNorm backwards in bf16 only
        det = bool(self.opts.deterministic)     # one stream (above), no epilogue atomics; the library's own reductions follow the "det" knob (the model sets it)
        # the value thirds of the blocks' qkv bias gradients (a [D] x [D, D] product each, below) in ONE launch at the end of the stack's backward instead
        # of one ~10-us launch per block inside it - unless something wants a block's gradients final as it completes (the data-parallel reducer), the
        # gradients accumulate over two passes, or the deterministic forms are asked for
        vm_batch = ln_batch = None
        if reducer is None and not accumulate and not det and self.opts.batch_reduce:
            key = (id(blocks), id(blocks2), split)
            lcache = self.__dict__.setdefault("_ln_batches", {})
            if key not in lcache:
                lcache[key] = (_LnBatch(self.D, dxo.device), blocks, blocks2)
            ln_batch = lcache[key][0]
            ln_batch.k = 0
            cache = self.__dict__.setdefault("_vm_batches", {})
            if key not in cache:
                vb = ops.VecmatBatch()
                for lo_, hi_, bl in ranges:
                    for i in range(self.nblocks):
                        vb.add(bl[i].proj.gb, bl[i].proj.w, bl[i].qkv.gb[2 * self.D:])
                vb.build(dxo.device)
                cache[key] = (vb, blocks, blocks2)                # (the parameter lists are kept alive with their id()s)
            vm_batch = cache[key][0]

        done = set()

        def block_done(j):
            """Every kernel writing block j's parameter gradients is queued on the current stream or joined into it."""
            if j in done:
                return
            done.add(j)
            if reducer is not None:
                for bl in (blocks, blocks2):
                    if bl is not None:
                        for a, b in bl[j].ranges:
                            reducer.ready(a, b)

        def wgrads(blk, key, *jobs, rows=None):
            """rows: the jobs' token rows when they are not the stack's (the pruned last block's compact rows)"""
            def fn():      # the jobs of one call share their token rows: one grouped launch per row range (ops.gemm_tn_group)
                for lo, hi, bl in (ranges if rows is None else [(0, rows, blocks)]):
                    if self.fp8_wgrad:
                        # fp8 mode 3: the e5m2 copy of the gradient (written by its producer or by the input-gradient GEMM's quantising
                        # pass just before) x the block's e4m3 copy of the layer input, with the two operands' device records
                        src = {"fc2": (self.dx8[0], "dbo", self.act8), "fc1": (self.dfc1_8, "dfc1", self.ln2_8), "proj": (self.dx8[1], "dbm", self.att8),
                               "qkv": (self.dqkv8, "dqkv", self.ln1_8)}
                        jobs8 = []
                        for a, b, name in jobs:
                            g8buf, gname, x8 = src[name]
                            assert (blk, gname) in self.g8_seen and (blk, name) in self.f8_seen
                            jobs8.append((g8buf[lo:], x8[blk][lo:], getattr(bl[blk], name).gw, self.g8.rec(blk * 4 + G8[gname]), self._rec(blk, name, 0)))
                        ops.gemm_tn_fp8_group(jobs8, hi - lo)
                        continue
                    trip = [(a[lo:], b[lo:], getattr(bl[blk], name).gw) for a, b, name in jobs]
                    if self.opts.wgrad_group:
                        ops.gemm_tn_group(trip, hi - lo)
                    else:
                        for A, B, C in trip:
                            ops.gemm_tn(A, B, C, hi - lo)
            side.run(key, fn)

        f8b = self.fp8_bwd
        if f8b:
            self.g8.update()                   # delayed scaling of the gradient operands: last backward's amax -> this backward's scales
            self.g8_have = set()               # (block, operand) whose e5m2 copy a producer has written in this backward
            if self.wt8_batch is None and getattr(self, "_wt8_pending", None) is not None and len(self._wt8_pending) == 4 * self.nblocks * (2 if blocks2 is not None else 1):
                self.wt8_batch = ops.Fp8Batch(self.f8)      # the transposed copies use the forward's weight records (the same tensors)
                for src, dst, ridx in self._wt8_pending:
                    self.wt8_batch.add(src, dst, ridx)
                self.wt8_batch.build(dxo.device)
            if self.wt8_batch is not None:
                self.wt8_batch.run()
        G8 = {"dbo": 0, "dfc1": 1, "dbm": 2, "dqkv": 3}

        def g8rec(blk, name):                  # (e5m2 buffer, record) of a gradient operand once calibrated, else (None, None)
            if not f8b or blk < 0 or (blk, name) not in self.g8_seen:
                return None, None
            self.g8_have.add((blk, name))
            return {"dbo": self.dx8[0], "dbm": self.dx8[1], "dfc1": self.dfc1_8, "dqkv": self.dqkv8}[name], self.g8.rec(blk * 4 + G8[name])

        for i in reversed(range(self.nblocks)):
            bp, st = blocks[i], self.stats[i]
            b2 = blocks2[i] if blocks2 is not None else None
            pruned = bool(self.lq) and i == self.nblocks - 1        # the block's upper half (proj ... fc2) lives on the compact rows [0, Mq)
            Mu = self.Mq if pruned else M
            assert not pruned or (blocks2 is None and g16 and not f8b)
            if i < self.nrecomp:
                if not isinstance(side, _Inline):      # from here down the blocks share one set of activation buffers: one stream
                    side.join()
                    if excl and i + 1 < self.nblocks:
                        block_done(i + 1)
                    side, excl, grp = _Inline(), False, True
                self._block_forward(i, bp, b2, split, last_gemm=False)
            # fc2: d(gelu out) fused with GELU' -> d(fc1 pre-activation)
            if excl:
                side.join()
                if i + 1 < self.nblocks:
                    block_done(i + 1)          # its last wgrad (qkv) was the side stream's tail; fc2's bias came from this LN backward
            else:
                side.before_write("dfc1")
            if f8b:
                o8, r8_ = g8rec(i, "dfc1")
                # (lean mode 3: once the e5m2 copy is written by this epilogue nothing reads the bf16 gradient - no bf16 output)
                # (deterministic mode: the bf16 gradient is written even in the lean form - the column-sum kernel below reads it)
                self._dgrad_fp8(i, "dbo", "fc2", dbo, self.dx8[0], bp.fc2, b2.fc2 if b2 else None, split, None if (self.fp8_lean and o8 is not None and not det) else self.dfc1, act=2, aux=self.fc1[i],
                                colsum=None if det else bp.fc1.gb, colsum2=None if det else (b2.fc1.gb if b2 else None), **({"out8": o8, "q8": r8_} if o8 is not None else {}))
            else:
                ops.gemm_nt(dbo, bp.fc2.wt, self.dfc1, Mu, aux=self.fc1[i], act=2, colsum=None if det else bp.fc1.gb,   # + fc1 bias gradient
                            dual=(split, b2.fc2.wt, None, None if det else b2.fc1.gb) if b2 is not None else None)
            if det:                   # deterministic mode: the fc1 bias gradient by the column-sum kernel (one writer per element) instead of the epilogue's atomics
                for lo, hi, bl in (ranges if not pruned else [(0, Mu, blocks)]):
                    ops.colsum(self.dfc1[lo:], bl[i].fc1.gb, hi - lo)
            if not grp:
                wgrads(i, "dbo", (dbo, self.act[i], "fc2"), rows=Mu if pruned else None)
            if i == self.nblocks - 1 and not last_fc2_bias_done:
                for lo, hi, bl in (ranges if not pruned else [(0, Mu, blocks)]):
                    ops.colsum(dbo[lo:], bl[i].fc2.gb, hi - lo)
            # fc1
            if f8b:
                self._dgrad_fp8(i, "dfc1", "fc1", self.dfc1, self.dfc1_8, bp.fc1, b2.fc1 if b2 else None, split, self.dln)
            else:
                ops.gemm_nt(self.dfc1, bp.fc1.wt, self.dln, Mu, dual=(split, b2.fc1.wt, None, None) if b2 is not None else None)
            if not grp:
                wgrads(i, "dfc1", (self.dfc1, self.ln2[i], "fc1"), rows=Mu if pruned else None)
                side.before_write("dbm")
            for lo, hi, bl in (ranges if not pruned else [(0, Mu, blocks)]):
                if accumulate:            # value third of the qkv bias gradient, below: minus what proj.gb holds before this block adds to it
                    ops.vecmat(bl[i].proj.gb, bl[i].proj.w, bl[i].qkv.gb[2 * self.D:], -1.0)
                d8, q8_ = g8rec(i, "dbm")
                _ln_bwd(self.dln[lo:], self.xmid[i][lo:], st[2][lo:], st[3][lo:], bl[i].n2, None if g16 else dxm[lo:], self.lnws, hi - lo,
                        None if one else self.row_mod, dres=(dbo if g16 else dxo)[lo:], dx_bf16=dbm[lo:], dcol=bl[i].proj.gb,
                        dx8=d8[lo:] if d8 is not None else None, q8=q8_, batch=ln_batch)
            # proj
            if f8b:
                self._dgrad_fp8(i, "dbm", "proj", dbm, self.dx8[1], bp.proj, b2.proj if b2 else None, split, self.datt)
            else:
                ops.gemm_nt(dbm, bp.proj.wt, self.datt, Mu, dual=(split, b2.proj.wt, None, None) if b2 is not None else None)
            if grp:                   # the three wgrads whose operands exist now run beside the attention backward
                wgrads(i, "blockA", (dbo, self.act[i], "fc2"), (self.dfc1, self.ln2[i], "fc1"), (dbm, self.att[i], "proj"), rows=Mu if pruned else None)
            else:
                wgrads(i, "dbm", (dbm, self.att[i], "proj"), rows=Mu if pruned else None)
                side.before_write("dqkv")
            a8 = {}
            if f8b:                                # the attention backward kernels write the e5m2 copy of dqkv themselves once its record is calibrated
                d8, q8_ = g8rec(i, "dqkv")
                if d8 is not None:
                    # (lean mode 3: the qkv input- and weight-gradient GEMMs read the e5m2 copy; of the bf16 dqkv only the query third has a reader)
                    a8 = {"dqkv8": d8, "q8": q8_, "kv_bf16": not self.fp8_lean}
            if pruned:
                # queries: the first lq rows of every sequence (out / dO compact); dk, dv for every row.  The query third of the other rows of dqkv
                # is zero by definition (they asked nothing), and the residual gradient of the compact rows goes back into the packed numbering
                # for LayerNorm-1's backward - zeros for the rows that were keys / values only - through the freed dO buffer
                ops.attn_bwd(self.qkv[i], self.tiles_bwd, self.H, self.att[i], self.datt, self.lse[i], self.delta, self.dqkv, lq=self.lq)
                ops.expand_rows(None, self.q_compact, self.dqkv, M, cols=self.D)
                ops.expand_rows(dbm, self.q_compact, self.datt, M)
            elif self.tiles_bwd.ntiles:
                ops.attn_bwd(self.qkv[i], self.tiles_bwd, self.H, self.att[i], self.datt, self.lse[i], self.delta, self.dqkv, **a8)
            for sq in self.fused_bwd:
                ops.attn_bwd_fused(self.qkv[i], sq, self.H, self.att[i], self.datt, self.lse[i], self.dqkv, **a8)
            # qkv
            if excl:
                side.join()
            if f8b:
                self._dgrad_fp8(i, "dqkv", "qkv", self.dqkv, self.dqkv8, bp.qkv, b2.qkv if b2 else None, split, self.dln)
            else:
                ops.gemm_nt(self.dqkv, bp.qkv.wt, self.dln, M, dual=(split, b2.qkv.wt, None, None) if b2 is not None else None)
            wgrads(i, "dqkv", (self.dqkv, self.ln1[i], "qkv"))
            if not excl:
                side.before_write("dbo")
            for lo, hi, bl in ranges:
                # qkv bias gradient: only the query third needs the dqkv matrix.  value third = column sum of dO = (proj bias
                # gradient, complete since this block's LayerNorm-2 backward) . W_proj; key third = 0 exactly (ops.vecmat)
                D = self.D
                ops.colsum(self.dqkv[lo:, :D], bl[i].qkv.gb[:D], hi - lo)
                # (the vector-matrix product reads the WHOLE accumulated proj bias gradient: a second backward over the same block
                #  between two zero-fills must say accumulate=True, or the value third would be counted twice - ADVICE r3)
                #  - checked at the top of this method)
                if vm_batch is None:
                    ops.vecmat(bl[i].proj.gb, bl[i].proj.w, bl[i].qkv.gb[2 * D:])
                d8, q8_ = g8rec(i - 1, "dbo")          # the block below reads this gradient through its fc2 input-gradient GEMM
                _ln_bwd(self.dln[lo:], self.x[i][lo:], st[0][lo:], st[1][lo:], bl[i].n1, None if g16 and i > 0 else dxo[lo:], self.lnws,
                        hi - lo, None if one else self.row_mod, dres=(self.datt if pruned else dbm if g16 else dxm)[lo:], dx_bf16=dbo[lo:],
                        dcol=bl[i - 1].fc2.gb if i > 0 else None, dx8=d8[lo:] if d8 is not None else None, q8=q8_, batch=ln_batch)
        side.join()
        if ln_batch is not None:
            ln_batch.finish()          # (before the value thirds: they read the complete proj bias gradients, which are LayerNorm-2 column sums)
        if vm_batch is not None:
            vm_batch.run()
        for j in reversed(range(self.nblocks)):        # (mode 2 has reported all but block 0 on the way)
            block_done(j)


class PatchEmbedder:
    """Kept-token patch embedding: im2col gather -> GEMM with fused (+bias +pos_embed[token]) * 2 epilogue
    (cav_mae_base.py:444-455: conv, +pos, then `x + norm_pre(x)` with norm_pre = Identity)."""

    def __init__(self, arena, dev, rows, audio, cfg, row_src=None, row_tok=None):
        self.audio, self.rows, self.cfg = audio, rows, cfg
        pre = "vit_base.patch_embed_a" if audio else "vit_base.patch_embed"
        self.lin = Linear(arena, pre + ".proj.weight", pre + ".proj.bias", need_t=False)
        pname = "vit_base.pos_embed_a" if audio else "vit_base.pos_embed"
        D = cfg.embed_dim
        skip = 0 if audio else 1          # the visual table starts with the cls row (pos_embed[:, 1:, :], cav_mae_base.py:453): token t is row 1 + t
        self.pos = arena.view(pname).view(-1, D)[skip:]
        self.gpos = arena.gview(pname).view(-1, D)[skip:] if arena.g is not None else None
        K = cfg.patch * cfg.patch * (1 if audio else cfg.in_chans)
        rp = ops.pad_rows(rows, 128)
        self.cols = _z((rp, K), BF16, dev)
        self.dy = _z((rp, D), BF16, dev)
        # sample (audio) / frame image (video) and token id of each row; may be slices of a buffer the mask-plan kernel fills
        self.row_src = row_src if row_src is not None else _z((rows,), I32, dev)
        self.row_tok = row_tok if row_tok is not None else _z((rows,), I32, dev)      # = row of self.pos / self.gpos

    def set_rows(self, row_src, row_tok):
        self.row_src.copy_(row_src, non_blocking=True)
        self.row_tok.copy_(row_tok, non_blocking=True)

    def forward(self, inp, out, xf=None):
        """inp: audio [B,time,mel] / video [NF,3,H,W] fp32 - or the raw input (un-normalised fbank / uint8 frames) with its
        transform `xf` (ops.InputXf); out: fp32 [rows(+pad), D] slice of the residual stream."""
        if self.audio:
            ops.im2col_audio(inp, self.row_src, self.row_tok, self.cols, self.rows, self.cfg.audio_t, xf, stride=self.cfg.st)
        else:
            ops.im2col_video(inp, self.row_src, self.row_tok, self.cols, self.rows, xf, stride=self.cfg.st)
        ops.gemm_nt(self.cols, self.lin.w, out, self.rows, bias=self.lin.b, res=self.pos, res_idx=self.row_tok, alpha=2.0)

    def backward(self, dx):
        """dx: fp32 [rows, D] gradient of the embedding output."""
        D = self.cfg.embed_dim
        ops.cast_scale(dx, self.dy, self.rows * D, 2.0)
        ops.gemm_tn(self.dy, self.cols, self.lin.gw, self.rows)
        ops.colsum(self.dy, self.lin.gb, self.rows)
        ops.scatter_add_rows(self.dy, self.row_tok, self.gpos, self.rows)


def _fold_frames(imgs, T):
    if imgs.dim() == 4:
        assert T == 1
        return imgs
    assert imgs.shape[1] == T
    return imgs.reshape(imgs.shape[0] * T, *imgs.shape[2:])


# =====================================================================================================
class ContrastivePass:
    """Pass 1 (forward_encoder_mmixed + forward_contrastive, cav_mae_base.py:508-594,641-661)."""

    def __init__(self, arena: ParamArena, cfg: AVSiamConfig, batch, dev, world=1, rank=0, comm=None, pool=None, opts=None):
        self.arena, self.cfg, self.B, self.dev, self.world, self.rank = arena, cfg, batch, dev, world, rank
        self.comm = comm
        self.pool = pool
        self.opts = opts = opts if opts is not None else EngineOptions.from_env()
        if pool is not None:
            pool.rewind()                      # (BufferPool: this pass's stacks start at the first byte, over the other pass's)
        assert world == 1 or comm is not None, "data parallel needs a collective (model.set_distributed)"
        self.dp = comm is not None and getattr(comm, "active", world > 1)
        T, D = cfg.frames, cfg.embed_dim
        sizes = group_sizes(batch, cfg.n_groups)
        self.sizes = sizes
        self.keep_a = [len_keep(cfg.audio_tokens, group_ratio(g)) for g in range(len(sizes))]
        self.keep_v = [len_keep(cfg.video_tokens, group_ratio(g)) for g in range(len(sizes))]
        lens_a = [self.keep_a[g] for g, n in enumerate(sizes) for _ in range(n)]
        lens_v = [self.keep_v[g] for g, n in enumerate(sizes) for _ in range(n * T)]
        self.rows_a, self.rows_v = sum(lens_a), sum(lens_v)
        rows = self.rows_a + self.rows_v
        self.rows = rows
        row_mod = torch.cat([torch.zeros(self.rows_a, dtype=U8), torch.ones(self.rows_v, dtype=U8)]).to(dev)
        self.stack = Stack(dev, rows, D, cfg.num_heads, D * cfg.mlp_ratio, lens_a + lens_v, cfg.depth, row_mod, pool=pool, opts=opts)
        self.blocks = [BlockParams(arena, f"vit_base.blocks.{i}", "_a", "_v") for i in range(cfg.depth)]
        self.final = [Norm(arena, "vit_base.norm_a"), Norm(arena, "vit_base.norm")]
        self.row_src_all, self.row_tok_all = _z((rows,), I32, dev), _z((rows,), I32, dev)
        self.emb_a = PatchEmbedder(arena, dev, self.rows_a, True, cfg, self.row_src_all[:self.rows_a], self.row_tok_all[:self.rows_a])
        self.emb_v = PatchEmbedder(arena, dev, self.rows_v, False, cfg, self.row_src_all[self.rows_a:], self.row_tok_all[self.rows_a:])
        # device-side plan: sequence descriptors (static except src_id = the sample a slot holds this step)
        La, Lv = cfg.audio_tokens, cfg.video_tokens
        nseq = batch + batch * T
        d = np.zeros((nseq, ops.PLAN_FIELDS), dtype=np.int32)
        self.slot_group = np.array([g for g, n in enumerate(sizes) for _ in range(n)], dtype=np.int64)
        off = 0
        for sl in range(batch):
            g = self.slot_group[sl]
            d[sl] = [La, self.keep_a[g], off, 0, -1, 0, cfg.audio_t, sl * La, 0, 0, 0, 0] + ops.PLAN_CLASSIC
            off += self.keep_a[g]
        for sl in range(batch):
            g = self.slot_group[sl]
            for t in range(T):
                d[batch + sl * T + t] = [Lv, self.keep_v[g], off, 0, -1, 0, 0, batch * La + (sl * T + t) * Lv, 0, 0, 0, 0] + ops.PLAN_CLASSIC
                off += self.keep_v[g]
        assert off == rows
        self.desc_host = d
        self.desc_dev = torch.from_numpy(d).to(dev)
        self.bits_host = np.zeros((3, nseq), dtype=np.int32)           # tmask_lo, tmask_hi, fmask per sequence
        self.bits_dev = torch.zeros((3, nseq), dtype=I32, device=dev)
        self.ids_dev = _z((batch * La + batch * T * Lv,), I32, dev)
        self.last_perm = None
        # pooling segments: B audio sequences then B video samples (T frames each), slot order = group-major
        seg = [0]
        for L in lens_a:
            seg.append(seg[-1] + L)
        k = 0
        for g, n in enumerate(sizes):
            for _ in range(n):
                seg.append(seg[-1] + T * self.keep_v[g])
        self.seg_start = torch.tensor(seg, dtype=I32, device=dev)
        self.yf = _z((self.stack.rp, D), F32, dev)
        self.fstat = [_z((self.stack.rp,), F32, dev) for _ in range(2)]
        # slot (group-major position in the packed stack) -> sample: row 0 maps a slot to its row of `reps` ([a samples | v samples]),
        # row 1 to its row of `dAV` ([dA of all W*B samples | dV ...]: this rank's slice) - both folded into the token-mean kernels
        self.slot_maps = _z((2, 2 * batch), I32, dev)
        self.reps = _z((2 * batch, D), F32, dev)
        N = world * batch
        self.N = N
        self.all_reps = _z((world, 2 * batch, D), F32, dev)
        if world == 1:
            self.A, self.V = self.reps[:batch], self.reps[batch:]         # one rank: the embeddings ARE the batch
        else:
            self.A, self.V = _z((N, D), F32, dev), _z((N, D), F32, dev)
        self.An, self.Vn = _z((N, D), F32, dev), _z((N, D), F32, dev)
        self.na, self.nv = _z((N,), F32, dev), _z((N,), F32, dev)
        self.total, self.dtotal = _z((N, N), F32, dev), _z((N, N), F32, dev)
        self.nstats, self.nout = _z((N, 4), F32, dev), _z((3,), F32, dev)
        self.dAn, self.dVn = _z((N, D), F32, dev), _z((N, D), F32, dev)
        self.dAV = _z((2 * N, D), F32, dev)
        self.dA, self.dV = self.dAV[:N], self.dAV[N:]

    # ---- plan -> index arrays (host, O(rows) ints) --------------------------------------------------
    def _set_plan(self, plan: ContrastivePlan):
        B, T = self.B, self.cfg.frames
        assert plan.batch == B
        ag, vg = plan.a_group.numpy(), plan.v_group.numpy()
        counts_a = np.bincount(ag, minlength=len(self.sizes)).tolist()
        counts_v = np.bincount(vg, minlength=len(self.sizes)).tolist()
        assert counts_a[:len(self.sizes)] == self.sizes and counts_v[:len(self.sizes)] == self.sizes, "plan groups do not match torch.chunk sizes"
        order_a = np.argsort(ag, kind="stable")
        order_v = np.argsort(vg, kind="stable")
        src_a, tok_a = [], []
        for s in order_a:
            ids = plan.a_keep[s]
            assert ids.numel() == self.keep_a[ag[s]]
            tok_a.append(ids)
            src_a.append(torch.full((ids.numel(),), int(s), dtype=torch.int64))
        src_v, tok_v = [], []
        for s in order_v:
            for t in range(T):
                ids = plan.v_keep[s][t]
                assert ids.numel() == self.keep_v[vg[s]]
                tok_v.append(ids)
                src_v.append(torch.full((ids.numel(),), int(s) * T + t, dtype=torch.int64))
        self.emb_a.set_rows(torch.cat(src_a).to(I32), torch.cat(tok_a).to(I32))
        self.emb_v.set_rows(torch.cat(src_v).to(I32), torch.cat(tok_v).to(I32))
        self._set_slot_maps(np.concatenate([order_a, B + order_v]))

    def _set_slot_maps(self, s2r):
        """s2r[slot] = row of `reps` (a sample below B, B + v sample from B) the slot's sequence(s) belong to"""
        B, r, N = self.B, self.rank, self.N
        s2r = np.asarray(s2r, dtype=np.int64)
        assert s2r.shape == (2 * B,) and sorted(s2r.tolist()) == list(range(2 * B))
        src = np.where(s2r < B, r * B + s2r, N + r * B + (s2r - B))           # the same sample's row of dAV (this rank's slice)
        self.slot_maps.copy_(torch.from_numpy(np.stack([s2r, src]).astype(np.int32)), non_blocking=True)

    def draw_device(self, seed, nprng, seed_dev=None, host=True):
        """Draw this step's plan on the device (ops.mask_plan): the host only picks the two batch permutations
        (torch.chunk(randperm) at cav_mae_base.py:533-538) and the structured time/frequency selections (:415-422).
        host=False: the host part (draw_host: numpy draws + three small host-to-device copies) has been done by the caller - a step
        replayed from a captured graph (graph_step) does it in front of every replay; seed_dev: the Philox key in device memory."""
        if host:
            self.draw_host(nprng)
        ops.mask_plan(self.desc_dev, self.desc_host, seed, self.row_src_all, self.row_tok_all, self.bits_dev[0], self.bits_dev[1], self.bits_dev[2],
                      ids_out=self.ids_dev, seed_dev=seed_dev)

    def draw_host(self, nprng):
        cfg, B, T = self.cfg, self.B, self.cfg.frames
        t, f = cfg.audio_t, cfg.audio_f
        perm_a, perm_v = nprng.permutation(B), nprng.permutation(B)
        d = self.desc_host
        d[:B, 3] = perm_a
        d[B:, 3] = (perm_v[:, None] * T + np.arange(T)[None, :]).reshape(-1)
        ratios = np.array([group_ratio(int(g)) for g in self.slot_group])
        nt = np.array([int(t * r * 0.7) for r in ratios])
        nf = np.array([int(f * r * 0.7) for r in ratios])
        # random.sample(range(t), n): the n smallest of t random keys
        rank_t = np.argsort(np.argsort(nprng.random((B, t)), axis=1), axis=1)
        rank_f = np.argsort(np.argsort(nprng.random((B, f)), axis=1), axis=1)
        assert t <= 96 and f <= 32, "structured masks are bit sets: at most 96 time and 32 frequency patches"
        tm = (rank_t < nt[:, None]).astype(np.uint64)
        fm = (rank_f < nf[:, None]).astype(np.uint64)
        tbits = (tm[:, :64] << np.arange(min(t, 64), dtype=np.uint64)[None, :]).sum(axis=1)
        fbits = (fm << np.arange(f, dtype=np.uint64)[None, :]).sum(axis=1)
        self.bits_host[:] = 0
        self.bits_host[0, :B] = (tbits & np.uint64(0xFFFFFFFF)).astype(np.uint32).view(np.int32)
        self.bits_host[1, :B] = (tbits >> np.uint64(32)).astype(np.uint32).view(np.int32)
        self.bits_host[2, :B] = fbits.astype(np.uint32).view(np.int32)
        if t > 64:                                               # time patches 64.. travel in the descriptor (PlanSeq.tmask_x)
            d[:B, 9] = (tm[:, 64:] << np.arange(t - 64, dtype=np.uint64)[None, :]).sum(axis=1).astype(np.uint32).view(np.int32)
        self.desc_dev.copy_(torch.from_numpy(d), non_blocking=True)
        self.bits_dev.copy_(torch.from_numpy(self.bits_host), non_blocking=True)
        self._set_slot_maps(np.concatenate([perm_a, B + perm_v]))
        self.last_perm = (perm_a, perm_v)

    def last_plan(self):
        """Rebuild the ContrastivePlan the device drew (tests / debugging; synchronises)."""
        cfg, B, T = self.cfg, self.B, self.cfg.frames
        La, Lv = cfg.audio_tokens, cfg.video_tokens
        ids = self.ids_dev.cpu().long()
        perm_a, perm_v = self.last_perm
        a_group, v_group = torch.zeros(B, dtype=torch.int64), torch.zeros(B, dtype=torch.int64)
        a_keep, v_keep = [None] * B, [None] * B
        for sl in range(B):
            g = int(self.slot_group[sl])
            a_group[perm_a[sl]] = g
            v_group[perm_v[sl]] = g
            a_keep[int(perm_a[sl])] = ids[sl * La: sl * La + self.keep_a[g]].clone()
            base = B * La + sl * T * Lv
            v_keep[int(perm_v[sl])] = [ids[base + t * Lv: base + t * Lv + self.keep_v[g]].clone() for t in range(T)]
        return ContrastivePlan(a_group, v_group, a_keep, v_keep)

    def forward(self, audio, imgs, plan, weight=1.0, xf=(None, None)):
        """-> (weight * nce [1] device tensor, c_acc [1] device tensor); leaves everything backward needs in place.
        plan: a ContrastivePlan to inject, or None when draw_device() already filled the index arrays.
        xf: (audio, frames) input transforms when the inputs are raw (ops.InputXf), else None."""
        cfg, st = self.cfg, self.stack
        if self.pool is not None:
            self.pool.owner = self             # (whose activations the shared memory holds: checked by backward)
        if plan is not None:
            self._set_plan(plan)
        x0 = st.x[0]
        self.emb_a.forward(audio, x0[:self.rows_a], xf[0])
        self.emb_v.forward(_fold_frames(imgs, cfg.frames), x0[self.rows_a:], xf[1])
        st.forward(self.blocks)
        _ln_fwd(st.out, self.final, self.yf, self.fstat[0], self.fstat[1], self.rows, LN_EPS_FINAL, st.row_mod)
        B, W, D = self.B, self.world, cfg.embed_dim
        ops.segment_mean_fwd(self.yf, self.seg_start, self.reps, 2 * B, row_map=self.slot_maps[0], max_row=2 * B)   # slot -> sample order
        if self.dp:
            self.comm.all_gather(self.all_reps.view(W * 2 * B, D), self.reps)             # c2: one [2,B,D] message per rank (RCCL)
        if W > 1:
            self.A.copy_(self.all_reps[:, :B].reshape(W * B, D))
            self.V.copy_(self.all_reps[:, B:].reshape(W * B, D))
        N = self.N
        ops.l2norm_fwd(self.A, self.An, self.na)
        ops.l2norm_fwd(self.V, self.Vn, self.nv)
        ops.gemm_f32_small(self.An, self.Vn, self.total, N, N, D, (D, 1), (1, D), 1.0 / cfg.temperature)
        ops.infonce_fwd(self.total, self.nstats, self.nout, weight)
        return self.nout[2:3], self.nout[1:2]

    def backward(self, gout, weight, reducer=None, accumulate=False):
        """gout: [1] fp32 device tensor (d loss / d loss_c_weighted); weight = contrast_loss_weight.
        accumulate: the gradient arena already holds the other pass's contribution (Stack.backward)."""
        cfg, st = self.cfg, self.stack
        if self.pool is not None and self.pool.owner is not self:
            raise RuntimeError("shared activation pool: another pass ran its forward since this one's - its activations are gone")
        B, W, D, N = self.B, self.world, cfg.embed_dim, self.N
        ops.infonce_dlogits(self.total, self.nstats, gout, weight, self.dtotal)
        ops.gemm_f32_small(self.dtotal, self.Vn, self.dAn, N, D, N, (N, 1), (D, 1), 1.0 / cfg.temperature)
        ops.gemm_f32_small(self.dtotal, self.An, self.dVn, N, D, N, (1, N), (D, 1), 1.0 / cfg.temperature)
        ops.l2norm_bwd(self.dAn, self.An, self.na, self.dA)
        ops.l2norm_bwd(self.dVn, self.Vn, self.nv, self.dV)
        # GatherLayer.backward all-reduces W identical copies and keeps the own slice (gather_layer.py:35-37):
        # that is W x the own-slice gradient, no collective needed.  The slot's row of this rank's slice of
        # [dA | dV] comes through the row map.
        ops.segment_mean_bwd(self.dAV, self.seg_start, self.yf, 2 * B, float(W), row_map=self.slot_maps[1], max_row=2 * N)
        _ln_bwd(self.yf, st.out, self.fstat[0], self.fstat[1], self.final, _dx_in(st), st.lnws, self.rows, st.row_mod,
                dx_bf16=st.dxb[0], dcol=self.blocks[-1].fc2.gb)
        st.backward(self.blocks, last_fc2_bias_done=True, reducer=reducer, accumulate=accumulate)
        self.emb_a.backward(st.dx[0][:self.rows_a])
        self.emb_v.backward(st.dx[0][self.rows_a:])


# =====================================================================================================
class MaePass:
    """Pass 2 (forward_encoder + mm layers + forward_decoder + forward_mae_loss, cav_mae_base.py:441-504,597-638,
    663-683,694-707)."""

    def __init__(self, arena: ParamArena, cfg: AVSiamConfig, batch, dev, pool=None, opts=None):
        self.arena, self.cfg, self.B, self.dev = arena, cfg, batch, dev
        self.pool = pool
        self.opts = opts = opts if opts is not None else EngineOptions.from_env()
        if pool is not None:
            pool.rewind()
        B, T, D, Dd = batch, cfg.frames, cfg.embed_dim, cfg.dec_dim
        ka, kv, La, Lv = cfg.keep_a, cfg.keep_v, cfg.audio_tokens, cfg.video_tokens
        self.rows_a, self.rows_v = B * ka, B * T * kv
        self.n_enc = ka + T * kv
        self.Ltot = La + T * Lv
        hid = D * cfg.mlp_ratio
        # The audio tower (ast_base blocks, plain norms) and the visual tower (vit_base blocks, '_v' norms) are independent and
        # structurally identical: when the audio rows end on a 256-row tile boundary they run as ONE packed stack whose GEMMs
        # take both weight sets per launch (8 192 audio rows alone fill 37 % of the chip with 256^2 tiles).  Otherwise two stacks.
        self.grouped = self.rows_a % 256 == 0 and opts.group_towers               # (EngineOptions.group_towers False: A/B measurements)
        if self.grouped:
            row_mod = torch.cat([torch.zeros(self.rows_a, dtype=U8), torch.ones(self.rows_v, dtype=U8)]).to(dev)
            self.st_t = Stack(dev, self.rows_a + self.rows_v, D, cfg.num_heads, hid, [ka] * B + [kv] * (B * T), cfg.depth, row_mod, pool=pool, opts=opts)
            self.st_a = self.st_v = None
        else:
            self.st_a = Stack(dev, self.rows_a, D, cfg.num_heads, hid, [ka] * B, cfg.depth, pool=pool, opts=opts)
            self.st_v = Stack(dev, self.rows_v, D, cfg.num_heads, hid, [kv] * (B * T), cfg.depth, pool=pool, opts=opts)
        self.st_mm = Stack(dev, B * self.n_enc, D, cfg.num_heads, hid, [self.n_enc] * B, 2, pool=pool, opts=opts)
        # EngineOptions.prune_dead: the decoder's rows are laid out [scored tokens (mask 1) | kept tokens] per sample and the last decoder block, decoder_norm,
        # the prediction heads and the loss run on the scored rows only (Stack q_rows; exact: unscored rows have zero loss and zero gradient, :679-682)
        self.ma, self.mv = La - ka, Lv - kv                                       # scored (masked) tokens per audio sequence / frame
        self.lq = self.ma + T * self.mv
        self.st_dec = Stack(dev, B * self.Ltot, Dd, cfg.dec_heads, Dd * cfg.mlp_ratio, [self.Ltot] * B, cfg.dec_depth, pool=pool, opts=opts,
                            q_rows=self.lq if opts.prune_dead else 0)
        self.prune = self.st_dec.lq > 0
        self.blk_a = [BlockParams(arena, f"ast_base.blocks.{i}", "") for i in range(cfg.depth)]      # :489
        self.blk_v = [BlockParams(arena, f"vit_base.blocks.{i}", "_v") for i in range(cfg.depth)]   # :487
        self.blk_mm = [BlockParams(arena, "mm_layer_1", "_a"), BlockParams(arena, "mm_layer_2", "_a")]   # :699-700
        self.blk_dec = [BlockParams(arena, f"decoder_blocks.{i}", "") for i in range(cfg.dec_depth)]
        self.fin_a, self.fin_v = [Norm(arena, "ast_base.norm_a")], [Norm(arena, "vit_base.norm")]   # :495,492
        self.dec_norm = [Norm(arena, "decoder_norm")]
        rows_e = self.rows_a + self.rows_v
        self.row_src_all, self.row_tok_all = _z((rows_e,), I32, dev), _z((rows_e,), I32, dev)
        self.emb_a = PatchEmbedder(arena, dev, self.rows_a, True, cfg, self.row_src_all[:self.rows_a], self.row_tok_all[:self.rows_a])
        self.emb_v = PatchEmbedder(arena, dev, self.rows_v, False, cfg, self.row_src_all[self.rows_a:], self.row_tok_all[self.rows_a:])
        self.dec_embed = Linear(arena, "decoder_embed.weight", "decoder_embed.bias")
        self.pred_a = Linear(arena, "decoder_pred_a.weight", "decoder_pred_a.bias")
        self.pred_v = Linear(arena, "decoder_pred_v.weight", "decoder_pred_v.bias")
        self.fstat_a = [_z((ops.pad_rows(self.rows_a, 128),), F32, dev) for _ in range(2)]
        self.fstat_v = [_z((ops.pad_rows(self.rows_v, 128),), F32, dev) for _ in range(2)]
        # static row maps: tower rows -> joint [B, n_enc] layout (torch.cat((ca, cv), dim=1), :503)
        b = torch.arange(B).view(B, 1)
        self.map_a = (b * self.n_enc + torch.arange(ka).view(1, ka)).reshape(-1).to(I32).to(dev)
        bt = torch.arange(B * T).view(B * T, 1)
        self.map_v = ((bt // T) * self.n_enc + ka + (bt % T) * kv + torch.arange(kv).view(1, kv)).reshape(-1).to(I32).to(dev)
        rj = self.st_mm.rp
        self.xj_b = _z((rj, D), BF16, dev)                    # bf16 copy of the mm output (decoder_embed operand)
        self.de = _z((rj, Dd), F32, dev)                      # decoder_embed output
        self.dde = _z((rj, Dd), F32, dev)
        self.dde_b = _z((rj, Dd), BF16, dev)
        # un-shuffle index arrays (src_row depends on the plan; pos/modality are static in the position-ordered layout)
        rows_d = B * self.Ltot
        self.src_row = _z((rows_d,), I32, dev)
        P = cfg.patch * cfg.patch
        ma, mv, lq = self.ma, self.mv, self.lq
        if not self.prune:
            pos = torch.cat([torch.arange(La), La + torch.arange(Lv).repeat(T)]).repeat(B)
            self.pos_row = pos.to(I32).to(dev)
            self.dmod = torch.cat([torch.zeros(La, dtype=U8), torch.ones(T * Lv, dtype=U8)]).repeat(B).to(dev)
            # decoder_norm output, modality-major: [B*La audio rows | pad | B*T*Lv video rows | pad]
            self.na_rows, self.nv_rows = B * La, B * T * Lv
            self.dn_rows_in = rows_d
            l = torch.arange(self.Ltot).view(1, -1)
            bb = torch.arange(B).view(B, 1)
            self.v_off = ops.pad_rows(self.na_rows, 128)
            omap = torch.where(l < La, bb * La + l, self.v_off + bb * (T * Lv) + (l - La))
            self.row_of_pos = self.pred_id = None
        else:
            # grouped layout of a sample's Ltot decoder rows: [scored audio (ma) | scored frame 0 .. T-1 (mv each) | kept audio (ka) | kept frames (kv each)];
            # WHICH token sits in a row comes from the plan (pos_row / row_of_pos / pred_id: csrc/maskplan.hip), the modality of a row is static
            self.pos_row = _z((rows_d,), I32, dev)
            self.row_of_pos = _z((rows_d,), I32, dev)
            self.pred_id = _z((B * ma + B * T * mv,), I32, dev)
            self.dmod = torch.cat([torch.zeros(ma, dtype=U8), torch.ones(T * mv, dtype=U8), torch.zeros(ka, dtype=U8), torch.ones(T * kv, dtype=U8)]).repeat(B).to(dev)
            # decoder_norm runs on the COMPACT rows the last block leaves (sample b: rows b * lq ..) -> modality-major compact predictions
            self.na_rows, self.nv_rows = B * ma, B * T * mv
            self.dn_rows_in = B * lq
            j = torch.arange(lq).view(1, -1)
            bb = torch.arange(B).view(B, 1)
            self.v_off = ops.pad_rows(self.na_rows, 128)
            omap = torch.where(j < ma, bb * ma + j, self.v_off + bb * (T * mv) + (j - ma))
        self.dn_map = omap.reshape(-1).to(I32).to(dev)
        dn_rows = self.v_off + ops.pad_rows(self.nv_rows, 128)
        self.dn = _z((dn_rows, Dd), BF16, dev)
        self.ddn = _z((dn_rows, Dd), BF16, dev)
        self.dn_stat = [_z((self.st_dec.rp,), F32, dev) for _ in range(2)]
        self.p_a = _z((ops.pad_rows(self.na_rows), P), F32, dev)
        self.p_v = _z((ops.pad_rows(self.nv_rows), P * cfg.in_chans), F32, dev)
        self.dp_a = _z((ops.pad_rows(self.na_rows), P), BF16, dev)
        self.dp_v = _z((ops.pad_rows(self.nv_rows), P * cfg.in_chans), BF16, dev)
        self.mask_all = _z((B * La + B * T * Lv,), F32, dev)       # [mask_a | mask_v], filled by the plan kernel
        self.mask_a = self.mask_all[:B * La].view(B, La)
        self.mask_v = self.mask_all[B * La:].view(B, T * Lv)
        # device-side plan: every descriptor field is static for this pass (75 % unstructured on all sequences)
        nseq = B + B * T
        d = np.zeros((nseq, ops.PLAN_FIELDS), dtype=np.int32)
        Lt = self.Ltot
        for b in range(B):
            g = [b * Lt, b * Lt + lq, b * ma, 0] if self.prune else ops.PLAN_CLASSIC       # dec_m_off, dec_k_off, pred_off, pos_base (csrc/maskplan.hip)
            d[b] = [La, ka, b * ka, b, b * Lt, b * self.n_enc, 0, b * La, b * La, 0, 0, 0] + g
            for t in range(T):
                i = b * T + t
                # (pos_base La for every frame: the frames share ONE positional table, decoder_pos_embed_v - as `pos` of the classic layout above)
                g = [b * Lt + ma + t * mv, b * Lt + lq + ka + t * kv, B * ma + i * mv, La] if self.prune else ops.PLAN_CLASSIC
                d[B + i] = [Lv, kv, self.rows_a + i * kv, i, b * Lt + La + t * Lv, b * self.n_enc + ka + t * kv, 0,
                            B * La + i * Lv, B * La + i * Lv, 0, 0, 0] + g
        self.desc_host = d
        self.desc_dev = torch.from_numpy(d).to(dev)
        self.ids_dev = _z((B * La + B * T * Lv,), I32, dev)
        self.rl_a, self.rl_v = _z((self.na_rows,), F32, dev), _z((self.nv_rows,), F32, dev)
        self.pid_a, self.pid_v = (self.pred_id[:B * ma], self.pred_id[B * ma:]) if self.prune else (None, None)
        self.losses = _z((3,), F32, dev)                       # loss_a, loss_v, loss_mae
        self.nmask_a = float(B * (La - ka))
        self.nmask_v = float(B * T * (Lv - kv))
        # parameter views used by the un-shuffle
        a = arena
        self.tok = {k: a.view(k).reshape(-1) for k in ("mask_token", "decoder_pos_embed_a", "decoder_pos_embed_v",
                                                      "decoder_modality_a", "decoder_modality_v")}
        self.gtok = {k: a.gview(k).reshape(-1) for k in self.tok}

    def _set_plan(self, plan: MaePlan):
        cfg, B, T = self.cfg, self.B, self.cfg.frames
        ka, kv, La, Lv = cfg.keep_a, cfg.keep_v, cfg.audio_tokens, cfg.video_tokens
        assert plan.batch == B and plan.ids_keep_a.shape == (B, ka) and plan.ids_keep_v.shape == (B, T, kv)
        self.emb_a.set_rows(torch.arange(B).repeat_interleave(ka).to(I32), plan.ids_keep_a.reshape(-1).to(I32))
        self.emb_v.set_rows(torch.arange(B * T).repeat_interleave(kv).to(I32), plan.ids_keep_v.reshape(-1).to(I32))
        ra, rv = plan.ids_restore_a, plan.ids_restore_v                                  # [B,La], [B,T,Lv]
        b = torch.arange(B).view(B, 1)
        src_a = torch.where(ra < ka, b * self.n_enc + ra, torch.full_like(ra, -1))
        t = torch.arange(T).view(1, T, 1)
        src_v = torch.where(rv < kv, b.view(B, 1, 1) * self.n_enc + ka + t * kv + rv, torch.full_like(rv, -1))
        src = torch.cat([src_a, src_v.reshape(B, T * Lv)], dim=1).reshape(-1).to(I32)            # indexed by POSITION (b, l)
        if self.prune:
            # the grouped decoder layout from an injected plan - what the plan kernel writes on the device (csrc/maskplan.hip): the rank of a token
            # in its sequence's shuffle decides its decoder row
            Lt, ma, mv, lq = self.Ltot, self.ma, self.mv, self.lq
            bL = b * Lt
            row_a = torch.where(ra < ka, bL + lq + ra, bL + (ra - ka))                                                        # [B, La]
            row_v = torch.where(rv < kv, bL.view(B, 1, 1) + lq + ka + t * kv + rv, bL.view(B, 1, 1) + ma + t * mv + (rv - kv))    # [B, T, Lv]
            rop = torch.cat([row_a, row_v.reshape(B, T * Lv)], dim=1).reshape(-1)                                             # position -> decoder row
            pos = torch.cat([torch.arange(La), La + torch.arange(Lv).repeat(T)]).repeat(B)      # which positional row a position takes (frames share one table)
            src_d = torch.empty_like(src)
            src_d[rop] = src
            pos_d = torch.empty(B * Lt, dtype=torch.int64)
            pos_d[rop] = pos
            pid = torch.empty(B * ma + B * T * mv, dtype=torch.int64)
            tok_a = torch.arange(La).view(1, La).expand(B, La)
            sel = ra >= ka
            pid[(b * ma + (ra - ka))[sel]] = (b * La + tok_a)[sel]
            it = (b.view(B, 1, 1) * T + t)                                                                                   # frame image index [B, T, 1]
            tok_v = torch.arange(Lv).view(1, 1, Lv).expand(B, T, Lv)
            selv = rv >= kv
            pid[(B * ma + it * mv + (rv - kv))[selv]] = (B * La + it * Lv + tok_v)[selv]
            self.src_row.copy_(src_d, non_blocking=True)
            self.pos_row.copy_(pos_d.to(I32), non_blocking=True)
            self.row_of_pos.copy_(rop.to(I32), non_blocking=True)
            self.pred_id.copy_(pid.to(I32), non_blocking=True)
        else:
            self.src_row.copy_(src, non_blocking=True)
        self.mask_a.copy_((ra >= ka).float(), non_blocking=True)                         # :385-388
        self.mask_v.copy_((rv >= kv).float().reshape(B, T * Lv), non_blocking=True)

    def draw_device(self, seed, nprng=None, seed_dev=None, host=True):
        """75 % unstructured masks of every audio / video sequence, drawn on the device (one launch; no host part)."""
        ops.mask_plan(self.desc_dev, self.desc_host, seed, self.row_src_all, self.row_tok_all, src_row=self.src_row,
                      mask_out=self.mask_all, ids_out=self.ids_dev, seed_dev=seed_dev,
                      grouped=(self.pos_row, self.row_of_pos, self.pred_id) if self.prune else None)

    def draw_host(self, nprng=None):
        pass

    def last_plan(self):
        """Rebuild the MaePlan the device drew (tests / debugging; synchronises)."""
        cfg, B, T = self.cfg, self.B, self.cfg.frames
        La, Lv, ka, kv = cfg.audio_tokens, cfg.video_tokens, cfg.keep_a, cfg.keep_v
        ids = self.ids_dev.cpu().long()
        sa = ids[:B * La].view(B, La)
        sv = ids[B * La:].view(B, T, Lv)
        return MaePlan(sa[:, :ka].contiguous(), torch.argsort(sa, dim=1), sv[..., :kv].contiguous(), torch.argsort(sv, dim=-1))

    def forward(self, audio, imgs, plan, xf=(None, None)):
        """plan: a MaePlan to inject, or None when draw_device() already filled the index arrays.
        xf: (audio, frames) input transforms when the inputs are raw (ops.InputXf), else None."""
        cfg, B, T = self.cfg, self.B, self.cfg.frames
        La, Lv = cfg.audio_tokens, cfg.video_tokens
        if self.pool is not None:
            self.pool.owner = self
        if plan is not None:
            self._set_plan(plan)
        self.audio, self.imgs, self.xf = audio, _fold_frames(imgs, T), xf
        ra = self.rows_a
        if self.grouped:
            st = self.st_t
            self.emb_a.forward(audio, st.x[0][:ra], xf[0])
            self.emb_v.forward(self.imgs, st.x[0][ra:], xf[1])
            st.forward(self.blk_a, self.blk_v, ra)
            out_a, out_v = st.out[:ra], st.out[ra:]
        else:
            self.emb_a.forward(audio, self.st_a.x[0], xf[0])
            self.emb_v.forward(self.imgs, self.st_v.x[0], xf[1])
            self.st_a.forward(self.blk_a)
            self.st_v.forward(self.blk_v)
            out_a, out_v = self.st_a.out, self.st_v.out
        xj = self.st_mm.x[0]
        _ln_fwd(out_a, self.fin_a, xj, self.fstat_a[0], self.fstat_a[1], self.rows_a, LN_EPS_FINAL, out_map=self.map_a)
        _ln_fwd(out_v, self.fin_v, xj, self.fstat_v[0], self.fstat_v[1], self.rows_v, LN_EPS_FINAL, out_map=self.map_v)
        self.st_mm.forward(self.blk_mm)
        rows_j = B * self.n_enc
        ops.cast_scale(self.st_mm.out, self.xj_b, rows_j * cfg.embed_dim, 1.0)
        ops.gemm_nt(self.xj_b, self.dec_embed.w, self.de, rows_j, bias=self.dec_embed.b)                  # :600
        rows_d = B * self.Ltot
        ops.unshuffle_fwd(self.de, self.src_row, self.pos_row, self.dmod, self.tok["mask_token"], self.tok["decoder_pos_embed_a"],
                          self.tok["decoder_pos_embed_v"], self.tok["decoder_modality_a"], self.tok["decoder_modality_v"],
                          self.st_dec.x[0], rows_d)
        self.st_dec.forward(self.blk_dec)
        # (prune: st_dec.out holds the scored rows only, compact; decoder_norm, the heads and the loss see nothing else)
        _ln_fwd(self.st_dec.out, self.dec_norm, self.dn, self.dn_stat[0], self.dn_stat[1], self.dn_rows_in, LN_EPS_BLOCK, out_map=self.dn_map)
        ops.gemm_nt(self.dn[:self.v_off], self.pred_a.w, self.p_a, self.na_rows, bias=self.pred_a.b)       # :634
        ops.gemm_nt(self.dn[self.v_off:], self.pred_v.w, self.p_v, self.nv_rows, bias=self.pred_v.b)       # :635
        ops.mae_loss_fwd(self.p_a, audio, self.mask_a.view(-1), self.rl_a, self.losses[0:1], True, La, self.nmask_a,
                         total=self.losses[2:3], total_init=True, xf=xf[0], stride=cfg.st, row_id=self.pid_a, id_base=0)
        ops.mae_loss_fwd(self.p_v, self.imgs, self.mask_v.view(-1), self.rl_v, self.losses[1:2], False, Lv, self.nmask_v,
                         total=self.losses[2:3], total_init=False, xf=xf[1], stride=cfg.st, row_id=self.pid_v, id_base=B * La)    # loss_mae = a + v (:707)
        return self.losses[2:3], self.losses[0:1], self.losses[1:2], self.mask_a, self.mask_v

    def predictions(self):
        """(pred_a [B, La, P], pred_v [B, T * Lv, P * C]) of the last forward, in the reference's layout (cav_mae_base.py:634-635).  With
        EngineOptions.prune_dead only the scored rows (mask 1) were computed: the others read NaN.  Tests / debugging (synchronises)."""
        cfg, B, T = self.cfg, self.B, self.cfg.frames
        La, Lv = cfg.audio_tokens, cfg.video_tokens
        pa, pv = self.p_a[:self.na_rows].float().cpu(), self.p_v[:self.nv_rows].float().cpu()
        if not self.prune:
            return pa.view(B, La, -1), pv.view(B, T * Lv, -1)
        ida, idv = self.pid_a.cpu().long(), self.pid_v.cpu().long() - B * La
        fa = torch.full((B * La, pa.shape[1]), float("nan"))
        fv = torch.full((B * T * Lv, pv.shape[1]), float("nan"))
        fa[ida] = pa
        fv[idv] = pv
        return fa.view(B, La, -1), fv.view(B, T * Lv, -1)

    def backward(self, gout, reducer=None, accumulate=False):
        cfg, B, T = self.cfg, self.B, self.cfg.frames
        if self.pool is not None and self.pool.owner is not self:
            raise RuntimeError("shared activation pool: another pass ran its forward since this one's - its activations are gone")
        La, Lv, D, Dd = cfg.audio_tokens, cfg.video_tokens, cfg.embed_dim, cfg.dec_dim
        ops.mae_loss_bwd(self.p_a, self.audio, self.mask_a.view(-1), gout, self.dp_a, True, La, self.nmask_a, xf=self.xf[0], stride=cfg.st,
                         row_id=self.pid_a, id_base=0)
        ops.mae_loss_bwd(self.p_v, self.imgs, self.mask_v.view(-1), gout, self.dp_v, False, Lv, self.nmask_v, xf=self.xf[1], stride=cfg.st,
                         row_id=self.pid_v, id_base=B * La)
        # prediction heads
        ops.gemm_nt(self.dp_a, self.pred_a.wt, self.ddn[:self.v_off], self.na_rows)
        ops.gemm_tn(self.dp_a, self.dn[:self.v_off], self.pred_a.gw, self.na_rows)
        ops.colsum(self.dp_a, self.pred_a.gb, self.na_rows)
        ops.gemm_nt(self.dp_v, self.pred_v.wt, self.ddn[self.v_off:], self.nv_rows)
        ops.gemm_tn(self.dp_v, self.dn[self.v_off:], self.pred_v.gw, self.nv_rows)
        ops.colsum(self.dp_v, self.pred_v.gb, self.nv_rows)
        sd = self.st_dec
        rows_d = B * self.Ltot
        _ln_bwd(self.ddn, sd.out, self.dn_stat[0], self.dn_stat[1], self.dec_norm, _dx_in(sd), sd.lnws, self.dn_rows_in,
                out_map=self.dn_map, dx_bf16=sd.dxb[0], dcol=self.blk_dec[-1].fc2.gb)
        sd.backward(self.blk_dec, last_fc2_bias_done=True, reducer=reducer, accumulate=accumulate)
        g = self.gtok
        ops.unshuffle_bwd(sd.dx[0], self.src_row, B, T, La, Lv, self.dde, g["decoder_pos_embed_a"], g["decoder_pos_embed_v"],
                          g["mask_token"], g["decoder_modality_a"], g["decoder_modality_v"], row_of_pos=self.row_of_pos)
        rows_j = B * self.n_enc
        ops.cast_scale(self.dde, self.dde_b, rows_j * Dd, 1.0)
        sm = self.st_mm
        if self.opts.grad_stream == "bf16":                       # the joint layers' backward starts from the bf16 gradient alone
            ops.gemm_nt(self.dde_b, self.dec_embed.wt, sm.dxb[0], rows_j)
        else:
            ops.gemm_nt(self.dde_b, self.dec_embed.wt, sm.dx[0], rows_j)
            ops.cast_scale(sm.dx[0], sm.dxb[0], rows_j * D, 1.0)
        ops.gemm_tn(self.dde_b, self.xj_b, self.dec_embed.gw, rows_j)
        ops.colsum(self.dde_b, self.dec_embed.gb, rows_j)
        sm.backward(self.blk_mm, reducer=reducer, accumulate=accumulate)
        if self.grouped:
            st, ra = self.st_t, self.rows_a
            for lo, fin, fstat, omap, rows, blks in ((0, self.fin_a, self.fstat_a, self.map_a, self.rows_a, self.blk_a),
                                                     (ra, self.fin_v, self.fstat_v, self.map_v, self.rows_v, self.blk_v)):
                _ln_bwd(sm.dx[0], st.out[lo:], fstat[0], fstat[1], fin, None if self.opts.grad_stream == "bf16" else st.dx[0][lo:], st.lnws, rows,
                        out_map=omap, dx_bf16=st.dxb[0][lo:], dcol=blks[-1].fc2.gb)
            st.backward(self.blk_a, last_fc2_bias_done=True, blocks2=self.blk_v, split=ra, reducer=reducer, accumulate=accumulate)
            self.emb_a.backward(st.dx[0][:ra])
            self.emb_v.backward(st.dx[0][ra:])
            return
        for st, fin, fstat, omap, rows, emb, blks in ((self.st_a, self.fin_a, self.fstat_a, self.map_a, self.rows_a, self.emb_a, self.blk_a),
                                                      (self.st_v, self.fin_v, self.fstat_v, self.map_v, self.rows_v, self.emb_v, self.blk_v)):
            _ln_bwd(sm.dx[0], st.out, fstat[0], fstat[1], fin, _dx_in(st), st.lnws, rows, out_map=omap, dx_bf16=st.dxb[0],
                    dcol=blks[-1].fc2.gb)
            st.backward(blks, last_fc2_bias_done=True, reducer=reducer, accumulate=accumulate)
            emb.backward(st.dx[0])
