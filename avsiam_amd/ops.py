"""Host-side wrappers of the C ABI: shape/dtype/device validation, then one kernel launch on torch's current
stream.  Every function here runs ONLY on the HIP kernels; there is no eager fallback.

Shapes are validated on the host before any launch: a hand-written kernel that faults can reset the GPU.
"""
import ctypes

import torch

from . import _lib

BF16, F32, I32, U8 = torch.bfloat16, torch.float32, torch.int32, torch.uint8


def _chk(t, dtype, name, ndim=None):
    if t is None:
        return
    if not t.is_cuda:
        raise _lib.AvsiamHipError(f"{name}: tensor must live on the GPU")
    if t.dtype != dtype:
        raise _lib.AvsiamHipError(f"{name}: expected {dtype}, got {t.dtype}")
    if not t.is_contiguous():
        raise _lib.AvsiamHipError(f"{name}: tensor must be contiguous")
    if ndim is not None and t.dim() != ndim:
        raise _lib.AvsiamHipError(f"{name}: expected {ndim} dims, got {tuple(t.shape)}")


def _stream():
    return _lib.current_stream()


class KernelProfiler:
    """HIP-event timing of individual launches on torch's current stream (the stream the kernels run on).
    Enabled by bench.py over its timed region: `ops.prof = KernelProfiler()`; read with .summary() after a sync.
    `only`: kind prefixes to time; `stride`: time every stride-th launch of those kinds.  An event pair costs the stream
    ~5 us, so the default bench run samples the dominant kernel only: with a stride co-prime to the launches per step
    every launch position of the step is covered equally often over the timed steps, and the sample mean equals the mean
    over all launches."""

    def __init__(self, only=None, stride=1):
        self.rec = {}
        self.only = tuple(only) if only else None
        self.stride = max(1, int(stride))
        self.seen = 0

    def wants(self, kind):
        if self.only is not None and not kind.startswith(self.only):
            return False
        self.seen += 1
        return (self.seen - 1) % self.stride == 0

    def add(self, kind, e0, e1, work, dispatches=1):
        """work: FLOP (or bytes) of the launch, or a pair (FLOP, algorithmic HBM bytes) for kernels priced against both roofs"""
        w, b = work if isinstance(work, tuple) else (work, 0.0)
        self.rec.setdefault(kind, []).append((e0, e1, w, dispatches, b))

    def summary(self):
        out = {}
        for kind, lst in self.rec.items():
            ms = sum(r[0].elapsed_time(r[1]) for r in lst)
            work = sum(r[2] for r in lst)
            out[kind] = {"launches": len(lst), "dispatches": sum(r[3] for r in lst), "total_ms": ms, "avg_us": 1e3 * ms / len(lst), "work": work,
                         "rate": work / (ms * 1e-3) if ms > 0 else 0.0, "bytes": sum(r[4] for r in lst)}
        return out


prof = None


def _launch(kind, work, cname, *args):
    if prof is None or not prof.wants(kind):
        return _lib.call(cname, *args)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    count = cname.startswith("avs_gemm_nt_bf16")         # a call is one or two kernel dispatches (leftover rows)
    d0 = _lib.load().avs_gemm_nt_dispatches() if count else 0
    e0.record()
    _lib.call(cname, *args)
    e1.record()
    prof.add(kind, e0, e1, work, _lib.load().avs_gemm_nt_dispatches() - d0 if count else 1)


def _call(cname, *args):
    """A kernel launch that is not one of the roofline families: timed (kind = the entry point's name) only when the profiler times
    every kind - bench.py's single-stream pass, where the per-family times must add up to the step."""
    if prof is None or prof.only is not None:
        return _lib.call(cname, *args)
    return timed(cname[4:] if cname.startswith("avs_") else cname, lambda: _lib.call(cname, *args))


def timed(kind, fn):
    """run fn() between two HIP events on the current stream when the profiler wants `kind` (torch-side work of the step, e.g. the
    gradient zero-fills, goes through here so that it shows up in bench.py's per-family sum)"""
    if prof is None or not prof.wants(kind):
        return fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    out = fn()
    e1.record()
    prof.add(kind, e0, e1, 0.0)
    return out


def pad_rows(n, mult=128):
    return (n + mult - 1) // mult * mult


# ---------------------------------------------------------------------------------------------------
def layernorm_ws(rows, D):
    return _lib.load().avs_layernorm_ws_floats(rows, D)


def layernorm_fwd(x, g0, b0, y, mean, rstd, rows, eps, g1=None, b1=None, row_mod=None, out_map=None, y8=None, q8=1.0, q8_dev=None):
    """y may be None beside y8: the e4m3 copy is then the only output (fp8 mode 3: nothing reads the bf16 one)."""
    D = x.shape[1]
    assert y is not None or y8 is not None
    _chk(x, F32, "ln.x", 2); _chk(mean, F32, "ln.mean"); _chk(rstd, F32, "ln.rstd")
    if y is not None:
        _chk(y, y.dtype if y.dtype in (BF16, F32) else BF16, "ln.y", 2)
        assert y.shape[1] == D
    _chk(g0, F32, "ln.g0"); _chk(b0, F32, "ln.b0"); _chk(g1, F32, "ln.g1"); _chk(b1, F32, "ln.b1")
    _chk(row_mod, U8, "ln.row_mod"); _chk(out_map, I32, "ln.out_map")
    assert x.shape[0] >= rows and mean.numel() >= rows and rstd.numel() >= rows
    assert g0.numel() == D and b0.numel() == D
    if row_mod is not None:
        assert row_mod.numel() >= rows and g1 is not None and b1 is not None and g1.numel() == D
    if out_map is not None:
        assert out_map.numel() >= rows
    else:
        assert y is None or y.shape[0] >= rows
    if y8 is not None:
        _chk(y8, U8, "ln.y8", 2)
        assert y8.shape[1] == D and y8.shape[0] >= rows and (y is None or y.dtype == BF16) and out_map is None
    ysz = (y.element_size() if y is not None else 0) + (1 if y8 is not None else 0)
    _launch("layernorm_fwd", float(rows) * D * (4 + ysz), "avs_layernorm_fwd_q8", x, g0, b0, g1, b1, row_mod, out_map, y, 1 if (y is not None and y.dtype == F32) else 0, mean, rstd, rows, D,
              float(eps), y8, float(q8), _qrec(q8_dev), _stream())


def layernorm_bwd_slabs(rows):
    """per-block slabs (5 * D floats each) a backward over `rows` rows leaves in its workspace (LnReduceBatch)"""
    return _lib.load().avs_layernorm_bwd_slabs(int(rows))


class LnReduceBatch:
    """The parameter-gradient reduces of many LayerNorm backwards in ONE launch (avs_layernorm_bwd_reduce_batched): each backward runs with
    defer=True on a workspace of its own and leaves its per-block partial sums there; run() adds them all to their targets."""

    def __init__(self, D):
        self.D, self.entries, self.keep, self.desc = D, [], [], None

    def add(self, ws, rows, dg0, db0, dg1=None, db1=None, dcol=None):
        assert self.desc is None, "table already built"
        slabs = layernorm_bwd_slabs(rows)
        _chk(ws, F32, "lnbatch.ws")
        assert ws.numel() >= slabs * 5 * self.D
        for t in (dg0, db0, dg1, db1, dcol):
            _chk(t, F32, "lnbatch.target")
            assert t is None or t.numel() == self.D
        self.entries.append([ws.data_ptr(), slabs] + [t.data_ptr() if t is not None else 0 for t in (dg0, db0, dg1, db1, dcol)])
        self.keep.append((ws, dg0, db0, dg1, db1, dcol))

    def build(self, dev):
        self.desc = torch.tensor(self.entries, dtype=torch.int64, device=dev)

    def run(self):
        _call("avs_layernorm_bwd_reduce_batched", self.desc, len(self.entries), self.D, _stream())


def layernorm_bwd(dy, x, mean, rstd, g0, dx, dg0, db0, ws, rows, g1=None, dg1=None, db1=None, row_mod=None,
                  out_map=None, dres=None, dx_bf16=None, dcol=None, dx8=None, q8=None, defer=False):
    """dres: fp32, or bf16 (the residual-gradient stream kept in bf16: the previous call's dx_bf16); dx (fp32) may be None when
    dx_bf16 is given.  defer: the parameter gradients (dg*, db*, dcol) are NOT formed here - the per-block partial sums stay in `ws`
    (this call's own, layernorm_bwd_slabs(rows) * 5 * D floats) for an LnReduceBatch holding the same targets."""
    D = x.shape[1]
    _chk(dy, dy.dtype if dy.dtype in (BF16, F32) else BF16, "lnb.dy", 2); _chk(x, F32, "lnb.x", 2); _chk(dx, F32, "lnb.dx", 2)
    _chk(dres, dres.dtype if dres is not None and dres.dtype in (BF16, F32) else F32, "lnb.dres", 2)
    _chk(dx_bf16, BF16, "lnb.dx_bf16", 2); _chk(dcol, F32, "lnb.dcol")
    assert dcol is None or dcol.numel() == D
    assert dx is not None or dx_bf16 is not None
    _chk(dx8, U8, "lnb.dx8", 2)
    assert (dx8 is None) == (q8 is None) and (dx8 is None or (dx8.shape[0] >= rows and dx8.shape[1] == D))
    if dx_bf16 is not None:
        assert dx_bf16.shape[0] >= rows and dx_bf16.shape[1] == D
        assert dres is None or dres.dtype != BF16 or dres.data_ptr() != dx_bf16.data_ptr(), "dx_bf16 must not alias a bf16 dres"
    _chk(ws, F32, "lnb.ws"); _chk(row_mod, U8, "lnb.row_mod"); _chk(out_map, I32, "lnb.out_map")
    for t, n in ((g0, "g0"), (g1, "g1"), (dg0, "dg0"), (db0, "db0"), (dg1, "dg1"), (db1, "db1"), (mean, "mean"), (rstd, "rstd")):
        _chk(t, F32, "lnb." + n)
    assert x.shape[0] >= rows and dy.shape[1] == D and (dx is None or (dx.shape[0] >= rows and dx.shape[1] == D))
    assert ws.numel() >= (layernorm_bwd_slabs(rows) * 5 * D if defer else layernorm_ws(rows, D))
    if defer:
        dg0 = db0 = dg1 = db1 = dcol = None
    if out_map is None:
        assert dy.shape[0] >= rows
    if dres is not None:
        assert dres.shape[0] >= rows and dres.shape[1] == D
    nbytes = dy.element_size() + 4 + (4 if dx is not None else 0) + (dres.element_size() if dres is not None else 0) + (2 if dx_bf16 is not None else 0)
    _launch("layernorm_bwd", float(rows) * D * nbytes,
            "avs_layernorm_bwd", dy, 1 if dy.dtype == F32 else 0, x, mean, rstd, g0, g1, row_mod, out_map, dres,
            1 if dres is not None and dres.dtype == BF16 else 0, dx, dx_bf16, dg0, db0, dg1, db1, dcol, ws, rows, D, dx8, _qrec(q8), _stream())


# ---------------------------------------------------------------------------------------------------
def gemm_nt(A, B, out, M, bias=None, res=None, res_idx=None, aux=None, out2=None, alpha=1.0, act=0, scale_cols=0, col_scale=1.0, colsum=None,
            dual=None):
    """x[M,N] = alpha * (A[M,K] @ B[N,K]^T + bias [* aux] + res[res_idx]); columns [0, scale_cols) also * col_scale.
    act 0: out = x.  act 1 (fc1 + GELU): out = gelu'(x), out2 = gelu(x).  act 2 (fc2 input gradient): aux = the gelu'(x) act 1 saved.
    colsum[n] += sum_m out[m, n] (bf16 output only).
    dual = (m_split, B2, bias2, colsum2): rows [m_split, M) use the second weight set (one launch for two towers)."""
    _chk(A, BF16, "gemm.A", 2); _chk(B, BF16, "gemm.B", 2); _chk(bias, F32, "gemm.bias"); _chk(res, F32, "gemm.res", 2)
    # gelu'(x) as 8-bit fixed-point codes (EngineOptions.gelu8; the fp8 GEMMs' convention, gemm_nt_fp8): a uint8 `out` with act 1 / a uint8 `aux` with act 2
    gp8_out = out.dtype == U8
    gp8_aux = aux is not None and aux.dtype == U8
    assert not gp8_out or act == 1, "a uint8 `out` is the 8-bit gelu'(x) of act 1"
    assert not gp8_aux or act == 2, "a uint8 `aux` is the 8-bit gelu'(x) read by act 2"
    _chk(res_idx, I32, "gemm.res_idx"); _chk(aux, U8 if gp8_aux else BF16, "gemm.aux", 2); _chk(out2, BF16, "gemm.out2", 2)
    if out.dtype not in (BF16, F32, U8) or not out.is_cuda or not out.is_contiguous() or out.dim() != 2:
        raise _lib.AvsiamHipError("gemm.out: need contiguous 2-D bf16/fp32 GPU tensor")
    N, K = B.shape
    assert A.shape[1] == K and A.shape[0] >= M and out.shape[0] >= M and out.shape[1] == N, (A.shape, B.shape, out.shape, M)
    if bias is not None:
        assert bias.numel() == N
    if res is not None:
        assert res.shape[1] == N
        if res_idx is None:
            assert res.shape[0] >= M
        else:
            assert res_idx.numel() >= M
    if colsum is not None:
        _chk(colsum, F32, "gemm.colsum")
        assert colsum.numel() == N and out.dtype == BF16
    if act == 1:
        assert out2 is not None and out2.shape[0] >= M and out2.shape[1] == N
    if act == 2:
        assert aux is not None and aux.shape[0] >= M and aux.shape[1] == N
    args = (A, A.stride(0), B, B.stride(0), M, N, K, bias, res, res.stride(0) if res is not None else 0,
            res_idx, aux, aux.stride(0) if aux is not None else 0, out, out.stride(0), 2 if gp8_out else 1 if out.dtype == F32 else 0, out2,
            out2.stride(0) if out2 is not None else 0, float(alpha), 3 if gp8_aux else act, int(scale_cols), float(col_scale), colsum)
    if dual is None:
        _launch("gemm_nt_act%d" % act, 2.0 * M * N * K, "avs_gemm_nt_bf16", *args, _stream())
    else:
        m_split, B2, bias2, colsum2 = dual
        _chk(B2, BF16, "gemm.B2", 2); _chk(bias2, F32, "gemm.bias2"); _chk(colsum2, F32, "gemm.colsum2")
        assert B2.shape == B.shape and B2.stride(0) == B.stride(0) and 0 < m_split < M and m_split % 256 == 0
        assert (bias2 is None) == (bias is None) and (colsum2 is None) == (colsum is None)
        assert bias2 is None or bias2.numel() == N
        assert colsum2 is None or colsum2.numel() == N
        _launch("gemm_nt_act%d" % act, 2.0 * M * N * K, "avs_gemm_nt_bf16_dual", *args, int(m_split), B2, bias2, colsum2, _stream())


FP8_MAX = 448.0          # largest finite OCP e4m3 value
BF8_MAX = 57344.0        # ... and OCP e5m2 (the gradient operands of the fp8 input-gradient GEMMs)


Q_STRIDE = 1024          # floats per device quantisation record (csrc/common.h AVS_Q_STRIDE): a 256-byte header line (scale, 1 / scale, amax
                         # floor, saturation events) + 15 amax shards, each in a 256-byte line of its own


def _qrec(q):
    """a device quantisation record (fp32 [1024]: scale, 1 / scale, running amax, saturation events | 15 amax shards; csrc/common.h AVS_Q_*) or None"""
    if q is None:
        return None
    if not (q.is_cuda and q.dtype == F32 and q.numel() == Q_STRIDE and q.is_contiguous()):
        raise _lib.AvsiamHipError(f"fp8 record: need a contiguous fp32 GPU tensor of {Q_STRIDE} elements")
    return q


class Fp8Records:
    """Device-resident fp8 quantisation state of `n` tensors with delayed scaling (engine.FP8): records q[n, 4] and an amax history
    ring hist[nhist, n].  update() is one tiny launch and never synchronises: hist[pos] <- amax since the last update,
    scale <- 448 / (margin * max(hist)), amax <- 0, saturation counter += (amax * old scale > 448)."""

    def __init__(self, n, dev, nhist=16, margin=2.0, fmax=FP8_MAX):
        """fmax: largest finite value of the format these tensors are quantised to - 448 (e4m3) or 57344 (e5m2, gradients)"""
        self.n, self.nhist, self.margin, self.pos, self.fmax = n, nhist, margin, 0, float(fmax)
        self.q = torch.zeros((n, Q_STRIDE), dtype=F32, device=dev)
        self.hist = torch.zeros((nhist, n), dtype=F32, device=dev)

    def rec(self, i):
        return self.q[i]

    def amax(self, i):
        """the amax gathered since the last update (synchronises: tests)"""
        return float(torch.cat([self.q[i, 2:3], self.q[i, 64::64]]).max().item())

    def update(self, first=0, count=None):
        """count None: the whole table, and the history ring advances (once per forward).  A sub-range (calibration of tensors seen
        for the first time) is written into the current history slot without advancing."""
        whole = count is None
        count = self.n - first if whole else count
        if count <= 0:
            return
        assert 0 <= first and first + count <= self.n
        _call("avs_fp8_scale_update", self.q, self.hist, self.n, self.nhist, self.pos, float(self.margin), int(first), int(count), self.fmax, _stream())
        if whole and first == 0:
            self.pos = (self.pos + 1) % self.nhist

    def state(self):
        return {"q": self.q.detach().cpu().clone(), "hist": self.hist.detach().cpu().clone(), "pos": self.pos, "margin": self.margin}

    def load(self, st):
        assert tuple(st["q"].shape) == tuple(self.q.shape) and tuple(st["hist"].shape) == tuple(self.hist.shape), "fp8 state of another shape"
        self.q.copy_(st["q"]); self.hist.copy_(st["hist"])
        self.pos, self.margin = int(st["pos"]), float(st["margin"])

    def saturation_events(self):
        """total number of (tensor, update) pairs that saturated under the scale they were quantised with (synchronises: tests / logs)"""
        return float(self.q[:, 3].sum().item())


def absmax(x):
    """max |x| of a fp32 / bf16 GPU tensor, as a python float (synchronises: calibration / tests; a training loop would keep it on the device)"""
    assert x.is_cuda and x.is_contiguous() and x.dtype in (F32, BF16)
    out = torch.zeros(1, device=x.device)
    _call("avs_absmax", x, 1 if x.dtype == F32 else 0, x.numel(), out, _stream())
    return float(out.item())


def absmax_into(x, q):
    """fold max |x| into the running amax of the device record q (no synchronisation)"""
    assert x.is_cuda and x.is_contiguous() and x.dtype in (F32, BF16)
    _call("avs_absmax", x, 1 if x.dtype == F32 else 0, x.numel(), _qrec(q)[2:3], _stream())


def quantize_fp8(x, scale, out=None, q=None, e5m2=False):
    """x (fp32 / bf16, contiguous) -> uint8 tensor holding OCP e4m3 of clamp(x * scale, +-448) (e5m2: +-57344); with a device record q the
    scale is q[0] and max |x| is folded into q[2]"""
    assert x.is_cuda and x.is_contiguous() and x.dtype in (F32, BF16) and x.numel() % 4 == 0
    y = out if out is not None else torch.empty(x.shape, dtype=U8, device=x.device)
    assert y.dtype == U8 and y.numel() == x.numel() and y.is_contiguous()
    _call("avs_quantize_fp8", x, 1 if x.dtype == F32 else 0, y, x.numel(), float(scale), _qrec(q), 1 if e5m2 else 0, _stream())
    return y


class Fp8Batch:
    """Quantise many bf16 tensors (a stack's weights) into their persistent fp8 copies with ONE launch: entries (src bf16, dst u8,
    record index into `records`); build() freezes the table.  The tensors must stay where they are (arena views do)."""

    def __init__(self, records, e5m2=False):
        self.records, self.e5m2 = records, e5m2
        self.entries, self.keep = [], []
        self.desc = self.cmap = None

    def add(self, src, dst, rec_index):
        assert self.desc is None, "table already built"
        _chk(src, BF16, "fp8batch.src"); _chk(dst, U8, "fp8batch.dst")
        assert src.numel() == dst.numel() and src.numel() % 4 == 0 and 0 <= rec_index < self.records.n
        self.entries.append((src.data_ptr(), dst.data_ptr(), src.numel() // 4, int(rec_index)))
        self.keep.append((src, dst))

    def build(self, dev):
        cmap = []
        for d, (_, _, n4, _) in enumerate(self.entries):
            cmap += [[d, g] for g in range(0, n4, 2048)]
        self.desc = torch.tensor(self.entries, dtype=torch.int64, device=dev)
        self.cmap = torch.tensor(cmap, dtype=I32, device=dev)
        self.nchunks = len(cmap)

    def run(self):
        _call("avs_quantize_fp8_batched", self.desc, self.cmap, self.nchunks, self.records.q, 1 if self.e5m2 else 0, _stream())


class ZeroTable:
    """Regions (row slices of contiguous 2-D tensors) zeroed together by one launch (avs_zero_batched)."""

    def __init__(self):
        self.entries, self.keep = [], []
        self.desc = self.cmap = None

    def add(self, t):
        assert self.desc is None, "table already built"
        nbytes = t.numel() * t.element_size()
        if nbytes == 0:
            return
        assert t.is_contiguous() and t.data_ptr() % 16 == 0 and nbytes % 16 == 0, "zero table: 16-byte aligned regions"
        self.entries.append((t.data_ptr(), nbytes // 16))
        self.keep.append(t)

    nchunks = 0

    def build(self, dev):
        cmap = []
        for d, (_, n16) in enumerate(self.entries):
            cmap += [[d, g] for g in range(0, n16, 4096)]
        self.nchunks = len(cmap)
        if self.nchunks:
            self.desc = torch.tensor(self.entries, dtype=torch.int64, device=dev)
            self.cmap = torch.tensor(cmap, dtype=I32, device=dev)

    def run(self):
        if self.nchunks:
            _call("avs_zero_batched", self.desc, self.cmap, self.nchunks, _stream())


def gemm_nt_fp8(A8, B8, out, M, alpha=1.0, bias=None, res=None, out2=None, act=0, scale_cols=0, col_scale=1.0, out8=None, out8_scale=1.0,
                qa=None, qw=None, q8=None, dual=None, grad=False, aux=None, colsum=None):
    """x = alpha * (A8[M, K] @ B8[N, K]^T) + bias (+ res); act 0: out = x; act 1: out = gelu'(x), out2 = gelu(x) (like gemm_nt).
    A8 / B8: uint8 tensors of e4m3 values (quantize_fp8), alpha = 1 / (scale_A * scale_B) - or the device records qa / qw (delayed
    scaling: the kernel reads qa[1] * qw[1]); q8: the record of out8.  dual = (m_split, B8_2, bias2, qw2[, colsum2]): a second weight set.
    grad=True: the input-gradient form - A8 holds e5m2 gradients; act 0 or 2 (aux = saved gelu'(x), colsum = fc1 bias gradient); out8
    then receives e5m2(out) for the next input-gradient GEMM."""
    _chk(A8, U8, "gemm8.A", 2); _chk(B8, U8, "gemm8.B", 2); _chk(bias, F32, "gemm8.bias"); _chk(res, F32, "gemm8.res", 2); _chk(out2, BF16, "gemm8.out2", 2)
    _chk(out8, U8, "gemm8.out8", 2); _chk(colsum, F32, "gemm8.colsum")
    # gelu'(x) as 8-bit fixed-point codes (fp8 backward, round 5): a uint8 `out` with act 1 / a uint8 `aux` with act 2 - the two epilogues that write and read it
    gp8_out = out is not None and out.dtype == U8
    gp8_aux = aux is not None and aux.dtype == U8
    _chk(aux, U8 if gp8_aux else BF16, "gemm8.aux", 2)
    assert not gp8_out or (act == 1 and not grad), "a uint8 `out` is the 8-bit gelu'(x) of act 1"
    assert not gp8_aux or (act == 2 and grad), "a uint8 `aux` is the 8-bit gelu'(x) read by act 2"
    N, K = B8.shape
    # 8-bit-only outputs (fp8 mode 3): out=None in the input-gradient form beside out8; out2=None with act 1 beside out8
    assert out is not None or (grad and out8 is not None)
    assert out is None or (out.dtype in (BF16, F32, U8) and out.dim() == 2 and out.is_contiguous() and out.shape[0] >= M and out.shape[1] == N)
    assert A8.shape[1] == K and A8.shape[0] >= M
    assert (qa is None) == (qw is None)
    assert act in (0, 1, 2) and (act != 2 or (grad and aux is not None and aux.shape[0] >= M and aux.shape[1] == N)) and (act != 1 or not grad)
    assert act != 1 or out2 is not None or out8 is not None
    assert colsum is None or (grad and colsum.numel() == N and (out is None or out.dtype == BF16))
    assert out8 is None or (out8.shape[0] >= M and out8.shape[1] == N)
    m_split, B2, bias2, qw2, colsum2 = (tuple(dual) + (None,))[:5] if dual is not None else (0, None, None, None, None)
    if dual is not None:
        _chk(B2, U8, "gemm8.B2", 2); _chk(bias2, F32, "gemm8.bias2"); _chk(colsum2, F32, "gemm8.colsum2")
        assert B2.shape == B8.shape and B2.stride(0) == B8.stride(0) and 0 < m_split < M and m_split % 256 == 0 and qa is not None and qw2 is not None
        assert (bias2 is None) == (bias is None) and (colsum2 is None) == (colsum is None)
    _launch("gemm_nt_fp8", 2.0 * M * N * K, "avs_gemm_nt_fp8", A8, A8.stride(0), B8, B8.stride(0), M, N, K, bias, res, res.stride(0) if res is not None else 0,
            out, out.stride(0) if out is not None else 0, 2 if gp8_out else 1 if (out is not None and out.dtype == F32) else 0, out2, out2.stride(0) if out2 is not None else 0, float(alpha), int(act), int(scale_cols),
            float(col_scale), out8, out8.stride(0) if out8 is not None else 0, float(out8_scale), _qrec(qa), _qrec(qw), _qrec(q8),
            int(m_split), B2, bias2, _qrec(qw2), 2 if gp8_aux else 1 if grad else 0, aux, aux.stride(0) if aux is not None else 0, colsum, colsum2, _stream())


def gemm_tn(A, B, C, M, splits=0):
    """C[N1,N2] += A[M,N1]^T @ B[M,N2]; A/B zero-padded to a multiple of 64 rows."""
    _chk(A, BF16, "wgrad.A", 2); _chk(B, BF16, "wgrad.B", 2); _chk(C, F32, "wgrad.C")
    N1, N2 = A.shape[1], B.shape[1]
    need = pad_rows(M, 64)
    assert A.shape[0] >= need and B.shape[0] >= need, "wgrad operands must be allocated (zero) to a multiple of 64 rows"
    assert C.numel() == N1 * N2
    _launch("gemm_tn", 2.0 * M * N1 * N2, "avs_gemm_tn_bf16", A, A.stride(0), B, B.stride(0), C, N2, M, N1, N2, splits, _stream())


def gemm_tn_group(jobs, M):
    """jobs: 1-3 triples (A[M,N1], B[M,N2], C[N1*N2]) over the same M token rows: C += A^T @ B, in ONE launch when the shapes allow
    (avs_gemm_tn_bf16_group3: fewer splits of the token rows, i.e. less fp32 atomic traffic, than one launch each)."""
    assert 1 <= len(jobs) <= 3
    if len(jobs) == 1:
        return gemm_tn(jobs[0][0], jobs[0][1], jobs[0][2], M)
    need = pad_rows(M, 64)
    args, flops = [], 0.0
    for A, B, C in jobs:
        _chk(A, BF16, "wgrad.A", 2); _chk(B, BF16, "wgrad.B", 2); _chk(C, F32, "wgrad.C")
        N1, N2 = A.shape[1], B.shape[1]
        assert A.shape[0] >= need and B.shape[0] >= need, "wgrad operands must be allocated (zero) to a multiple of 64 rows"
        assert C.numel() == N1 * N2 and N1 % 128 == 0 and N2 % 128 == 0
        args += [A, A.stride(0), B, B.stride(0), C, N1, N2]
        flops += 2.0 * M * N1 * N2
    args += [None, 0, None, 0, None, 0, 0] * (3 - len(jobs))
    _launch("gemm_tn", flops, "avs_gemm_tn_bf16_group3", *args, M, _stream())


def gemm_tn_fp8_group(jobs, M):
    """jobs: 1-3 tuples (A8 [M,N1] e5m2 gradient copy, B8 [M,N2] e4m3 activation copy, C [N1*N2] fp32, qa, qb) over the same M token rows:
    C += A8^T @ B8 / (scale_a * scale_b) in ONE launch of the fp8 weight-gradient kernel (avs_gemm_tn_fp8_group3; fp8 mode 3)"""
    assert 1 <= len(jobs) <= 3
    need = pad_rows(M, 64)
    args, flops = [], 0.0
    for A, B, C, qa, qb in jobs:
        _chk(A, U8, "wgrad8.A", 2); _chk(B, U8, "wgrad8.B", 2); _chk(C, F32, "wgrad8.C")
        N1, N2 = A.shape[1], B.shape[1]
        assert A.shape[0] >= need and B.shape[0] >= need, "wgrad operands must be allocated (zero) to a multiple of 64 rows"
        assert C.numel() == N1 * N2 and N1 % 256 == 0 and N2 % 256 == 0
        args += [A, A.stride(0), B, B.stride(0), C, N1, N2, _qrec(qa), _qrec(qb)]
        flops += 2.0 * M * N1 * N2
    args += [None, 0, None, 0, None, 0, 0, None, None] * (3 - len(jobs))
    _launch("gemm_tn_fp8", flops, "avs_gemm_tn_fp8_group3", *args, M, _stream())


# ---------------------------------------------------------------------------------------------------
class AttnTiles:
    """(sequence start, length, q0) per tile of `tile_rows` (128 or 64) rows for a packed batch of sequences."""

    def __init__(self, seq_lens, device, start_row=0, tile_rows=None, min_len=0):
        """min_len: only sequences LONGER than this get tiles (the shorter ones go to attn_bwd_fused); rows are still counted"""
        starts, lens, q0s = [], [], []
        row = start_row
        if tile_rows is None:      # 128-row (4-wave) workgroups measured faster than 64-row ones at every length (tools/bench_attn.py)
            tile_rows = 128
        assert tile_rows in (64, 128)
        self.tile_rows = tile_rows
        taken = []
        for L in seq_lens:
            if L > min_len:
                taken.append(L)
                for q0 in range(0, L, tile_rows):
                    starts.append(row); lens.append(L); q0s.append(q0)
            row += L
        self.rows = float(sum(taken))
        self.ntiles = len(starts)
        self.start = torch.tensor(starts, dtype=I32, device=device)
        self.len = torch.tensor(lens, dtype=I32, device=device)
        self.q0 = torch.tensor(q0s, dtype=I32, device=device)
        self.max_row = row
        self.sum_sq = float(sum(L * L for L in taken))           # sum of L^2: attention FLOPs = 4 * sum_sq * D
        # the common length when every sequence has it and all of them have tiles (the pruned last decoder block needs both), else 0
        self.uniform_len = seq_lens[0] if (len(seq_lens) and all(L == seq_lens[0] for L in seq_lens) and len(taken) == len(seq_lens) and start_row == 0) else 0


def attn_q_scale(hd):
    """Factor the q columns of qkv must carry for attn_fwd/attn_bwd: softmax scale hd^-0.5 times log2(e)."""
    return hd ** -0.5 * 1.4426950408889634


def attn_fwd(qkv, tiles, H, out, lse, out8=None, q8=None, lq=0):
    """out8 / q8 (fp8 mode): also write the e4m3 copy of the output for the proj GEMM, scaled by the device record q8.
    lq > 0 (equal-length sequences; avs_attn_fwd_cq): only the first lq rows of every sequence are queries, `out` is compact
    (sequence s owns rows s * lq ..) - the decoder's last block in the pruned form (engine.Stack)."""
    if lq:
        assert out8 is None and tiles.uniform_len and 0 < lq <= tiles.uniform_len
        _chk(qkv, BF16, "attn.qkv", 2); _chk(out, BF16, "attn.out", 2); _chk(lse, F32, "attn.lse", 2)
        D = qkv.shape[1] // 3
        nseq = tiles.max_row // tiles.uniform_len
        assert qkv.shape[1] == 3 * D and out.shape[1] == D and D % H == 0 and D // H in (32, 64, 80)
        assert qkv.shape[0] >= tiles.max_row and out.shape[0] >= nseq * lq and lse.shape[0] == H and lse.shape[1] >= tiles.max_row
        frac = lq / tiles.uniform_len
        _launch("attn_fwd_hd%d" % (D // H), (4.0 * tiles.sum_sq * frac * D, tiles.rows * ((4.0 + 4.0 * frac) * D + 4.0 * H * frac)), "avs_attn_fwd_cq", qkv, qkv.stride(0), D, H,
                tiles.start, tiles.len, tiles.q0, tiles.ntiles, tiles.tile_rows, out, out.stride(0), lse, lse.shape[1], int(lq), _stream())
        return
    _chk(qkv, BF16, "attn.qkv", 2); _chk(out, BF16, "attn.out", 2); _chk(lse, F32, "attn.lse", 2); _chk(out8, U8, "attn.out8", 2)
    assert (out8 is None) == (q8 is None) and (out8 is None or (out8.shape[0] >= tiles.max_row and out8.shape[1] == out.shape[1]))
    D = qkv.shape[1] // 3
    assert qkv.shape[1] == 3 * D and out.shape[1] == D and D % H == 0 and D // H in (32, 64, 80)
    assert qkv.shape[0] >= tiles.max_row and out.shape[0] >= tiles.max_row
    assert lse.shape[0] == H and lse.shape[1] >= tiles.max_row
    # algorithmic HBM bytes: q, k, v read and o written once per row (bf16), lse written per head and row
    _launch("attn_fwd_hd%d" % (D // H), (4.0 * tiles.sum_sq * D, tiles.rows * (8.0 * D + 4.0 * H)), "avs_attn_fwd_q8", qkv, qkv.stride(0), D, H, tiles.start, tiles.len, tiles.q0, tiles.ntiles, tiles.tile_rows, out, out.stride(0),
            lse, lse.shape[1], out8, out8.stride(0) if out8 is not None else 0, _qrec(q8), _stream())


def attn_bwd(qkv, tiles, H, out, dout, lse, delta, dqkv, dqkv8=None, q8=None, kv_bf16=True, lq=0):
    """dqkv8 / q8 (fp8 backward): also write the e5m2 copy of dqkv - the gradient operand of the fp8 qkv input-gradient GEMM - scaled by
    the device record q8, whose running amax takes the largest |dqkv| written.  kv_bf16=False (with dqkv8): the key / value thirds of
    the bf16 dqkv are not written.
    lq > 0 (avs_attn_bwd_cq): the backward of attn_fwd(lq=): out / dout compact, dq written for the first lq rows of every sequence only."""
    assert kv_bf16 or dqkv8 is not None
    if lq:
        assert dqkv8 is None and tiles.uniform_len and 0 < lq <= tiles.uniform_len
        _chk(qkv, BF16, "attnb.qkv", 2); _chk(out, BF16, "attnb.out", 2); _chk(dout, BF16, "attnb.dout", 2)
        _chk(lse, F32, "attnb.lse", 2); _chk(delta, F32, "attnb.delta", 2); _chk(dqkv, BF16, "attnb.dqkv", 2)
        D = qkv.shape[1] // 3
        nseq = tiles.max_row // tiles.uniform_len
        assert dqkv.shape == qkv.shape and out.shape[1] == D and dout.shape == out.shape and delta.shape == lse.shape and D // H in (32, 64, 80)
        assert qkv.shape[0] >= tiles.max_row and out.shape[0] >= nseq * lq and lse.shape[0] == H and lse.shape[1] >= tiles.max_row
        frac = lq / tiles.uniform_len
        _launch("attn_bwd_hd%d" % (D // H), (8.0 * tiles.sum_sq * frac * D, tiles.rows * ((16.0 + 8.0 * frac) * D + 16.0 * H * frac)), "avs_attn_bwd_cq", qkv, qkv.stride(0), D, H,
                tiles.start, tiles.len, tiles.q0, tiles.ntiles, tiles.tile_rows, out, dout, out.stride(0), lse, delta, lse.shape[1], dqkv, int(lq), _stream())
        return
    _chk(qkv, BF16, "attnb.qkv", 2); _chk(out, BF16, "attnb.out", 2); _chk(dout, BF16, "attnb.dout", 2); _chk(dqkv8, U8, "attnb.dqkv8", 2)
    assert (dqkv8 is None) == (q8 is None) and (dqkv8 is None or (dqkv8.shape[0] >= tiles.max_row and dqkv8.shape[1] == qkv.shape[1]))
    _chk(lse, F32, "attnb.lse", 2); _chk(delta, F32, "attnb.delta", 2); _chk(dqkv, BF16, "attnb.dqkv", 2)
    D = qkv.shape[1] // 3
    assert dqkv.shape == qkv.shape and out.shape[1] == D and dout.shape == out.shape and delta.shape == lse.shape and D // H in (32, 64, 80)
    assert qkv.shape[0] >= tiles.max_row and out.shape[0] >= tiles.max_row and lse.shape[0] == H and lse.shape[1] >= tiles.max_row
    # algorithmic FLOP: 8 * sum L^2 * D - the four products of the backward (dV, dP, dQ, dK); the recomputation of S = Q.K^T that the
    # kernels pay instead of keeping an L x L tensor is NOT counted.  Algorithmic HBM bytes of the two kernels: dQ
    # reads q, k, v, o, dO and writes dq (+ delta); dK/dV reads q, k, v, dO and writes dk, dv
    _launch("attn_bwd_hd%d" % (D // H), (8.0 * tiles.sum_sq * D, tiles.rows * (24.0 * D + 16.0 * H)), "avs_attn_bwd_q8", qkv, qkv.stride(0), D, H, tiles.start, tiles.len, tiles.q0, tiles.ntiles, tiles.tile_rows, out, dout,
            out.stride(0), lse, delta, lse.shape[1], dqkv, dqkv8, dqkv8.stride(0) if dqkv8 is not None else 0, _qrec(q8), 1 if kv_bf16 else 0, _stream())


class AttnSeqs:
    """(start row, length) of the sequences of a packed batch that the fused backward takes: min_len < L <= max_len"""

    def __init__(self, seq_lens, device, min_len, max_len):
        starts, lens, row = [], [], 0
        for L in seq_lens:
            if min_len < L <= max_len:
                starts.append(row); lens.append(L)
            row += L
        self.nseq, self.max_len, self.max_row = len(starts), max_len, row
        self.start = torch.tensor(starts, dtype=I32, device=device)
        self.len = torch.tensor(lens, dtype=I32, device=device)
        self.rows = float(sum(lens))
        self.sum_sq = float(sum(L * L for L in lens))


def attn_bwd_fused(qkv, seqs, H, out, dout, lse, dqkv, dqkv8=None, q8=None, kv_bf16=True):
    """dq, dk, dv of the sequences in `seqs` (each at most seqs.max_len = 64 | 128 | 224 tokens; 224: head dim 64, no e5m2 copy; head dim 80: 64 only) in one
    kernel: one read of q, k, v, o, dO and one evaluation of S per (sequence, head).  Rows of other sequences are not touched."""
    _chk(qkv, BF16, "attnf.qkv", 2); _chk(out, BF16, "attnf.out", 2); _chk(dout, BF16, "attnf.dout", 2)
    _chk(lse, F32, "attnf.lse", 2); _chk(dqkv, BF16, "attnf.dqkv", 2); _chk(dqkv8, U8, "attnf.dqkv8", 2)
    assert (dqkv8 is None) == (q8 is None) and (dqkv8 is None or (dqkv8.shape[0] >= seqs.max_row and dqkv8.shape[1] == qkv.shape[1]))
    D = qkv.shape[1] // 3
    assert seqs.nseq > 0 and seqs.max_len in (64, 128, 224) and D // H in (32, 64, 80) and D % H == 0
    assert seqs.max_len != 224 or (D // H == 64 and dqkv8 is None)
    assert D // H != 80 or seqs.max_len == 64
    assert dqkv.shape == qkv.shape and out.shape[1] == D and dout.shape == out.shape
    assert qkv.shape[0] >= seqs.max_row and out.shape[0] >= seqs.max_row and lse.shape[0] == H and lse.shape[1] >= seqs.max_row
    # algorithmic work: 8 * sum L^2 * D FLOP; q, k, v, o, dO read and dq, dk, dv written once (bf16), lse read
    _launch("attn_bwd_hd%d" % (D // H), (8.0 * seqs.sum_sq * D, seqs.rows * (16.0 * D + 4.0 * H)), "avs_attn_bwd_fused_q8", qkv, qkv.stride(0), D, H, seqs.start,
            seqs.len, seqs.nseq, seqs.max_len, out, dout, out.stride(0), lse, lse.shape[1], dqkv, dqkv8, dqkv8.stride(0) if dqkv8 is not None else 0,
            _qrec(q8), 1 if kv_bf16 else 0, _stream())


# ---------------------------------------------------------------------------------------------------
class InputXf(ctypes.Structure):
    """include/avsiam_hip.h: avs_input_xf - how a raw input becomes the tensor the model sees (read on the host at call time)."""
    _fields_ = [("kind", ctypes.c_int), ("mean", ctypes.c_float * 3), ("std", ctypes.c_float * 3), ("shift", ctypes.c_void_p),
                ("amp", ctypes.c_void_p), ("seed", ctypes.c_ulonglong)]

    @classmethod
    def audio(cls, mean, std, shift=None, amp=None, seed=0):
        """un-normalised fp32 fbank: (x - mean) / std [+ amp_b * U, rolled by shift_b] (dataloader.py:505-513)"""
        x = cls(1, (ctypes.c_float * 3)(float(mean), 0, 0), (ctypes.c_float * 3)(float(std), 1, 1), None, None, int(seed) & 0xFFFFFFFFFFFFFFFF)
        if (shift is None) != (amp is None):
            raise _lib.AvsiamHipError("InputXf.audio: shift and amp go together (the loader's noise augmentation draws both per sample)")
        if shift is not None:
            _chk(shift, I32, "xf.shift"); _chk(amp, F32, "xf.amp")
            x.shift, x.amp = shift.data_ptr(), amp.data_ptr()
            x._keep = (shift, amp)
        return x

    @classmethod
    def frames(cls, mean=(0.485, 0.456, 0.406), std=(0.229, 0.224, 0.225)):
        """uint8 frames: (x / 255 - mean_c) / std_c (dataloader.py:461-462, 152-155)"""
        return cls(2, (ctypes.c_float * 3)(*[float(m) for m in mean]), (ctypes.c_float * 3)(*[float(v) for v in std]), None, None, 0)


def _xf_arg(xf, kind, per_sample=None):
    if xf is None:
        return None
    if not isinstance(xf, InputXf) or xf.kind != kind:
        raise _lib.AvsiamHipError(f"input transform of kind {kind} expected")
    if per_sample is not None and xf.shift:
        assert xf._keep[0].numel() >= per_sample and xf._keep[1].numel() >= per_sample
    return ctypes.addressof(xf)


def im2col_audio(a, row_b, row_tok, out, rows, t_patches, xf=None, stride=16):
    """stride < 16: patch stride on 16 x 16 patch storage (config.stride) - the S x S corner of every 256-wide row is filled, the rest zero"""
    _chk(a, F32, "im2col.a", 3); _chk(row_b, I32, "im2col.row_b"); _chk(row_tok, I32, "im2col.row_tok"); _chk(out, BF16, "im2col.out", 2)
    assert out.shape[1] == 256 and out.shape[0] >= rows and row_b.numel() >= rows and row_tok.numel() >= rows
    assert a.shape[1] == t_patches * stride and a.shape[2] % stride == 0
    _call("avs_im2col_audio_s", a, row_b, row_tok, out, rows, a.shape[1], a.shape[2], t_patches, int(stride), _xf_arg(xf, 1, a.shape[0]), _stream())


def im2col_video(v, row_img, row_tok, out, rows, xf=None, stride=16):
    _chk(v, U8 if xf is not None else F32, "im2col.v", 4); _chk(row_img, I32, "im2col.row_img"); _chk(row_tok, I32, "im2col.row_tok"); _chk(out, BF16, "im2col.out", 2)
    NF, C, H, W = v.shape
    assert out.shape[1] == C * 256 and out.shape[0] >= rows and row_img.numel() >= rows and row_tok.numel() >= rows and W % stride == 0
    _call("avs_im2col_video_s", v, row_img, row_tok, out, rows, C, H, W, int(stride), _xf_arg(xf, 2), _stream())


PLAN_FIELDS = 16      # int32 per sequence descriptor of avs_mask_plan (csrc/maskplan.hip PlanSeq; fields 12..15: the grouped decoder layout)
PLAN_CLASSIC = [-1, 0, 0, 0]      # fields 12..15 of a sequence in the classic (position-ordered) decoder layout


def mask_plan(seqs_dev, seqs_host, seed, row_src, row_tok, tmask_lo=None, tmask_hi=None, fmask=None, src_row=None, mask_out=None,
              ids_out=None, seed_dev=None, grouped=None):
    """Draw the random masks of every sequence in `seqs` on the device.  seqs_host (numpy int32 [nseq, PLAN_FIELDS]) is the host copy
    of seqs_dev used to validate every offset before the launch.  seed_dev (int64 [1] device tensor): the Philox key is read from it
    when the kernel runs instead of `seed` (graph_step: kernel arguments are frozen in a captured graph).
    grouped = (pos_row, row_of_pos, pred_id): the sequences with a decoder part are laid out in the GROUPED decoder order (scored rows of a
    sample first: csrc/maskplan.hip) and the three extra index arrays are written."""
    _chk(seqs_dev, I32, "plan.seqs", 2); _chk(row_src, I32, "plan.row_src"); _chk(row_tok, I32, "plan.row_tok")
    _chk(src_row, I32, "plan.src_row"); _chk(mask_out, F32, "plan.mask"); _chk(ids_out, I32, "plan.ids")
    for t in (tmask_lo, tmask_hi, fmask):
        _chk(t, I32, "plan.bitmask")
    nseq = seqs_host.shape[0]
    assert seqs_dev.shape == (nseq, PLAN_FIELDS) and seqs_host.shape[1] == PLAN_FIELDS
    L, keep, row_off, dec_off, t_p, ids_off, mask_off = (seqs_host[:, i] for i in (0, 1, 2, 4, 6, 7, 8))
    assert (L > 0).all() and (L <= 1024).all() and (keep >= 0).all() and (keep <= L).all()
    assert (row_off >= 0).all() and int((row_off + keep).max()) <= min(row_src.numel(), row_tok.numel())
    if (dec_off >= 0).any():
        sel = dec_off >= 0
        assert src_row is not None and mask_out is not None
        assert int((dec_off[sel] + L[sel]).max()) <= src_row.numel() and int((mask_off[sel] + L[sel]).max()) <= mask_out.numel()
        assert (mask_off[sel] >= 0).all()
    if (ids_off >= 0).any():
        sel = ids_off >= 0
        assert ids_out is not None and int((ids_off[sel] + L[sel]).max()) <= ids_out.numel()
    if (t_p > 0).any():
        assert all(t is not None and t.numel() >= nseq for t in (tmask_lo, tmask_hi, fmask))
        sel = t_p > 0
        # time patches: 64 bits in tmask_lo / tmask_hi + 32 in the descriptor (field 9, PlanSeq.tmask_x); frequency patches: 32 bits
        assert (t_p[sel] <= 96).all() and (L[sel] % t_p[sel] == 0).all() and (L[sel] // t_p[sel] <= 32).all()
    dm = seqs_host[:, 12]
    if grouped is None:
        assert (dm < 0).all(), "grouped decoder layout in the descriptors but no index arrays for it"
    else:
        pos_row, row_of_pos, pred_id = grouped
        _chk(pos_row, I32, "plan.pos_row"); _chk(row_of_pos, I32, "plan.row_of_pos"); _chk(pred_id, I32, "plan.pred_id")
        sel = dec_off >= 0
        assert sel.any() and (dm[sel] >= 0).all() and src_row is not None and mask_out is not None
        dk, po, pb = seqs_host[:, 13], seqs_host[:, 14], seqs_host[:, 15]
        nm = L - keep
        assert (dk[sel] >= 0).all() and (po[sel] >= 0).all() and (pb[sel] >= 0).all()
        assert int((dm[sel] + nm[sel]).max()) <= min(src_row.numel(), pos_row.numel()) and int((dk[sel] + keep[sel]).max()) <= min(src_row.numel(), pos_row.numel())
        assert int((dec_off[sel] + L[sel]).max()) <= row_of_pos.numel() and int((po[sel] + nm[sel]).max()) <= pred_id.numel()
        if seed_dev is not None:
            assert seed_dev.dtype == torch.int64 and seed_dev.is_cuda and seed_dev.numel() >= 1
        _call("avs_mask_plan_grouped", seqs_dev, nseq, tmask_lo, tmask_hi, fmask, (int(seed) & 0xFFFFFFFFFFFFFFFF) if seed_dev is None else 0, seed_dev,
              row_src, row_tok, src_row, mask_out, ids_out, pos_row, row_of_pos, pred_id, _stream())
        return
    if seed_dev is not None:
        assert seed_dev.dtype == torch.int64 and seed_dev.is_cuda and seed_dev.numel() >= 1
        _call("avs_mask_plan_dev", seqs_dev, nseq, tmask_lo, tmask_hi, fmask, seed_dev, row_src, row_tok, src_row, mask_out, ids_out, _stream())
        return
    _call("avs_mask_plan", seqs_dev, nseq, tmask_lo, tmask_hi, fmask, int(seed) & 0xFFFFFFFFFFFFFFFF, row_src, row_tok, src_row,
              mask_out, ids_out, _stream())


def cast_scale(x, y, n, alpha):
    _chk(x, F32, "cast.x"); _chk(y, BF16, "cast.y")
    assert x.numel() >= n and y.numel() >= n and n % 4 == 0
    _call("avs_cast_scale_bf16", x, y, n, float(alpha), _stream())


def scatter_add_rows(src, idx, dst, rows, scale=1.0):
    _chk(src, BF16, "scatter.src", 2); _chk(idx, I32, "scatter.idx"); _chk(dst, F32, "scatter.dst")
    D = src.shape[1]
    assert src.shape[0] >= rows and idx.numel() >= rows and dst.numel() % D == 0
    _call("avs_scatter_add_rows", src, idx, dst, rows, D, float(scale), _stream())


def colsum(x, out, rows):
    """out[c] += sum_r x[r, c]; x may be a column range of a wider row-major matrix (stride(1) == 1)"""
    if not (x.is_cuda and x.dtype == BF16 and x.dim() == 2 and x.stride(1) == 1 and x.stride(0) >= x.shape[1] and x.storage_offset() % 8 == 0):
        raise _lib.AvsiamHipError("colsum.x: need a row-major bf16 GPU matrix (or a column range of one, 16-byte aligned)")
    _chk(out, F32, "colsum.out")
    assert x.shape[0] >= rows and out.numel() == x.shape[1]
    _call("avs_colsum_bf16", x, x.stride(0), out, rows, x.shape[1], _stream())


def vecmat(x, W, y, alpha=1.0):
    """y[n] += alpha * sum_k x[k] * W[k, n]   (x fp32 [K], W bf16 [K, N], y fp32 [N])"""
    _chk(x, F32, "vecmat.x"); _chk(W, BF16, "vecmat.W", 2); _chk(y, F32, "vecmat.y")
    K, N = W.shape
    assert x.numel() == K and y.numel() == N and N % 256 == 0 and K % 32 == 0
    _call("avs_vecmat_bf16", x, W, W.stride(0), y, K, N, float(alpha), _stream())


class VecmatBatch:
    """y_i += x_i . W_i for many (x, W, y) of one shape with ONE launch (avs_vecmat_bf16_batched): the value thirds of a stack's qkv bias gradients"""

    def __init__(self):
        self.entries, self.keep, self.desc, self.shape = [], [], None, None

    def add(self, x, W, y):
        assert self.desc is None, "table already built"
        _chk(x, F32, "vecmat.x"); _chk(W, BF16, "vecmat.W", 2); _chk(y, F32, "vecmat.y")
        K, N = W.shape
        assert x.numel() == K and y.numel() == N and N % 256 == 0 and K % 32 == 0
        shape = (K, N, W.stride(0))
        assert self.shape in (None, shape), "one shape per batch"
        self.shape = shape
        self.entries.append((x.data_ptr(), W.data_ptr(), y.data_ptr()))
        self.keep.append((x, W, y))

    def build(self, dev):
        self.desc = torch.tensor(self.entries, dtype=torch.int64, device=dev)

    def run(self, alpha=1.0):
        K, N, ld = self.shape
        _call("avs_vecmat_bf16_batched", self.desc, len(self.entries), ld, K, N, float(alpha), _stream())


def unshuffle_fwd(x, src_row, pos_row, row_mod, mask_token, pos_a, pos_v, mod_a, mod_v, out, rows):
    _chk(x, F32, "unshuffle.x", 2); _chk(out, F32, "unshuffle.out", 2); _chk(src_row, I32, "unshuffle.src"); _chk(pos_row, I32, "unshuffle.pos")
    _chk(row_mod, U8, "unshuffle.mod")
    D = x.shape[1]
    for t in (mask_token, pos_a, pos_v, mod_a, mod_v):
        _chk(t, F32, "unshuffle.param")
    assert out.shape[1] == D and out.shape[0] >= rows and src_row.numel() >= rows and pos_row.numel() >= rows and row_mod.numel() >= rows
    assert mask_token.numel() == D and mod_a.numel() == D and mod_v.numel() == D
    _call("avs_unshuffle_fwd", x, src_row, pos_row, row_mod, mask_token, pos_a, pos_v, pos_a.numel() // D, mod_a, mod_v, out,
              rows, D, _stream())


def unshuffle_bwd(dout, src_row, B, T, La, Lv, dx, dpos_a, dpos_v, dmask, dmod_a, dmod_v, row_of_pos=None):
    """row_of_pos (grouped decoder layout): the decoder row of position (b, l); None: the rows are in position order"""
    _chk(dout, F32, "unshuffleb.dout", 2); _chk(dx, F32, "unshuffleb.dx", 2); _chk(src_row, I32, "unshuffleb.src"); _chk(row_of_pos, I32, "unshuffleb.map")
    D = dout.shape[1]
    assert dout.shape[0] >= B * (La + T * Lv) and src_row.numel() >= B * (La + T * Lv) and dx.shape[1] == D
    assert dpos_a.numel() == La * D and dpos_v.numel() == Lv * D and dmask.numel() == D
    assert row_of_pos is None or row_of_pos.numel() >= B * (La + T * Lv)
    _call("avs_unshuffle_bwd_map", dout, src_row, B, T, La, Lv, dx, dpos_a, dpos_v, dmask, dmod_a, dmod_v, D, row_of_pos, _stream())


def expand_rows(inp, src_row, out, rows, cols=None):
    """out[r, :cols] = inp[src_row[r], :cols] where src_row[r] >= 0, else 0 (bf16).  inp None: only the rows with src_row[r] < 0 are zeroed."""
    _chk(inp, BF16, "expand.in", 2); _chk(out, BF16, "expand.out", 2); _chk(src_row, I32, "expand.src")
    cols = out.shape[1] if cols is None else cols
    assert out.shape[0] >= rows and src_row.numel() >= rows and cols <= out.shape[1] and (inp is None or inp.shape[1] >= cols)
    _call("avs_expand_rows_bf16", inp, inp.stride(0) if inp is not None else 0, src_row, out, out.stride(0), rows, cols, _stream())


def segment_mean_fwd(y, seg_start, reps, nseg, row_map=None, max_row=None):
    """row_map (int32 [nseg], values < max_row <= reps rows; validated by the caller who built it): segment s -> reps[row_map[s]]"""
    _chk(y, F32, "segmean.y", 2); _chk(seg_start, I32, "segmean.seg"); _chk(reps, F32, "segmean.reps", 2); _chk(row_map, I32, "segmean.map")
    assert seg_start.numel() >= nseg + 1 and reps.shape[0] >= (nseg if row_map is None else max_row) and reps.shape[1] == y.shape[1]
    assert row_map is None or (row_map.numel() >= nseg and max_row is not None)
    _call("avs_segment_mean_fwd", y, seg_start, reps, nseg, y.shape[1], row_map, _stream())


def segment_mean_bwd(dreps, seg_start, dy, nseg, scale=1.0, row_map=None, max_row=None):
    _chk(dy, F32, "segmeanb.dy", 2); _chk(seg_start, I32, "segmeanb.seg"); _chk(dreps, F32, "segmeanb.dreps", 2); _chk(row_map, I32, "segmeanb.map")
    assert seg_start.numel() >= nseg + 1 and dreps.shape[0] >= (nseg if row_map is None else max_row) and dreps.shape[1] == dy.shape[1]
    assert row_map is None or (row_map.numel() >= nseg and max_row is not None)
    _call("avs_segment_mean_bwd", dreps, seg_start, dy, nseg, dy.shape[1], float(scale), row_map, _stream())


def mae_loss_fwd(pred, inp, mask, row_loss, loss, audio, L, nmask, total=None, total_init=True, xf=None, stride=16, row_id=None, id_base=0):
    """row_id / id_base (compact predictions: only the scored rows exist): prediction row r scores the (sample, token) row_id[r] - id_base of mask"""
    _chk(row_id, I32, "mae.row_id")
    _chk(pred, F32, "mae.pred", 2); _chk(inp, U8 if (xf is not None and not audio) else F32, "mae.inp"); _chk(mask, F32, "mae.mask"); _chk(row_loss, F32, "mae.row_loss"); _chk(loss, F32, "mae.loss")
    _chk(total, F32, "mae.total")
    rows = mask.numel()
    if audio:
        C, H, W = 1, inp.shape[1], inp.shape[2]          # [B, time, mel]
        assert inp.dim() == 3 and rows == inp.shape[0] * L and pred.shape[1] == 256 and (H // stride) * (W // stride) == L
    else:
        C, H, W = inp.shape[-3:]
        assert rows == inp.numel() // (C * H * W) * L and pred.shape[1] == 256 * C and (H // stride) * (W // stride) == L
    prows = rows if row_id is None else row_id.numel()
    assert pred.shape[0] >= prows and row_loss.numel() >= prows
    _call("avs_mae_loss_fwd_id", pred, inp, mask, row_loss, loss, total, int(bool(total_init)), prows, int(audio), L, C, H, W,
              float(nmask), int(stride), _xf_arg(xf, 1 if audio else 2, inp.shape[0] if audio else None), row_id, int(id_base), _stream())


def mae_loss_bwd(pred, inp, mask, gout, dpred, audio, L, nmask, xf=None, stride=16, row_id=None, id_base=0):
    _chk(row_id, I32, "maeb.row_id")
    _chk(pred, F32, "maeb.pred", 2); _chk(inp, U8 if (xf is not None and not audio) else F32, "maeb.inp"); _chk(mask, F32, "maeb.mask"); _chk(gout, F32, "maeb.gout"); _chk(dpred, BF16, "maeb.dpred", 2)
    rows = mask.numel()
    if audio:
        C, H, W = 1, inp.shape[1], inp.shape[2]
    else:
        C, H, W = inp.shape[-3:]
    prows = rows if row_id is None else row_id.numel()
    assert pred.shape[0] >= prows and dpred.shape[0] >= prows and dpred.shape[1] == pred.shape[1] == 256 * C
    _call("avs_mae_loss_bwd_id", pred, inp, mask, gout, dpred, prows, int(audio), L, C, H, W, float(nmask), int(stride),
              _xf_arg(xf, 1 if audio else 2, inp.shape[0] if audio else None), row_id, int(id_base), _stream())


def l2norm_fwd(x, xn, norm):
    _chk(x, F32, "l2.x", 2); _chk(xn, F32, "l2.xn", 2); _chk(norm, F32, "l2.norm")
    assert xn.shape == x.shape and norm.numel() >= x.shape[0]
    _call("avs_l2norm_fwd", x, xn, norm, x.shape[0], x.shape[1], _stream())


def l2norm_bwd(dxn, xn, norm, dx, scale=1.0):
    for t in (dxn, xn, dx):
        _chk(t, F32, "l2b", 2)
    _chk(norm, F32, "l2b.norm")
    assert dxn.shape == xn.shape == dx.shape
    _call("avs_l2norm_bwd", dxn, xn, norm, dx, xn.shape[0], xn.shape[1], float(scale), _stream())


def gemm_f32_small(A, B, C, M, N, K, sa, sb, alpha=1.0):
    """C[M,N] = alpha * sum_k A(m,k) B(k,n); sa = (stride_m, stride_k) of A, sb = (stride_k, stride_n) of B, in elements."""
    _chk(A, F32, "sgemm.A"); _chk(B, F32, "sgemm.B"); _chk(C, F32, "sgemm.C", 2)
    assert C.shape[0] >= M and C.shape[1] == N
    assert (M - 1) * sa[0] + (K - 1) * sa[1] < A.numel() and (K - 1) * sb[0] + (N - 1) * sb[1] < B.numel()
    _call("avs_gemm_f32_small", A, sa[0], sa[1], B, sb[0], sb[1], C, C.stride(0), M, N, K, float(alpha), _stream())


def infonce_fwd(total, stats, out, weight=1.0):
    """out = {nce, c_acc, weight * nce}"""
    _chk(total, F32, "nce.total", 2); _chk(stats, F32, "nce.stats", 2); _chk(out, F32, "nce.out")
    N = total.shape[0]
    assert total.shape[1] == N and stats.shape == (N, 4) and out.numel() >= 3
    _call("avs_infonce_fwd", total, stats, out, N, float(weight), _stream())


def infonce_dlogits(total, stats, gout, weight, dtotal):
    _chk(total, F32, "nceb.total", 2); _chk(stats, F32, "nceb.stats", 2); _chk(gout, F32, "nceb.gout"); _chk(dtotal, F32, "nceb.dtotal", 2)
    N = total.shape[0]
    assert dtotal.shape == total.shape
    _call("avs_infonce_dlogits", total, stats, gout, float(weight), dtotal, N, _stream())


def transpose_bf16(x, out):
    _chk(x, BF16, "tr.x", 2); _chk(out, BF16, "tr.out", 2)
    assert out.shape == (x.shape[1], x.shape[0])
    _call("avs_transpose_bf16", x, out, x.shape[0], x.shape[1], _stream())


def transpose_batched(desc, tile_map, ntiles):
    """desc: int64 [nmat, 6] on the device, built (and bounds-checked) by ParamArena from its own views."""
    _chk(desc, torch.int64, "trb.desc", 2); _chk(tile_map, I32, "trb.map")
    assert desc.shape[1] == 6 and tile_map.numel() == ntiles
    _call("avs_transpose_batched", desc, tile_map, ntiles, _stream())


def cast_bf16(x, y, n):
    _chk(x, F32, "castb.x"); _chk(y, BF16, "castb.y")
    assert x.numel() >= n and y.numel() >= n
    _call("avs_cast_bf16", x, y, n, _stream())


def adam(p, g, m, v, p_bf16, n, lr, step, beta1=0.95, beta2=0.999, eps=1e-8, weight_decay=5e-7, grad_scale=1.0, step_dev=None):
    """step_dev (int32 [1] device tensor): the step count is read from it when the kernel runs instead of `step` (graph_step)"""
    for t in (p, g, m, v):
        _chk(t, F32, "adam")
    _chk(p_bf16, BF16, "adam.p_bf16")
    assert n % 4 == 0 and all(t.numel() >= n for t in (p, g, m, v)) and (p_bf16 is None or p_bf16.numel() >= n)
    if step_dev is not None:
        assert step_dev.dtype == torch.int32 and step_dev.is_cuda and step_dev.numel() >= 1
        _call("avs_adam_dev", p, g, m, v, p_bf16, n, float(lr), float(beta1), float(beta2), float(eps), float(weight_decay), step_dev,
              float(grad_scale), _stream())
        return
    _call("avs_adam", p, g, m, v, p_bf16, n, float(lr), float(beta1), float(beta2), float(eps), float(weight_decay), int(step),
              float(grad_scale), _stream())
