"""Forward-only schedule of the fine-tuned model's inference modes on the HIP kernels.

``CAVMAEFT_BASE.forward(a, v, mode, is_eval)`` (/root/reference/src/models/cav_mae_base.py:827-1035) runs the Siamese
ViT over all 512 audio tokens and all T x 196 frame tokens (nothing is masked), then, depending on the mode, pools and
classifies per modality (audioonly :828-847, videoonly :850-866), returns the normalised token matrices (retrieval
:869-892), or concatenates audio and frame tokens and runs the two fusion blocks before the joint head (mm_grad
:894-1035; the evaluation variant loops over the 10 frames of a clip, :940-961).

Layout: as in the training engine every sequence is packed into one [rows, D] token matrix - audio rows first, then the
B x T frames - so the 12 shared blocks are 12 x {LN(row-selected affine), QKV GEMM, varlen attention, proj, LN, fc1+GELU,
fc2} over all rows at once instead of one pass per modality.  The 10-frame evaluation loop of mm_grad becomes ONE packed
batch of B x 10 (audio | frame t) sequences: the audio rows are written 10 times by the final LayerNorm's row map.
No activation is kept (engine.Stack(inference=True)).
"""
import torch

from . import ops
from .arena import ParamArena
from .config import AVSiamConfig
from .engine import BF16, F32, I32, U8, LN_EPS_BLOCK, LN_EPS_FINAL, BlockParams, Norm, PatchEmbedder, Stack, _ln_fwd, _z

EVAL_FRAMES = 10        # `for t_idx in range(10)` at cav_mae_base.py:940


class Head:
    """nn.Sequential(LayerNorm(width), Linear(width, label_dim)) (cav_mae_base.py:809-815).  The GEMM kernel wants N in
    multiples of 128, so the bf16 weight copy is zero-padded to that (refresh() re-derives it from the fp32 master)."""

    def __init__(self, arena: ParamArena, name, width, label_dim, max_rows, dev):
        self.arena, self.name, self.width, self.L = arena, name, width, label_dim
        self.norm = Norm(arena, f"{name}.0")
        self.npad = ops.pad_rows(label_dim, 128)
        self.w = _z((self.npad, width), BF16, dev)
        self.b = _z((self.npad,), F32, dev)
        rp = ops.pad_rows(max_rows, 256)
        self.h = _z((rp, width), BF16, dev)
        self.stat = [_z((rp,), F32, dev) for _ in range(2)]
        self.out = _z((rp, self.npad), F32, dev)
        self.refresh()

    def refresh(self):
        self.w[:self.L].copy_(self.arena.w(f"{self.name}.1.weight"))
        self.b[:self.L].copy_(self.arena.w(f"{self.name}.1.bias"))

    def forward(self, x, n):
        """x: fp32 [>= n, width] pooled features -> fp32 [n, label_dim] logits (a view of the head's output buffer)."""
        ops.layernorm_fwd(x, self.norm.g, self.norm.b, self.h, self.stat[0], self.stat[1], n, LN_EPS_BLOCK)
        ops.gemm_nt(self.h, self.w, self.out, n, bias=self.b)
        return self.out[:n, :self.L]


class Encoder:
    """The shared ViT over a packed batch of full sequences: `na` audio sequences (La tokens, '_a' norms) followed by
    `nv` frame sequences (Lv tokens, '_v' norms), then the per-modality final norm (cav_mae_base.py:832-841,853-860)."""

    def __init__(self, arena, cfg: AVSiamConfig, na, nv, blocks, final, dev):
        D, La, Lv = cfg.embed_dim, cfg.audio_tokens, cfg.video_tokens
        self.cfg, self.na, self.nv = cfg, na, nv
        self.rows_a, self.rows_v = na * La, nv * Lv
        self.rows = self.rows_a + self.rows_v
        self.blocks, self.final = blocks, final
        row_mod = torch.cat([torch.zeros(self.rows_a, dtype=U8), torch.ones(self.rows_v, dtype=U8)]).to(dev)
        self.stack = Stack(dev, self.rows, D, cfg.num_heads, D * cfg.mlp_ratio, [La] * na + [Lv] * nv, cfg.depth, row_mod, inference=True)
        self.emb_a = self.emb_v = None
        if na:
            self.emb_a = PatchEmbedder(arena, dev, self.rows_a, True, cfg)
            self.emb_a.set_rows(torch.arange(na).repeat_interleave(La).to(I32), torch.arange(La).repeat(na).to(I32))
        if nv:
            self.emb_v = PatchEmbedder(arena, dev, self.rows_v, False, cfg)
            self.emb_v.set_rows(torch.arange(nv).repeat_interleave(Lv).to(I32), torch.arange(Lv).repeat(nv).to(I32))
        self.yf = _z((self.stack.rp, D), F32, dev)              # final-norm output, fp32 token matrix
        self.fstat = [_z((self.stack.rp,), F32, dev) for _ in range(2)]
        seg = [0]
        for L in [La] * na + [Lv] * nv:
            seg.append(seg[-1] + L)
        self.nseq = na + nv
        self.seg_start = torch.tensor(seg, dtype=I32, device=dev)
        self.pooled = _z((ops.pad_rows(self.nseq, 256), D), F32, dev)

    def forward(self, audio, frames):
        st = self.stack
        if self.na:
            self.emb_a.forward(audio, st.x[0][:self.rows_a])
        if self.nv:
            self.emb_v.forward(frames, st.x[0][self.rows_a:])
        st.forward(self.blocks)
        _ln_fwd(st.out, self.final, self.yf, self.fstat[0], self.fstat[1], self.rows, LN_EPS_FINAL, st.row_mod)

    def pool(self):
        """Mean over the tokens of every sequence (`.mean(dim=1)`, :843,862) -> [na audio | nv frames] x D."""
        ops.segment_mean_fwd(self.yf, self.seg_start, self.pooled, self.nseq)
        return self.pooled


class FtForward:
    """All inference modes for one (batch, frames) shape; encoders are built on first use of a mode."""

    def __init__(self, arena: ParamArena, cfg: AVSiamConfig, label_dim, batch, frames, dev):
        self.arena, self.cfg, self.B, self.T, self.dev, self.L = arena, cfg, batch, frames, dev, label_dim
        D = cfg.embed_dim
        self.blocks = [BlockParams(arena, f"vit_base.blocks.{i}", "_a", "_v") for i in range(cfg.depth)]
        self.final = [Norm(arena, "vit_base.norm_a"), Norm(arena, "vit_base.norm")]
        self.blk_mm = [BlockParams(arena, "mm_layer_1", "_a"), BlockParams(arena, "mm_layer_2", "_a")]
        nmax = batch * max(frames, 1)
        self.head_v = Head(arena, "mlp_head", D, label_dim, nmax, dev)
        self.head_a = Head(arena, "mlp_head_a", D, label_dim, batch, dev)
        self.head_mm = Head(arena, "mlp_head_mm", 2 * D, label_dim, nmax, dev)
        self._enc = {}
        self._joint = {}

    def refresh_heads(self):
        for h in (self.head_v, self.head_a, self.head_mm):
            h.refresh()

    def encoder(self, kind):
        if kind not in self._enc:
            na = self.B if "a" in kind else 0
            nv = self.B * self.T if "v" in kind else 0
            self._enc[kind] = Encoder(self.arena, self.cfg, na, nv, self.blocks, self.final, self.dev)
        return self._enc[kind]

    # ---- modes ---------------------------------------------------------------------------------------------------
    def audioonly(self, audio):
        enc = self.encoder("a")
        enc.forward(audio, None)
        return self.head_a.forward(enc.pool(), self.B)                                     # [B, L]

    def videoonly(self, frames):
        enc = self.encoder("v")
        enc.forward(None, frames)
        return self.head_v.forward(enc.pool(), self.B * self.T).view(self.B, self.T, self.L)

    def retrieval(self, audio, frames, frame_index=5):
        """-> (audio tokens [B, La, D], tokens of frame 5 [B, Lv, D]) after the final norms (:892)."""
        cfg = self.cfg
        enc = self.encoder("av")
        enc.forward(audio, frames)
        a = enc.yf[:enc.rows_a].view(self.B, cfg.audio_tokens, cfg.embed_dim)
        v = enc.yf[enc.rows_a:enc.rows].view(self.B, self.T, cfg.video_tokens, cfg.embed_dim)
        return a, v[:, frame_index]

    def _joint_stack(self, nf):
        """Fusion stage for `nf` frames per clip: B*nf sequences [audio tokens | tokens of frame t] (:944,1022)."""
        if nf not in self._joint:
            cfg, B, T, dev = self.cfg, self.B, self.T, self.dev
            La, Lv, D = cfg.audio_tokens, cfg.video_tokens, cfg.embed_dim
            Lj = La + Lv
            nseq = B * nf
            st = Stack(dev, nseq * Lj, D, cfg.num_heads, D * cfg.mlp_ratio, [Lj] * nseq, 2, inference=True)
            b = torch.arange(B).view(B, 1)
            i = torch.arange(La).view(1, La)
            maps_a = [((b * nf + t) * Lj + i).reshape(-1).to(I32).to(dev) for t in range(nf)]
            s = torch.arange(nseq).view(nseq, 1)                                   # frame sequence b*T+t == joint sequence b*nf+t (T == nf)
            map_v = (s * Lj + La + torch.arange(Lv).view(1, Lv)).reshape(-1).to(I32).to(dev)
            seg = []
            for q in range(nseq):
                seg += [q * Lj, q * Lj + La]
            seg.append(nseq * Lj)
            seg_start = torch.tensor(seg, dtype=I32, device=dev)
            pooled = _z((ops.pad_rows(2 * nseq, 256), D), F32, dev)
            self._joint[nf] = (st, maps_a, map_v, seg_start, pooled)
        return self._joint[nf]

    def mm_grad(self, audio, frames, is_eval):
        """is_eval: [B, 10, L] joint logits, one per frame (:961).  Otherwise (out, out_a, out_v) for single-frame clips
        (:1035; the reference's torch.cat of [B,...] audio and [B*T,...] frame tokens only works for T == 1)."""
        B, T = self.B, self.T
        nf = EVAL_FRAMES if is_eval else 1
        if T != nf:
            raise ValueError(f"mm_grad(is_eval={is_eval}) needs clips of {nf} frame(s), got {T} (cav_mae_base.py:940,1022)")
        enc = self.encoder("av")
        enc.forward(audio, frames)
        st, maps_a, map_v, seg_start, pooled = self._joint_stack(nf)
        xj = st.x[0]
        so = enc.stack.out
        fin_a, fin_v = self.final[:1], self.final[1:]
        for t in range(nf):                                                        # audio tokens go to every (b, t) sequence
            _ln_fwd(so, fin_a, xj, enc.fstat[0], enc.fstat[1], enc.rows_a, LN_EPS_FINAL, out_map=maps_a[t])
        _ln_fwd(so[enc.rows_a:], fin_v, xj, enc.fstat[0][enc.rows_a:], enc.fstat[1][enc.rows_a:], enc.rows_v, LN_EPS_FINAL, out_map=map_v)
        st.forward(self.blk_mm)
        ops.segment_mean_fwd(st.out, seg_start, pooled, 2 * B * nf)                # [a-part mean | v-part mean] per sequence (:948-951)
        out = self.head_mm.forward(pooled.view(-1, 2 * self.cfg.embed_dim), B * nf)
        if is_eval:
            return out.view(B, nf, self.L)
        pa = enc.pool()
        out_a = self.head_a.forward(pa, B)
        out_v = self.head_v.forward(pa[B:], B)
        return out, out_a, out_v
