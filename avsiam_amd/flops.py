"""Algorithmic FLOPs of one pre-training step (the denominator of the MFMA roofline).

Block on N tokens of width D: 24 N D^2 + 4 N^2 D forward (QKV 6ND^2, proj 2ND^2, MLP 16ND^2, attention 4N^2D).
Training = 3x forward (fwd + dgrad + wgrad) except the patch embedding (2x: no input gradient), counted on
KEPT tokens only.  Attention recompute, dead parameters and discarded tokens are NOT counted.
"""
from .config import AVSiamConfig
from .maskplan import group_ratio, group_sizes, len_keep


def blk(n, d):
    return 24 * n * d * d + 4 * n * n * d


def step_flops(cfg: AVSiamConfig, batch: int):
    """-> dict(fwd_pass1, fwd_pass2, fwd_embed, train_total) in FLOPs for one step of `batch` samples."""
    D, Dd, T = cfg.embed_dim, cfg.dec_dim, cfg.frames
    La, Lv, ka, kv = cfg.audio_tokens, cfg.video_tokens, cfg.keep_a, cfg.keep_v
    pa = cfg.patch * cfg.patch
    pv = pa * cfg.in_chans
    p1 = 0
    emb = 0
    for g, n in enumerate(group_sizes(batch, cfg.n_groups)):
        na, nv = len_keep(La, group_ratio(g)), len_keep(Lv, group_ratio(g))
        p1 += n * cfg.depth * (blk(na, D) + T * blk(nv, D))
        emb += n * (na * 2 * pa * D + T * nv * 2 * pv * D)
    n_enc = ka + T * kv
    per = cfg.depth * (blk(ka, D) + T * blk(kv, D)) + 2 * blk(n_enc, D) + n_enc * 2 * D * Dd \
        + cfg.dec_depth * blk(La + T * Lv, Dd) + La * 2 * Dd * pa + T * Lv * 2 * Dd * pv
    p2 = batch * per
    emb += batch * (ka * 2 * pa * D + T * kv * 2 * pv * D)
    return {"fwd_pass1": p1, "fwd_pass2": p2, "fwd_embed": emb, "train_total": 3 * (p1 + p2) + 2 * emb}


def gflop_per_sample(cfg: AVSiamConfig, batch: int):
    return step_flops(cfg, batch)["train_total"] / batch / 1e9
