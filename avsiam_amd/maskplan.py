"""Explicit mask plans for the two passes of the AVSiam pre-training step.

The reference draws its masks inside ``forward`` from three global RNGs (torch.rand / torch.randperm
/ python ``random.sample``) and breaks ties with an unstable argsort
(/root/reference/src/models/cav_mae_base.py:365-439,533-550), so its results are only reproducible
when the *plan* - which tokens each sample keeps - is made an explicit input.
This module holds the plan data structures and a host generator with the reference's distribution.

ContrastivePlan (pass 1, forward_encoder_mmixed :508-594)
    a_group[b], v_group[b]   multi-ratio group (0..n_groups-1) of sample b for audio / video
    a_keep[b]                1-D LongTensor of kept audio token ids (length keep(La, 0.2*group))
    v_keep[b][t]             same per frame (length keep(Lv, 0.2*group))
MaePlan (pass 2, forward_encoder :441-504, 75 % unstructured)
    ids_keep_a [B,keep_a]  ids_restore_a [B,La]  ids_keep_v [B,T,keep_v]  ids_restore_v [B,T,Lv]
"""
import math
import random as _pyrandom
from dataclasses import dataclass
from typing import List

import torch

from .config import AVSiamConfig


def group_ratio(g: int) -> float:
    return 0 + 0.2 * g                                    # cav_mae_base.py:546,549 (same float expression)


def len_keep(L: int, ratio: float) -> int:
    return int(L * (1 - ratio))                           # :372,399


def group_sizes(batch: int, n_groups: int = 5):
    """Sizes produced by ``torch.chunk(perm, 5)`` (:534): ceil(B/5)-sized chunks, possibly fewer than 5."""
    c = math.ceil(batch / n_groups)
    sizes = []
    left = batch
    while left > 0:
        sizes.append(min(c, left))
        left -= c
    return sizes


@dataclass
class ContrastivePlan:
    a_group: torch.Tensor
    v_group: torch.Tensor
    a_keep: List[torch.Tensor]
    v_keep: List[List[torch.Tensor]]

    @property
    def batch(self):
        return len(self.a_keep)


@dataclass
class MaePlan:
    ids_keep_a: torch.Tensor
    ids_restore_a: torch.Tensor
    ids_keep_v: torch.Tensor
    ids_restore_v: torch.Tensor

    @property
    def batch(self):
        return self.ids_keep_a.shape[0]

    def mask_a(self):
        """0 = kept, 1 = removed, in original token order (:385-388)."""
        return (self.ids_restore_a >= self.ids_keep_a.shape[1]).float()

    def mask_v(self):
        return (self.ids_restore_v >= self.ids_keep_v.shape[-1]).float()


def _unstructured(noise: torch.Tensor, keep: int):
    ids_shuffle = torch.argsort(noise, dim=-1, stable=True)
    ids_restore = torch.argsort(ids_shuffle, dim=-1, stable=True)
    return ids_shuffle[..., :keep], ids_restore


def make_mae_plan(cfg: AVSiamConfig, batch: int, gen: torch.Generator) -> MaePlan:
    """random_masking_unstructured at ratio 0.75 for both modalities (:476-477, ratios hard-coded :696)."""
    na = torch.rand(batch, cfg.audio_tokens, generator=gen)
    nv = torch.rand(batch, cfg.frames, cfg.video_tokens, generator=gen)
    ka, ra = _unstructured(na, cfg.keep_a)
    kv, rv = _unstructured(nv, cfg.keep_v)
    return MaePlan(ka.contiguous(), ra.contiguous(), kv.contiguous(), rv.contiguous())


def make_contrastive_plan(cfg: AVSiamConfig, batch: int, gen: torch.Generator, pyrng: _pyrandom.Random) -> ContrastivePlan:
    """Two independent batch permutations chunked into <=5 groups (:533-538); group g masks audio with
    random_masking_structured(ratio 0.2 g, t, f=8, 'tf') (:546, :392-439) and video with
    random_masking_unstructured(ratio 0.2 g) (:549)."""
    t, f = cfg.audio_t, cfg.audio_f
    sizes = group_sizes(batch, cfg.n_groups)
    perm_a = torch.randperm(batch, generator=gen)
    perm_v = torch.randperm(batch, generator=gen)
    a_group = torch.zeros(batch, dtype=torch.int64)
    v_group = torch.zeros(batch, dtype=torch.int64)
    a_keep = [None] * batch
    v_keep = [None] * batch
    off = 0
    for g, n in enumerate(sizes):
        r = group_ratio(g)
        idx_a = perm_a[off:off + n]
        idx_v = perm_v[off:off + n]
        off += n
        noise = torch.rand(n, cfg.audio_tokens, generator=gen).reshape(n, f, t)
        for i in range(n):                                    # :415-418 time columns
            for k in pyrng.sample(range(t), int(t * r * 0.7)):
                noise[i, :, k] = 1.1
        for i in range(n):                                    # :419-422 frequency rows
            for k in pyrng.sample(range(f), int(f * r * 0.7)):
                noise[i, k, :] = 1.1
        keep, _ = _unstructured(noise.reshape(n, cfg.audio_tokens), len_keep(cfg.audio_tokens, r))
        for i in range(n):
            a_group[idx_a[i]] = g
            a_keep[int(idx_a[i])] = keep[i].clone()
        nv = torch.rand(n, cfg.frames, cfg.video_tokens, generator=gen)
        keepv, _ = _unstructured(nv, len_keep(cfg.video_tokens, r))
        for i in range(n):
            v_group[idx_v[i]] = g
            v_keep[int(idx_v[i])] = [keepv[i, tt].clone() for tt in range(cfg.frames)]
    return ContrastivePlan(a_group, v_group, a_keep, v_keep)


def plan_to_arrays(plan):
    """Flatten a plan into plain int64 tensors (for .npz fixtures)."""
    if isinstance(plan, MaePlan):
        return {"ids_keep_a": plan.ids_keep_a, "ids_restore_a": plan.ids_restore_a,
                "ids_keep_v": plan.ids_keep_v, "ids_restore_v": plan.ids_restore_v}
    B = plan.batch
    T = len(plan.v_keep[0])
    la = max(k.numel() for k in plan.a_keep)
    lv = max(k.numel() for fr in plan.v_keep for k in fr)
    a_pad = torch.full((B, la), -1, dtype=torch.int64)
    v_pad = torch.full((B, T, lv), -1, dtype=torch.int64)
    for b in range(B):
        a_pad[b, :plan.a_keep[b].numel()] = plan.a_keep[b]
        for t in range(T):
            v_pad[b, t, :plan.v_keep[b][t].numel()] = plan.v_keep[b][t]
    return {"a_group": plan.a_group, "v_group": plan.v_group, "a_keep": a_pad, "v_keep": v_pad}


def plan_from_arrays(d):
    d = {k: torch.as_tensor(v) for k, v in d.items()}
    if "ids_keep_a" in d:
        return MaePlan(d["ids_keep_a"].long(), d["ids_restore_a"].long(), d["ids_keep_v"].long(), d["ids_restore_v"].long())
    B, T = d["v_keep"].shape[:2]
    a_keep = [d["a_keep"][b][d["a_keep"][b] >= 0].long() for b in range(B)]
    v_keep = [[d["v_keep"][b, t][d["v_keep"][b, t] >= 0].long() for t in range(T)] for b in range(B)]
    return ContrastivePlan(d["a_group"].long(), d["v_group"].long(), a_keep, v_keep)
