"""ORACLE - CPU restatement of the AVSiam pre-training hot path (test infrastructure, NOT product).

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this file.  The
product path (avsiam_amd/) never imports it and raises if the HIP extension is missing.

Pure fp32 PyTorch, no timm: restates ``CAVMAE_BASE.forward`` of
/root/reference/src/models/cav_mae_base.py with the mask plan as an explicit input (the reference draws
it from global RNGs, :365-439).  Parameters are a plain {reference state-dict name: tensor} mapping.
Pinned against the reference itself: oracle/gen_golden.py runs the unmodified cav_mae_base.py on CPU
(import recipe in oracle/ref_shim/) and commits outputs under tests/golden/; tests/test_oracle_golden.py
checks this file against them.  Third-party boundary: timm==0.9.5 (requirements.txt:101) is absent from
/root/reference; its surface used by the path (Mlp = fc1 -> exact GELU -> fc2; ViT final-norm eps 1e-6;
norm_pre = Identity) is restated here and in the shim - parity at that boundary is unpinned.

Generalisations beyond the reference (no reference oracle exists for them; SURVEY.md section 8(a) note):
T frames per sample (frames folded into the visual batch), arbitrary audio token count, ViT-L dims.
"""
import math

import torch
import torch.nn.functional as F

LN_EPS_BLOCK = 1e-5      # nn.LayerNorm default inside the reference's Block (:120-122,135-137)
LN_EPS_FINAL = 1e-6      # timm ViT norm_layer eps: vit_base.norm / norm_a (:299) [timm boundary]


def _ln(x, P, name, eps):
    return F.layer_norm(x, (x.shape[-1],), P[name + ".weight"], P[name + ".bias"], eps)


def attention(x, P, prefix, heads):
    """Attention.forward, cav_mae_base.py:58-83 (fused SDPA path == softmax(q k^T / sqrt(hd)) v)."""
    B, N, C = x.shape
    hd = C // heads
    qkv = F.linear(x, P[prefix + ".qkv.weight"], P[prefix + ".qkv.bias"])
    qkv = qkv.reshape(B, N, 3, heads, hd).permute(2, 0, 3, 1, 4)
    q, k, v = qkv.unbind(0)
    attn = (q * hd ** -0.5) @ k.transpose(-2, -1)
    attn = attn.softmax(dim=-1)
    x = (attn @ v).transpose(1, 2).reshape(B, N, C)
    return F.linear(x, P[prefix + ".proj.weight"], P[prefix + ".proj.bias"])


def mlp(x, P, prefix):
    """timm.layers.mlp.Mlp (passed at :259,327): fc2(GELU_erf(fc1(x))) [timm boundary]."""
    x = F.linear(x, P[prefix + ".fc1.weight"], P[prefix + ".fc1.bias"])
    x = F.gelu(x)
    return F.linear(x, P[prefix + ".fc2.weight"], P[prefix + ".fc2.bias"])


def block(x, P, prefix, heads, modality=None):
    """Block.forward, :149-193: pre-LN block with modality-selected LayerNorms."""
    sfx = "" if modality is None else "_" + modality
    x = x + attention(_ln(x, P, f"{prefix}.norm1{sfx}", LN_EPS_BLOCK), P, prefix + ".attn", heads)
    x = x + mlp(_ln(x, P, f"{prefix}.norm2{sfx}", LN_EPS_BLOCK), P, prefix + ".mlp")
    return x


def patch_embed(img, w, b, stride=0):
    """PatchEmbed.forward :98-99 - conv k=s=patch, flatten(2).transpose(1,2).  stride (config.stride, e.g. 14 on 16 x 16 storage):
    the stride x stride corner of the stored kernel is the convolution kernel, the rest of it is dead weight."""
    s = stride or w.shape[-1]
    return F.conv2d(img, w[..., :s, :s], b, stride=s).flatten(2).transpose(1, 2)


def embed_audio(P, a, stride=0):
    """:444-450 / :511-517.  a [B, time, mel] -> [B,1,mel,time] -> tokens f*t_patches + t.
    ``a + norm_pre_a(a)`` with norm_pre = Identity doubles the embedding."""
    a = a.unsqueeze(1).transpose(2, 3)
    a = patch_embed(a, P["vit_base.patch_embed_a.proj.weight"], P["vit_base.patch_embed_a.proj.bias"], stride)
    a = a + P["vit_base.pos_embed_a"]
    return a + a


def embed_video(P, v, stride=0):
    """:453-455 / :520-522.  v [N,3,H,W]."""
    v = patch_embed(v, P["vit_base.patch_embed.proj.weight"], P["vit_base.patch_embed.proj.bias"], stride)
    v = v + P["vit_base.pos_embed"][:, 1:]
    return v + v


def _fold_frames(imgs):
    """[B,3,H,W] -> ([B,3,H,W], 1);  [B,T,3,H,W] -> ([(b t),3,H,W], T)  (fold as :857,903)."""
    if imgs.dim() == 4:
        return imgs, 1
    B, T = imgs.shape[:2]
    return imgs.reshape(B * T, *imgs.shape[2:]), T


def _gather(x, ids):
    return torch.gather(x, 1, ids.unsqueeze(-1).expand(-1, -1, x.shape[-1]))


def forward_encoder_mae(P, cfg, a, v, plan):
    """forward_encoder :441-504 with the 75 % unstructured plan supplied."""
    a = embed_audio(P, a, cfg.stride)
    vv, T = _fold_frames(v)
    vv = embed_video(P, vv, cfg.stride)
    B = a.shape[0]
    a = _gather(a, plan.ids_keep_a)
    vv = _gather(vv, plan.ids_keep_v.reshape(B * T, -1))
    for i in range(cfg.depth):
        vv = block(vv, P, f"vit_base.blocks.{i}", cfg.num_heads, "v")       # :487
        a = block(a, P, f"ast_base.blocks.{i}", cfg.num_heads, None)        # :489 separate audio tower
    cv = _ln(vv, P, "vit_base.norm", LN_EPS_FINAL)                          # :492
    ca = _ln(a, P, "ast_base.norm_a", LN_EPS_FINAL)                         # :495
    cv = cv.reshape(B, T * cv.shape[1], cv.shape[2])
    return torch.cat((ca, cv), dim=1)                                       # :503


def forward_encoder_mmixed(P, cfg, a, v, plan):
    """forward_encoder_mmixed :508-594: multi-ratio groups through the shared (Siamese) blocks,
    final norm + token mean.  Samples with equal kept length are batched; results are returned in
    natural sample order (the reference's inverse permutation :575-590 does the same)."""
    a = embed_audio(P, a, cfg.stride)
    vv, T = _fold_frames(v)
    vv = embed_video(P, vv, cfg.stride)
    B = a.shape[0]
    vv = vv.reshape(B, T, vv.shape[1], vv.shape[2])
    ca = [None] * B
    cv = [None] * B
    for g in sorted(set(plan.a_group.tolist())):
        idx = [b for b in range(B) if int(plan.a_group[b]) == g]
        x = torch.stack([a[b][plan.a_keep[b]] for b in idx])
        for i in range(cfg.depth):
            x = block(x, P, f"vit_base.blocks.{i}", cfg.num_heads, "a")     # :557
        x = _ln(x, P, "vit_base.norm_a", LN_EPS_FINAL).mean(dim=1, keepdim=True)   # :566
        for j, b in enumerate(idx):
            ca[b] = x[j]
    for g in sorted(set(plan.v_group.tolist())):
        idx = [b for b in range(B) if int(plan.v_group[b]) == g]
        x = torch.stack([vv[b, t][plan.v_keep[b][t]] for b in idx for t in range(T)])
        for i in range(cfg.depth):
            x = block(x, P, f"vit_base.blocks.{i}", cfg.num_heads, "v")     # :558
        x = _ln(x, P, "vit_base.norm", LN_EPS_FINAL)                        # :563
        x = x.reshape(len(idx), T * x.shape[1], x.shape[2]).mean(dim=1, keepdim=True)
        for j, b in enumerate(idx):
            cv[b] = x[j]
    return torch.stack(ca), torch.stack(cv)                                 # [B,1,D] each


def forward_decoder(P, cfg, x, plan, T):
    """forward_decoder :597-638."""
    La, Lv = cfg.audio_tokens, cfg.video_tokens
    B = x.shape[0]
    keep_a = plan.ids_keep_a.shape[1]
    keep_v = plan.ids_keep_v.shape[-1]
    x = F.linear(x, P["decoder_embed.weight"], P["decoder_embed.bias"])     # :600
    Dd = x.shape[-1]
    mt = P["mask_token"]
    a_ = torch.cat([x[:, :keep_a], mt.expand(B, La - keep_a, Dd)], dim=1)   # :604-606
    a_ = _gather(a_, plan.ids_restore_a)                                    # :607
    xv = x[:, keep_a:].reshape(B * T, keep_v, Dd)
    v_ = torch.cat([xv, mt.expand(B * T, Lv - keep_v, Dd)], dim=1)          # :610-611
    v_ = _gather(v_, plan.ids_restore_v.reshape(B * T, Lv))                 # :612
    a_ = a_ + P["decoder_pos_embed_a"] + P["decoder_modality_a"]            # :615,625
    v_ = v_ + P["decoder_pos_embed_v"] + P["decoder_modality_v"]            # :616,626
    x = torch.cat([a_, v_.reshape(B, T * Lv, Dd)], dim=1)                   # :617
    for i in range(cfg.dec_depth):
        x = block(x, P, f"decoder_blocks.{i}", cfg.dec_heads, None)         # :629-630
    x = _ln(x, P, "decoder_norm", LN_EPS_BLOCK)                             # :631
    pred_a = F.linear(x[:, :La], P["decoder_pred_a.weight"], P["decoder_pred_a.bias"])   # :634
    pred_v = F.linear(x[:, La:], P["decoder_pred_v.weight"], P["decoder_pred_v.bias"])   # :635
    return pred_a, pred_v


def patchify(imgs, c, h, w, p=16):
    """:343-351  (N,c,H,W) -> (N, h*w, p*p*c) in (p,q,c) order."""
    x = imgs.contiguous().reshape(imgs.shape[0], c, h, p, w, p)
    x = torch.einsum('nchpwq->nhwpqc', x)
    return x.contiguous().reshape(imgs.shape[0], h * w, p * p * c)


def mae_loss(cfg, inp, pred, mask, modality):
    """forward_mae_loss :663-683 (norm_pix_loss branch is commented out in the reference)."""
    p = cfg.st
    c = 1 if modality == 'a' else cfg.in_chans
    if modality == 'a':
        inp = inp.unsqueeze(1).transpose(2, 3)
        target = patchify(inp, 1, inp.shape[2] // p, inp.shape[3] // p, p)
    else:
        vv, T = _fold_frames(inp)
        target = patchify(vv, cfg.in_chans, vv.shape[2] // p, vv.shape[3] // p, p)
    if p != cfg.patch:
        # patch stride on larger patch storage (config.stride): the prediction row keeps patch x patch positions per channel in (p, q, c)
        # order; the stride x stride corner is scored, the rest of the row is dead
        P0 = cfg.patch
        pred = pred.reshape(*pred.shape[:-1], P0, P0, c)[..., :p, :p, :].reshape(*pred.shape[:-1], p * p * c)
    target = target.reshape(pred.shape)
    loss = ((pred - target) ** 2).mean(dim=-1)
    return (loss * mask).sum() / mask.sum()


def contrastive(audio_rep, video_rep, temperature=0.05):
    """forward_contrastive(bidirect_contrast=True) :641-661."""
    audio_rep = F.normalize(audio_rep, dim=-1)
    video_rep = F.normalize(video_rep, dim=-1)
    total = torch.mm(audio_rep, video_rep.t()) / temperature
    n = total.shape[0]
    ar = torch.arange(n)
    nce_1 = -torch.mean(torch.diag(F.log_softmax(total, dim=0)))
    nce_2 = -torch.mean(torch.diag(F.log_softmax(total.t(), dim=0)))
    acc_1 = torch.sum(torch.eq(torch.argmax(F.softmax(total, dim=0), dim=0), ar)) / n
    acc_2 = torch.sum(torch.eq(torch.argmax(F.softmax(total.t(), dim=0), dim=0), ar)) / n
    return (nce_1 + nce_2) / 2, (acc_1 + acc_2) / 2, total


class _GatherLayer(torch.autograd.Function):
    """gather_layer.py:21-37: all_gather forward; all_reduce(SUM) + own slice backward."""

    @staticmethod
    def forward(ctx, x):
        import torch.distributed as dist
        out = [torch.zeros_like(x) for _ in range(dist.get_world_size())]
        dist.all_gather(out, x)
        return tuple(out)

    @staticmethod
    def backward(ctx, *grads):
        import torch.distributed as dist
        g = torch.stack(grads)
        dist.all_reduce(g)
        return g[dist.get_rank()]


def _gather_all(x):
    import torch.distributed as dist
    if dist.is_available() and dist.is_initialized() and dist.get_world_size() > 1:
        return torch.cat(_GatherLayer.apply(x), dim=0)
    return x


def forward(P, cfg, audio, imgs, plan, mae_loss_weight=1., contrast_loss_weight=0.01, extras=None):
    """CAVMAE_BASE.forward :685-741.  ``plan`` is a MaePlan when mae_loss_weight != 0 and a ContrastivePlan
    when contrast_loss_weight != 0 (a dict {'mae':..., 'contrastive':...} when both).  Returns the
    reference 8-tuple; ``extras`` (dict) receives pred_a/pred_v/logits when given."""
    if isinstance(plan, dict):
        plan_m, plan_c = plan.get("mae"), plan.get("contrastive")
    else:
        plan_m = plan if mae_loss_weight != 0 else None
        plan_c = plan if contrast_loss_weight != 0 else None
    zero = torch.tensor(0.0)
    mask_a = mask_v = None
    if mae_loss_weight != 0:
        T = 1 if imgs.dim() == 4 else imgs.shape[1]
        x = forward_encoder_mae(P, cfg, audio, imgs, plan_m)
        x = block(x, P, "mm_layer_1", cfg.num_heads, "a")                   # :699
        x = block(x, P, "mm_layer_2", cfg.num_heads, "a")                   # :700
        pred_a, pred_v = forward_decoder(P, cfg, x, plan_m, T)
        mask_a = plan_m.mask_a()
        mask_v = plan_m.mask_v().reshape(audio.shape[0], -1)
        loss_mae_a = mae_loss(cfg, audio, pred_a, mask_a, 'a')
        loss_mae_v = mae_loss(cfg, imgs, pred_v, mask_v, 'v')
        loss_mae = loss_mae_a + loss_mae_v                                  # :707 (weight is NOT applied, :739)
        if extras is not None:
            extras["pred_a"], extras["pred_v"] = pred_a, pred_v
    else:
        loss_mae_a, loss_mae_v, loss_mae = zero, zero, zero
    if contrast_loss_weight != 0:
        ca, cv = forward_encoder_mmixed(P, cfg, audio, imgs, plan_c)
        ca = _gather_all(ca)                                                # :724
        cv = _gather_all(cv)                                                # :725
        loss_c, c_acc, total = contrastive(ca.mean(dim=1), cv.mean(dim=1), cfg.temperature)   # :729
        loss_c = contrast_loss_weight * loss_c                              # :735
        mask_a = mask_v = None                                              # :594 returns None masks
        if extras is not None:
            extras["logits"] = total
            extras["rep_a"], extras["rep_v"] = ca, cv
    else:
        loss_c, c_acc = zero, zero
    loss = loss_c + loss_mae                                                # :739
    return loss, loss_mae, loss_mae_a, loss_mae_v, loss_c, mask_a, mask_v, c_acc


def adam_step(p, g, m, v, step, lr, beta1=0.95, beta2=0.999, eps=1e-8, weight_decay=5e-7):
    """torch.optim.Adam semantics used by the reference loop (traintest_cavmae_base.py:64-66):
    L2 weight decay folded into the gradient, bias-corrected moments.  In-place on p, m, v."""
    g = g + weight_decay * p
    m.mul_(beta1).add_(g, alpha=1 - beta1)
    v.mul_(beta2).addcmul_(g, g, value=1 - beta2)
    bc1 = 1 - beta1 ** step
    bc2 = 1 - beta2 ** step
    denom = (v.sqrt() / math.sqrt(bc2)).add_(eps)
    p.addcdiv_(m, denom, value=-lr / bc1)


# =====================================================================================================
# CAVMAEFT_BASE inference modes (cav_mae_base.py:827-1035).  P holds the fine-tuned model's state dict.
def _head(x, P, name):
    """nn.Sequential(LayerNorm, Linear) heads, :809-815."""
    return F.linear(_ln(x, P, name + ".0", LN_EPS_BLOCK), P[name + ".1.weight"], P[name + ".1.bias"])


def ft_encode_audio(P, cfg, a):
    """:829-841 - all 512 audio tokens through the shared blocks with the '_a' norms, then norm_a."""
    a = embed_audio(P, a, cfg.stride)
    for i in range(cfg.depth):
        a = block(a, P, f"vit_base.blocks.{i}", cfg.num_heads, "a")
    return _ln(a, P, "vit_base.norm_a", LN_EPS_FINAL)


def ft_encode_video(P, cfg, v):
    """:851-860 - v [B,T,3,H,W] folded to (b t); '_v' norms, then norm.  -> [(b t), Lv, D]"""
    vv = v.reshape(v.shape[0] * v.shape[1], *v.shape[2:])
    vv = embed_video(P, vv, cfg.stride)
    for i in range(cfg.depth):
        vv = block(vv, P, f"vit_base.blocks.{i}", cfg.num_heads, "v")
    return _ln(vv, P, "vit_base.norm", LN_EPS_FINAL)


def _ft_fuse(P, cfg, a, vt):
    """:944-952 / :1022-1029 - concat tokens, two fusion blocks with the 'a' norms, per-part means side by side."""
    La = a.shape[1]
    av = torch.cat((a, vt), dim=1)
    av = block(av, P, "mm_layer_1", cfg.num_heads, "a")
    av = block(av, P, "mm_layer_2", cfg.num_heads, "a")
    av = torch.cat((av[:, :La].mean(dim=1), av[:, La:].mean(dim=1)), dim=-1)
    return _head(av, P, "mlp_head_mm")


def ft_forward(P, cfg, a, v, mode, is_eval=False):
    """CAVMAEFT_BASE.forward, :827-1035."""
    if mode == "audioonly":
        out = _head(ft_encode_audio(P, cfg, a).mean(dim=1), P, "mlp_head_a")
        return out.unsqueeze(1) if is_eval else out                                       # :845-847
    if mode == "videoonly":
        B, T = v.shape[:2]
        x = _head(ft_encode_video(P, cfg, v).mean(dim=1), P, "mlp_head")
        return x.reshape(B, T, -1).squeeze(1)                                             # :865
    if mode == "retrieval":
        B, T = v.shape[:2]
        ta = ft_encode_audio(P, cfg, a)
        tv = ft_encode_video(P, cfg, v)
        return ta, tv.reshape(B, T, *tv.shape[1:])[:, 5]                                  # :892
    if mode == "mm_grad":
        B, T = v.shape[:2]
        ta = ft_encode_audio(P, cfg, a)
        tv = ft_encode_video(P, cfg, v)
        if is_eval:
            tv = tv.reshape(B, T, *tv.shape[1:])
            return torch.stack([_ft_fuse(P, cfg, ta, tv[:, t]) for t in range(10)], dim=1)   # :940-961  [B,10,L]
        out_a = _head(ta.mean(dim=1), P, "mlp_head_a")                                    # :1019
        out_v = _head(tv.mean(dim=1), P, "mlp_head")                                      # :1020
        return _ft_fuse(P, cfg, ta, tv), out_a, out_v                                     # :1035
    return None
