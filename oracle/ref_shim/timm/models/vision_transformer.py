"""ViT skeleton with the attribute names the reference constructor reads (cav_mae_base.py:236-300)."""
import sys
from functools import partial

import torch
import torch.nn as nn

from ..layers import Mlp, PatchEmbed  # noqa: F401


class Attention(nn.Module):
    def __init__(self, dim, num_heads=8, qkv_bias=False, **kw):
        super().__init__()
        self.num_heads = num_heads
        self.qkv = nn.Linear(dim, dim * 3, bias=qkv_bias)
        self.proj = nn.Linear(dim, dim)


class Block(nn.Module):
    def __init__(self, dim, num_heads, mlp_ratio=4., qkv_bias=True, norm_layer=nn.LayerNorm):
        super().__init__()
        attn_cls = sys.modules[__name__].Attention       # honours the reference's monkey-patch (:230)
        self.norm1 = norm_layer(dim)
        self.attn = attn_cls(dim, num_heads=num_heads, qkv_bias=qkv_bias)
        self.ls1 = nn.Identity()
        self.drop_path1 = nn.Identity()
        self.norm2 = norm_layer(dim)
        self.mlp = Mlp(in_features=dim, hidden_features=int(dim * mlp_ratio))
        self.ls2 = nn.Identity()
        self.drop_path2 = nn.Identity()


class VisionTransformer(nn.Module):
    def __init__(self, img_size=224, patch_size=16, in_chans=3, num_classes=21843, embed_dim=768, depth=12,
                 num_heads=12, mlp_ratio=4.):
        super().__init__()
        norm_layer = partial(nn.LayerNorm, eps=1e-6)
        self.patch_embed = PatchEmbed(img_size, patch_size, in_chans, embed_dim)
        self.cls_token = nn.Parameter(torch.zeros(1, 1, embed_dim))
        self.pos_embed = nn.Parameter(torch.randn(1, self.patch_embed.num_patches + 1, embed_dim) * .02)
        self.pos_drop = nn.Dropout(0.)
        self.patch_drop = nn.Identity()
        self.norm_pre = nn.Identity()
        self.blocks = nn.Sequential(*[Block(embed_dim, num_heads, mlp_ratio, True, norm_layer) for _ in range(depth)])
        self.norm = norm_layer(embed_dim)
        self.fc_norm = nn.Identity()
        self.head_drop = nn.Dropout(0.)
        self.head = nn.Linear(embed_dim, num_classes)


def create_model(name, pretrained=False, **kw):
    assert name.startswith('vit_base_patch16_224'), name
    return VisionTransformer()
