"""Input normalisation on the device - the per-sample host work of the reference dataloader
(/root/reference/src/dataloader.py:505-513 audio, :461-462 + :152-155 frames) as two HBM-bound kernels, so that raw
AudioSet-shaped tensors can be fed to ``CAVMAE_BASE.forward`` without a CPU pass.  No CPU fallback."""
import ctypes

import numpy as np
import torch

from . import _lib

IMAGENET_DEFAULT_MEAN = (0.485, 0.456, 0.406)
IMAGENET_DEFAULT_STD = (0.229, 0.224, 0.225)


def normalize_fbank(fbank, norm_mean, norm_std, noise=False, seed=0, rng=None, shift=None, amp=None):
    """fbank [B, T, F] fp32 (GPU) -> (fbank - norm_mean) / norm_std (dataloader.py:505-506).  ``noise=True`` adds the
    reference's training augmentation (:510-513): + U[0,1) * amp with amp = rand()/10 per sample, then a roll along time
    by a per-sample shift in [-T, T); amp and shift come from ``rng`` (a numpy Generator, default seeded by ``seed``) unless
    given explicitly (int32 / fp32 device tensors), the per-element noise from a device Philox stream keyed by ``seed``.
    (The training path does not call this: forward(..., input_xf=) applies the same arithmetic inside the kernels that read
    the input; this two-pass form is what tests compare it with.)"""
    if not (fbank.is_cuda and fbank.dtype == torch.float32 and fbank.dim() == 3 and fbank.is_contiguous()):
        raise _lib.AvsiamHipError("normalize_fbank: need a contiguous fp32 [B, T, F] GPU tensor")
    B, T, F = fbank.shape
    out = torch.empty_like(fbank)
    if noise and shift is None:
        rng = rng if rng is not None else np.random.default_rng(seed)
        amp = torch.from_numpy((rng.random(B) / 10).astype(np.float32)).to(fbank.device)
        shift = torch.from_numpy(rng.integers(-T, T, B).astype(np.int32)).to(fbank.device)
    _lib.call("avs_normalize_audio", fbank, out, B, T, F, float(norm_mean), float(norm_std), shift, amp, int(seed), _lib.current_stream())
    return out


def normalize_frames(frames_u8, mean=IMAGENET_DEFAULT_MEAN, std=IMAGENET_DEFAULT_STD):
    """frames [..., 3, H, W] uint8 (GPU) -> fp32 (x / 255 - mean_c) / std_c (dataloader.py:461-462 and my_normalize)."""
    if not (frames_u8.is_cuda and frames_u8.dtype == torch.uint8 and frames_u8.dim() >= 3 and frames_u8.shape[-3] == 3
            and frames_u8.is_contiguous()):
        raise _lib.AvsiamHipError("normalize_frames: need a contiguous uint8 [..., 3, H, W] GPU tensor")
    plane = frames_u8.shape[-1] * frames_u8.shape[-2]
    n = frames_u8.numel() // (3 * plane)
    out = torch.empty(frames_u8.shape, dtype=torch.float32, device=frames_u8.device)
    m3 = (ctypes.c_float * 3)(*[float(x) for x in mean])
    s3 = (ctypes.c_float * 3)(*[float(x) for x in std])
    _lib.call("avs_normalize_frames_u8", frames_u8, out, n, plane, ctypes.cast(m3, ctypes.c_void_p), ctypes.cast(s3, ctypes.c_void_p),
              _lib.current_stream())
    return out
