"""Shared test helpers: golden-fixture loading and gradient statistics."""
import json
import os
import zlib

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
GOLDEN = os.path.join(ROOT, "tests", "golden")


def record_margin(test, **values):
    """Log a measured parity margin (worst cosine, norm ratio, loss error ...) into gpurun_out/parity_margins.json, from which
    profiles/rNN/parity_margins.json is committed: the asserts' tolerances are set at ~3x what is recorded here."""
    path = os.path.join(ROOT, "gpurun_out", "parity_margins.json")
    try:
        os.makedirs(os.path.dirname(path), exist_ok=True)
        data = {}
        if os.path.exists(path):
            with open(path) as f:
                data = json.load(f)
        data.setdefault(test, {}).update({k: (float(v) if isinstance(v, (int, float, np.floating)) else v) for k, v in values.items()})
        with open(path, "w") as f:
            json.dump(data, f, indent=1, sort_keys=True)
    except Exception:                                            # a log, never a gate
        pass


def load_golden(name):
    d = np.load(os.path.join(GOLDEN, name + ".npz"), allow_pickle=False)
    return {k: d[k] for k in d.files}


def golden_plan(d):
    """The mask plan a golden case recorded: a MaePlan / ContrastivePlan, or {'mae':..., 'contrastive':...} for the
    combined forward (keys planm_* / planc_*)."""
    from avsiam_amd.maskplan import plan_from_arrays
    if any(k.startswith("planm_") for k in d):
        return {"mae": plan_from_arrays({k[len("planm_"):]: v for k, v in d.items() if k.startswith("planm_")}),
                "contrastive": plan_from_arrays({k[len("planc_"):]: v for k, v in d.items() if k.startswith("planc_")})}
    return plan_from_arrays({k[len("plan_"):]: v for k, v in d.items() if k.startswith("plan_")})


def sample_positions(name, numel, k=8):
    h = zlib.crc32(name.encode())
    return [((h + 1) * (i + 1) * 2654435761) % numel for i in range(k)]


def golden_grads(d):
    names = json.loads(str(d["grad_names"]))
    none = json.loads(str(d["grad_none"]))
    return names, none, d["grad_sum"], d["grad_l2"], d["grad_samples"]


def check_grads_against_golden(d, grads, rel_l2=1e-4, abs_samples=None):
    """grads: {name: tensor or None}.  Checks liveness set, L2 norms and the 8 sampled elements."""
    names, none, gsum, gl2, gsamp = golden_grads(d)
    got_live = sorted(n for n, g in grads.items() if g is not None)
    assert got_live == sorted(names), (set(got_live) ^ set(names))
    worst = 0.0
    for i, n in enumerate(names):
        g = grads[n].detach().double().reshape(-1)
        l2 = g.norm().item()
        assert abs(l2 - gl2[i]) <= rel_l2 * max(gl2[i], 1e-12) + 1e-9, (n, l2, gl2[i])
        samp = np.array([g[j].item() for j in sample_positions(n, g.numel())])
        tol = (abs_samples if abs_samples is not None else rel_l2 * 10) * max(gl2[i] / np.sqrt(g.numel()), 1e-12) + 1e-9
        err = np.abs(samp - gsamp[i]).max()
        assert err <= tol * 10, (n, err, tol)
        worst = max(worst, abs(l2 - gl2[i]) / max(gl2[i], 1e-12))
    return worst


def gpu_grads_vs_golden(d, grad_of, tag, l2_rel, samp_rel, sum_rel):
    """bf16 HIP gradients against EVERYTHING the reference golden stores per live tensor (oracle/gen_golden.py:40-52): the L2
    norm, the 8 sampled elements and the element sum.  Errors are normalised so one tolerance serves all tensors:
      l2   |norm - norm_ref| / norm_ref
      samp max |g[i] - ref[i]| / rms_ref          (rms_ref = norm_ref / sqrt(numel): the scale of one element)
      sum  |sum - sum_ref| / norm_ref              (independent element errors e*rms add up to e*norm over the tensor)
    grad_of(name) -> tensor or None.  Records the worst values (record_margin) and asserts the given tolerances."""
    names, none, gsum, gl2, gsamp = golden_grads(d)
    worst = {"l2": (0.0, None), "samp": (0.0, None), "sum": (0.0, None)}
    for i, n in enumerate(names):
        g = grad_of(n)
        assert g is not None, n
        g = g.detach().double().reshape(-1).cpu()
        ref = max(gl2[i], 1e-30)
        if gl2[i] == 0:
            assert float(g.norm()) < 1e-6, n
            continue
        rms = ref / np.sqrt(g.numel())
        e = {"l2": abs(float(g.norm()) - ref) / ref,
             "samp": float(np.abs(np.array([g[j].item() for j in sample_positions(n, g.numel())]) - gsamp[i]).max()) / rms,
             "sum": abs(float(g.sum()) - gsum[i]) / ref}
        for k, val in e.items():
            if val > worst[k][0]:
                worst[k] = (val, n)
    for n in none:
        g = grad_of(n)
        assert g is None or float(g.abs().max()) == 0.0, f"{n}: dead parameter received a gradient"
    record_margin(tag, **{f"grad_{k}_err": v[0] for k, v in worst.items()}, **{f"grad_{k}_tensor": v[1] for k, v in worst.items()})
    assert worst["l2"][0] <= l2_rel, ("l2", worst["l2"])
    assert worst["samp"][0] <= samp_rel, ("samples", worst["samp"])
    assert worst["sum"][0] <= sum_rel, ("sum", worst["sum"])
    return worst


FT_CASES = ["ft_audio", "ft_audio_eval", "ft_video", "ft_retrieval", "ft_mm", "ft_mm_eval"]


def ft_case_inputs(d, cfg):
    """Inputs of a fine-tuned-model golden case (oracle/gen_golden_ft.py::ft_inputs): a [B,1024,128], v [B,T,3,224,224]."""
    import dataclasses
    from avsiam_amd.weights import synth_inputs
    B, T = int(d["batch"]), int(d["frames"])
    a, v = synth_inputs(dataclasses.replace(cfg, frames=T), B, int(d["input_seed"]))
    return a, (v.unsqueeze(1) if T == 1 else v)


def ft_outputs_as_dict(d, out):
    """Name the outputs of CAVMAEFT_BASE.forward the way the golden files do."""
    if str(d["mode"]) == "retrieval":
        return {"tokens_a": out[0], "tokens_v": out[1]}
    if isinstance(out, tuple):
        return dict(zip(("out", "out_a", "out_v"), out))
    return {"out": out}


from avsiam_amd.comm import HostStagedComm  # noqa: E402,F401  (gloo through the host: several ranks on the one GPU of the test box)


class _Waited:
    """a handle whose collective has already completed (test doubles of the comm interface)"""

    def wait(self):
        pass


def correlated_av_batch(cfg, B, seed, rank=8, strength=2.0, shuffle_pairs=False):
    """AudioSet-shaped synthetic pairs WITH audio<->visual correspondence (a contrastive objective can learn them; i.i.d. Gaussian inputs
    - weights.synth_inputs - cannot be told apart after the token mean): every clip has a latent z ~ N(0, I_rank); its spectrogram is
    noise + strength * sum_k z_k E_k with E_k a fixed spectral envelope (constant over time, so every audio token carries it), its frame(s)
    noise + strength * sum_k z_k T_k with T_k a fixed 16 x 16 x 3 texture tiled over the image (so every visual token carries it).
    The envelopes / textures depend only on `cfg`; `seed` draws z and the noise.  shuffle_pairs: the frames of clip i are paired with the
    audio of another clip's latent - the SAME marginals without the correspondence (what a test that can fail is compared with).
    -> a [B, audio_len, n_mels], v [B, (T,) 3, H, W] fp32."""
    g0 = torch.Generator().manual_seed(1234567)
    env = torch.randn(rank, cfg.n_mels, generator=g0)                                    # spectral envelopes
    tex = torch.randn(rank, cfg.in_chans, cfg.patch, cfg.patch, generator=g0)          # textures
    g = torch.Generator().manual_seed(1000 + int(seed))
    z = torch.randn(B, rank, generator=g) / rank ** 0.5
    zv = z[torch.randperm(B, generator=g)] if shuffle_pairs else z
    a = torch.randn(B, cfg.audio_len, cfg.n_mels, generator=g) + strength * (z @ env)[:, None, :]
    reps = cfg.img_size // cfg.patch
    timg = (zv @ tex.reshape(rank, -1)).reshape(B, cfg.in_chans, cfg.patch, cfg.patch).repeat(1, 1, reps, reps)
    if cfg.frames == 1:
        v = torch.randn(B, cfg.in_chans, cfg.img_size, cfg.img_size, generator=g) + strength * timg
    else:
        v = torch.randn(B, cfg.frames, cfg.in_chans, cfg.img_size, cfg.img_size, generator=g) + strength * timg[:, None]
    return a.contiguous(), v.contiguous()
