"""Shared test helpers: golden-fixture loading and gradient statistics."""
import json
import os
import zlib

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
GOLDEN = os.path.join(ROOT, "tests", "golden")


def load_golden(name):
    d = np.load(os.path.join(GOLDEN, name + ".npz"), allow_pickle=False)
    return {k: d[k] for k in d.files}


def golden_plan(d):
    from avsiam_amd.maskplan import plan_from_arrays
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
