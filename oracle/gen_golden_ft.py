"""Golden vectors of the fine-tuned model's inference modes: runs the UNMODIFIED reference ``CAVMAEFT_BASE`` on CPU
(build container only; import recipe in oracle/ref_import.py).

    python oracle/gen_golden_ft.py         # writes tests/golden/ft_*.npz and ft_schema.json

Stored: data only - the input seed and the outputs of each mode (logits in full; the two token matrices of the retrieval
mode as {sum, L2, 256 sampled elements}).  Weights are re-synthesised from (seed, name) by avsiam_amd.weights.
Cases follow cav_mae_base.py:827-1035: audioonly, videoonly (2 frames), retrieval (6 frames: frame 5 is returned),
mm_grad training-shape (1 frame) and mm_grad evaluation (10 frames, one joint logit row per frame).
"""
import dataclasses
import json
import os
import sys
import zlib

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

from avsiam_amd.config import AVSiamConfig                      # noqa: E402
from avsiam_amd.param_spec import alias_of, state_dict_keys_ft  # noqa: E402
from avsiam_amd.weights import synth_inputs, synth_state_ft     # noqa: E402
from oracle import ref_import                                   # noqa: E402

GOLDEN = os.path.join(ROOT, "tests", "golden")
WEIGHT_SEED = 4321
LABEL_DIM = 527             # AudioSet (egs/audioset/run_base_ft.sh)

#        name            mode         B  T   is_eval
CASES = [("ft_audio",     "audioonly", 2, 1,  False),
         ("ft_audio_eval", "audioonly", 2, 1,  True),
         ("ft_video",     "videoonly", 2, 2,  False),
         ("ft_retrieval", "retrieval", 1, 6,  False),
         ("ft_mm",        "mm_grad",   2, 1,  False),
         ("ft_mm_eval",   "mm_grad",   1, 10, True)]


def ft_inputs(cfg, B, T, seed):
    a, v = synth_inputs(dataclasses.replace(cfg, frames=T), B, seed)
    return a, (v.unsqueeze(1) if T == 1 else v)


def sample_positions(name, numel, k=256):
    h = zlib.crc32(name.encode())
    return [((h + 1) * (i + 1) * 2654435761) % numel for i in range(k)]


def summarise(name, t):
    p = t.detach().double().reshape(-1)
    return {name + "_shape": np.array(t.shape), name + "_sum": np.array(p.sum().item()), name + "_l2": np.array(p.norm().item()),
            name + "_samples": np.array([p[i].item() for i in sample_positions(name, p.numel())])}


def full_state(cfg):
    st = synth_state_ft(cfg, LABEL_DIM, WEIGHT_SEED, "random")
    return {k: st[alias_of(k)] for k in state_dict_keys_ft(cfg, LABEL_DIM)}


def main():
    if not ref_import.reference_available():
        print("reference not present - nothing to generate")
        return
    os.makedirs(GOLDEN, exist_ok=True)
    cfg = AVSiamConfig()
    model = ref_import.build_reference_ft_model(LABEL_DIM)
    keys = list(model.state_dict().keys())
    assert keys == state_dict_keys_ft(cfg, LABEL_DIM), "state-dict schema mismatch"
    res = model.load_state_dict(full_state(cfg), strict=True)
    assert not res.missing_keys and not res.unexpected_keys
    model.eval()
    for name, mode, B, T, is_eval in CASES:
        a, v = ft_inputs(cfg, B, T, 87)
        with torch.no_grad():
            out = model(a, v, mode, is_eval=is_eval)
        d = {"mode": np.array(mode), "batch": np.array(B), "frames": np.array(T), "is_eval": np.array(is_eval), "input_seed": np.array(87),
             "weight_seed": np.array(WEIGHT_SEED), "label_dim": np.array(LABEL_DIM)}
        if mode == "retrieval":
            d.update(summarise("tokens_a", out[0]))
            d.update(summarise("tokens_v", out[1]))
        elif isinstance(out, tuple):
            for k, o in zip(("out", "out_a", "out_v"), out):
                d[k] = o.numpy()
        else:
            d["out"] = out.numpy()
        np.savez_compressed(os.path.join(GOLDEN, name + ".npz"), **d)
        print(name, mode, "B", B, "T", T, {k: (v.shape if hasattr(v, "shape") else v) for k, v in d.items() if k.startswith(("out", "tokens"))}, flush=True)
    shapes = {k: list(v.shape) for k, v in model.state_dict().items()}
    eps = {n: m.eps for n, m in model.named_modules() if isinstance(m, torch.nn.LayerNorm)}
    with open(os.path.join(GOLDEN, "ft_schema.json"), "w") as f:
        json.dump({"n_keys": len(keys), "keys": keys, "shapes": shapes, "ln_eps": eps}, f)
    print("schema ok:", len(keys), "keys")


if __name__ == "__main__":
    main()
