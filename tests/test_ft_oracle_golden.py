"""Fine-tuned model (CAVMAEFT_BASE) inference modes: the oracle's restatement (oracle/ref_cpu.py::ft_forward) against
golden vectors the unmodified reference produced (oracle/gen_golden_ft.py), and the 553-key schema.
fp32 vs fp32: logits rel 1e-4 / abs 1e-4 (summation order only)."""
import json
import os

import numpy as np
import pytest
import torch

from avsiam_amd.config import AVSiamConfig
from avsiam_amd.param_spec import build_spec_ft, state_dict_keys_ft, alias_of
from avsiam_amd.weights import synth_state_ft
from oracle import ref_cpu
from tests.helpers import FT_CASES, ft_case_inputs, ft_outputs_as_dict, load_golden, sample_positions


@pytest.mark.parametrize("name", FT_CASES)
def test_ft_oracle_matches_reference(name):
    torch.set_num_threads(8)
    d = load_golden(name)
    cfg = AVSiamConfig()
    a, v = ft_case_inputs(d, cfg)
    P = synth_state_ft(cfg, int(d["label_dim"]), int(d["weight_seed"]), "random")
    with torch.no_grad():
        out = ref_cpu.ft_forward(P, cfg, a, v, str(d["mode"]), bool(d["is_eval"]))
    for k, t in ft_outputs_as_dict(d, out).items():
        if k.startswith("tokens"):
            assert tuple(t.shape) == tuple(d[k + "_shape"])
            p = t.double().reshape(-1)
            assert abs(p.norm().item() - d[k + "_l2"]) <= 1e-5 * d[k + "_l2"]
            s = np.array([p[i].item() for i in sample_positions(k, p.numel(), 256)])
            np.testing.assert_allclose(s, d[k + "_samples"], rtol=1e-4, atol=1e-4)
        else:
            assert tuple(t.shape) == d[k].shape
            np.testing.assert_allclose(t.numpy(), d[k], rtol=1e-4, atol=1e-4)


def test_ft_schema_matches_reference(golden_dir):
    with open(os.path.join(golden_dir, "ft_schema.json")) as f:
        sch = json.load(f)
    cfg = AVSiamConfig()
    L = 527
    keys = state_dict_keys_ft(cfg, L)
    assert keys == sch["keys"] and len(keys) == sch["n_keys"] == 553          # same keys in the same order
    spec = {s.name: s for s in build_spec_ft(cfg, L)}
    for k in keys:
        assert list(spec[alias_of(k)].shape) == sch["shapes"][k], k
    # the product model exposes exactly this schema (constructible without a GPU; forward is not)
    from avsiam_amd.models import CAVMAEFT_BASE
    m = CAVMAEFT_BASE(L)
    sd = m.state_dict()
    assert list(sd.keys()) == sch["keys"]
    assert all(list(sd[k].shape) == sch["shapes"][k] for k in keys)
    # constructor identities of the reference (:770-774,790-822): modality norms are copies, fusion blocks = blocks 10 / 11
    assert torch.equal(sd["vit_base.blocks.3.norm1_a.weight"], sd["vit_base.blocks.3.norm1_v.weight"])
    assert torch.equal(sd["mm_layer_1.mlp.fc1.weight"], sd["vit_base.blocks.10.mlp.fc1.weight"])
    assert torch.equal(sd["mm_layer_2.mlp.fc1.weight"], sd["vit_base.blocks.11.mlp.fc1.weight"])
    assert torch.equal(sd["vit_base.patch_embed_a.proj.weight"], sd["vit_base.patch_embed.proj.weight"].mean(dim=1, keepdim=True))
    # a checkpoint saved from a DDP-wrapped reference model loads (module. prefix, traintest_ft_base.py:255)
    m.load_state_dict({"module." + k: v.clone() for k, v in sd.items()})
    with pytest.raises(Exception, match="needs a GPU"):
        m(torch.zeros(1, 1024, 128), None, "audioonly")
    assert m(None, None, "joint_av") is None                                   # unknown modes fall through, as in the reference
