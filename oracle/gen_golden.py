"""Generate golden vectors by running the UNMODIFIED reference model on CPU (build container only).

    python oracle/gen_golden.py            # writes tests/golden/*.npz

What is stored (data only - no reference source, no weights): the input seed, the mask plan the
reference drew (recorded, see ref_import.PlanRecorder), the 8-tuple it returned, contrastive logits,
prediction checksums + samples, and per-parameter gradient statistics {sum, L2, 8 sampled elements}
plus the list of parameters whose grad is None.  Weights are re-synthesised from
(seed, name) by avsiam_amd.weights on every machine.

Cases: La=512, Lv=196, T=1 (the only shape the unmodified reference runs).
"""
import json
import os
import random
import sys
import zlib

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

from avsiam_amd.config import AVSiamConfig                      # noqa: E402
from avsiam_amd.maskplan import ContrastivePlan, MaePlan, group_sizes, plan_to_arrays   # noqa: E402
from avsiam_amd.param_spec import state_dict_keys, alias_of    # noqa: E402
from avsiam_amd.weights import synth_inputs, synth_state, synth_vit_checkpoint       # noqa: E402
from oracle import ref_import                                  # noqa: E402

GOLDEN = os.path.join(ROOT, "tests", "golden")
WEIGHT_SEED = 1234


def sample_positions(name, numel, k=8):
    h = zlib.crc32(name.encode())
    return [((h + 1) * (i + 1) * 2654435761) % numel for i in range(k)]


def grad_stats(named_grads):
    names, sums, l2s, samples, none = [], [], [], [], []
    for n, g in named_grads:
        if g is None:
            none.append(n)
            continue
        g = g.detach().double().reshape(-1)
        names.append(n)
        sums.append(g.sum().item())
        l2s.append(g.norm().item())
        samples.append([g[i].item() for i in sample_positions(n, g.numel())])
    return {"grad_names": np.array(json.dumps(names)), "grad_sum": np.array(sums), "grad_l2": np.array(l2s),
            "grad_samples": np.array(samples), "grad_none": np.array(json.dumps(none))}


def full_state(cfg, seed, mode):
    st = synth_state(cfg, seed, mode)
    return {k: st[alias_of(k)] for k in state_dict_keys(cfg)}


def plan_from_recorder(rec, cfg, B, which, calls=None, perms=None):
    calls = rec.calls if calls is None else calls
    perms = rec.perms if perms is None else perms
    if which == "both":
        # the combined forward (:694-739) runs forward_encoder first - two masking calls and two UNUSED randperms (:465-470) -
        # then forward_encoder_mmixed
        return {"mae": plan_from_recorder(rec, cfg, B, "mae", calls[:2], perms[:2]),
                "contrastive": plan_from_recorder(rec, cfg, B, "contrastive", calls[2:], perms[2:])}
    if which == "mae":
        (_, ka, ra), (_, kv, rv) = calls
        return MaePlan(ka, ra, kv.unsqueeze(1), rv.unsqueeze(1))
    perm_a, perm_v = perms[0], perms[1]
    rec = type("R", (), {"calls": calls})
    sizes = group_sizes(B, cfg.n_groups)
    a_group = torch.zeros(B, dtype=torch.int64)
    v_group = torch.zeros(B, dtype=torch.int64)
    a_keep, v_keep = [None] * B, [None] * B
    off = 0
    for g, n in enumerate(sizes):
        (ks, ka, _), (ku, kv, _) = rec.calls[2 * g], rec.calls[2 * g + 1]
        assert ks == "s" and ku == "u"
        for i in range(n):
            ba, bv = int(perm_a[off + i]), int(perm_v[off + i])
            a_group[ba], v_group[bv] = g, g
            a_keep[ba] = ka[i].clone()
            v_keep[bv] = [kv[i].clone()]
        off += n
    return ContrastivePlan(a_group, v_group, a_keep, v_keep)


COMBINED_WEIGHTS = (3.0, 0.01)       # run_cavmae_pretrain_base.py defaults (--mae_loss_weight 3.0 --contrast_loss_weight 0.01): what validate() passes


def run_case(model, cfg, which, B, input_seed, constant=None, rank=0):
    a, v = synth_inputs(cfg, B, input_seed, constant)
    torch.manual_seed(1000 + input_seed)
    random.seed(2000 + input_seed)
    cap = {}
    orig_fc, orig_fd = model.forward_contrastive, model.forward_decoder

    def fc(ar, vr, bidirect_contrast=False):
        cap["rep_a"], cap["rep_v"] = ar.detach().clone(), vr.detach().clone()
        return orig_fc(ar, vr, bidirect_contrast=bidirect_contrast)

    def fd(*args, **kw):
        pa, pv = orig_fd(*args, **kw)
        cap["pred_a"], cap["pred_v"] = pa.detach().clone(), pv.detach().clone()
        return pa, pv

    model.forward_contrastive, model.forward_decoder = fc, fd
    model.zero_grad(set_to_none=True)
    with ref_import.PlanRecorder(model) as rec:
        if which == "mae":
            out = model(a, v, 0.75, 0.75, mae_loss_weight=1, contrast_loss_weight=0)
        elif which == "both":
            out = model(a, v, 0.75, 0.75, mae_loss_weight=COMBINED_WEIGHTS[0], contrast_loss_weight=COMBINED_WEIGHTS[1])
        else:
            out = model(a, v, 0.75, 0.75, mae_loss_weight=0, contrast_loss_weight=1)
    out[0].backward()
    del model.forward_contrastive, model.forward_decoder
    plan = plan_from_recorder(rec, cfg, B, which)
    d = {"which": np.array(which), "batch": np.array(B), "input_seed": np.array(input_seed),
         "constant": np.array(np.nan if constant is None else constant), "weight_seed": np.array(WEIGHT_SEED),
         "rank": np.array(rank)}
    if which == "both":
        for k, t in plan_to_arrays(plan["mae"]).items():
            d["planm_" + k] = t.numpy()
        for k, t in plan_to_arrays(plan["contrastive"]).items():
            d["planc_" + k] = t.numpy()
        d["loss_weights"] = np.array(COMBINED_WEIGHTS)
        assert out[5] is None and out[6] is None                  # the mixed encoder's None masks win (:722)
    else:
        for k, t in plan_to_arrays(plan).items():
            d["plan_" + k] = t.numpy()
    d["out_scalars"] = np.array([out[i].item() for i in (0, 1, 2, 3, 4, 7)], dtype=np.float64)
    if which == "mae":
        d["mask_a"], d["mask_v"] = out[5].numpy(), out[6].numpy()
    if which in ("mae", "both"):
        for k in ("pred_a", "pred_v"):
            p = cap[k].double().reshape(-1)
            d[k + "_sum"] = np.array(p.sum().item())
            d[k + "_l2"] = np.array(p.norm().item())
            d[k + "_samples"] = np.array([p[i].item() for i in sample_positions(k, p.numel(), 64)])
    if which != "mae":
        ra = torch.nn.functional.normalize(cap["rep_a"], dim=-1)
        rv = torch.nn.functional.normalize(cap["rep_v"], dim=-1)
        d["logits"] = (torch.mm(ra, rv.t()) / 0.05).numpy()
        d["rep_a"], d["rep_v"] = cap["rep_a"].numpy(), cap["rep_v"].numpy()
    seen, named = set(), []
    for n, p in model.named_parameters():          # named_parameters de-duplicates the my_blocks alias
        named.append((alias_of(n), p.grad))
    d.update(grad_stats(named))
    return d


def _worker(rank, world, port, cases, outdir):
    import torch.distributed as dist
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    torch.set_num_threads(max(1, 8 // world))
    cfg = AVSiamConfig()
    model = ref_import.build_reference_model()
    missing = model.load_state_dict(full_state(cfg, WEIGHT_SEED, "random"), strict=True)
    assert not missing.missing_keys and not missing.unexpected_keys
    model.train()
    for name, which, B, seed, const in cases:
        d = run_case(model, cfg, which, B, seed + rank, const, rank)
        suffix = f"_r{rank}" if world > 1 else ""
        np.savez_compressed(os.path.join(outdir, f"{name}{suffix}.npz"), **d)
        if rank == 0:
            print(name, which, "B", B, "W", world, "scalars", d["out_scalars"], flush=True)
    dist.barrier()
    dist.destroy_process_group()


def main():
    if not ref_import.reference_available():
        print("reference not present - nothing to generate")
        return
    os.makedirs(GOLDEN, exist_ok=True)
    single = [("c_w1_b4", "contrastive", 4, 87, None),
              ("m_w1_b4", "mae", 4, 87, None),
              ("m_w1_b2_const", "mae", 2, 87, 0.01),
              ("mc_w1_b4", "both", 4, 91, None),
              # the 5-way chunk partition at its other two shapes (cav_mae_base.py:540-570): B = 5 -> five groups of ONE sample,
              # B = 10 -> five groups of two (B in {4, 5, 10})
              ("c_w1_b5", "contrastive", 5, 93, None),
              ("c_w1_b10", "contrastive", 10, 95, None)]
    only = [a for a in sys.argv[1:] if not a.startswith("-")]
    if only:                                    # `python -m oracle.gen_golden c_w1_b5 c_w1_b10`: add cases without touching the others
        _worker(0, 1, 29611, [c for c in single if c[0] in only], GOLDEN)
        return
    _worker(0, 1, 29611, single, GOLDEN)
    import torch.multiprocessing as mp
    multi = [("c_w2_b3", "contrastive", 3, 87, None)]
    mp.spawn(_worker, args=(2, 29612, multi, GOLDEN), nprocs=2, join=True)
    # schema pin: the reference's own state-dict key list must equal ours
    model = ref_import.build_reference_model()
    keys = list(model.state_dict().keys())
    assert sorted(keys) == sorted(state_dict_keys(AVSiamConfig())), "state-dict schema mismatch"
    shapes = {k: list(v.shape) for k, v in model.state_dict().items()}
    eps = {n: m.eps for n, m in model.named_modules() if isinstance(m, torch.nn.LayerNorm)}
    with open(os.path.join(GOLDEN, "schema.json"), "w") as f:
        json.dump({"n_keys": len(keys), "n_params": sum(p.numel() for p in model.parameters()),
                   "shapes": shapes, "ln_eps": eps}, f)
    print("schema ok:", len(keys), "keys")
    pretrained_init_golden()


VIT_SEED = 4242


def pretrained_init_golden():
    """Row 8(f)1: run the reference CONSTRUCTOR (cav_mae_base.py:236-307) with its torch.load of the hard-coded checkpoint path
    answered by a synthetic timm-shaped state dict, and record a checksum of every tensor it derives from it."""
    import torch as _t
    cfg = AVSiamConfig()
    ckpt = synth_vit_checkpoint(cfg, VIT_SEED)
    mod = ref_import.import_reference_module()
    real_load = _t.load

    def fake_load(path, *a, **k):
        if isinstance(path, str) and path.startswith("/mnt/"):
            return {k2: v.clone() for k2, v in ckpt.items()}
        return real_load(path, *a, **k)

    _t.load = fake_load
    try:
        import contextlib
        import io
        with contextlib.redirect_stdout(io.StringIO()):
            model = mod.CAVMAE_BASE()
    finally:
        _t.load = real_load
    derived = ("vit_base.", "ast_base.", "mm_layer_1.", "mm_layer_2.", "my_patch_embed.", "my_patch_embed_a.", "my_blocks.")
    zero = ("decoder_pos_embed_a", "decoder_pos_embed_v", "mask_token", "decoder_modality_a", "decoder_modality_v")
    rec = {}
    for k, v in model.state_dict().items():
        if k.startswith(derived) or k in zero:
            t = v.detach().contiguous()
            rec[k] = {"crc32": zlib.crc32(t.numpy().tobytes()), "sum": float(t.double().sum()), "l2": float(t.double().norm()),
                      "shape": list(t.shape)}
    with open(os.path.join(GOLDEN, "pretrained_init.json"), "w") as f:
        json.dump({"vit_seed": VIT_SEED, "n_checkpoint_keys": len(ckpt), "tensors": rec}, f)
    print("pretrained-init golden:", len(rec), "tensors from", len(ckpt), "checkpoint keys")


if __name__ == "__main__":
    main()
