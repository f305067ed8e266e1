"""CPU checks of the host-side logic around the kernels: the mask-plan bookkeeping (group sizes, keep counts, the structured audio masks, the
.npz form of a plan) against the reference's expressions and against the plans the UNMODIFIED reference drew for the golden fixtures, and the
algorithmic FLOP count the roofline is priced with."""
import math
import os
import random

import numpy as np
import pytest
import torch

from avsiam_amd import flops, maskplan
from avsiam_amd.config import AVSiamConfig, vit_huge14, vit_large

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


@pytest.mark.parametrize("batch", list(range(1, 24)) + [64, 65, 69, 70, 512])
def test_group_sizes_are_torch_chunk_sizes(batch):
    """/root/reference/src/models/cav_mae_base.py:534 - `torch.chunk(torch.randperm(B), 5)`: ceil(B / 5)-sized chunks, possibly fewer than 5"""
    want = [c.numel() for c in torch.chunk(torch.arange(batch), 5)]
    assert maskplan.group_sizes(batch, 5) == want and sum(want) == batch


def test_keep_counts_follow_the_reference_expressions():
    """:372 / :399 `int(L * (1 - mask_ratio))` with the ratio `0 + 0.2 * i` of :546 / :549 - in floating point, as the reference evaluates it"""
    for L in (196, 256, 512, 657, 128, 49):
        for g in range(5):
            r = 0 + 0.2 * g
            assert maskplan.group_ratio(g) == r and maskplan.len_keep(L, r) == int(L * (1 - r))
    assert maskplan.len_keep(196, maskplan.group_ratio(3)) == 78 and maskplan.len_keep(512, maskplan.group_ratio(3)) == 204      # (0.6000000000000001)
    cfg = AVSiamConfig()
    assert (cfg.keep_a, cfg.keep_v) == (128, 49) and (cfg.audio_t, cfg.audio_f, cfg.video_tokens) == (64, 8, 196)
    h = vit_huge14()
    assert h.audio_tokens == 657 and h.audio_f == 9 and h.video_tokens == 256 and vit_large().embed_dim == 1024


@pytest.mark.parametrize("name,batch", [("c_w1_b4", 4), ("c_w1_b5", 5), ("c_w1_b10", 10)])
def test_reference_drawn_contrastive_plans_have_the_structure_the_host_code_assumes(name, batch):
    """The golden plans were drawn by the unmodified reference (oracle/gen_golden.py).  Group populations = the chunk sizes of two independent
    batch permutations; every sample keeps exactly len_keep(L, 0.2 g) distinct tokens per modality (and per frame); the .npz form round-trips."""
    d = np.load(os.path.join(GOLD, name + ".npz"))
    plan = maskplan.plan_from_arrays({k[5:]: d[k] for k in d.files if k.startswith("plan_")})
    assert plan.batch == batch
    sizes = maskplan.group_sizes(batch)
    for grp in (plan.a_group, plan.v_group):
        assert sorted(torch.bincount(grp, minlength=len(sizes)).tolist(), reverse=True) == sorted(sizes, reverse=True)
        assert torch.bincount(grp, minlength=len(sizes)).tolist() == sizes
    La, Lv = int(d["plan_a_keep"].shape[1]), int(d["plan_v_keep"].shape[2])
    for b in range(batch):
        ka = plan.a_keep[b]
        assert ka.numel() == maskplan.len_keep(La, maskplan.group_ratio(int(plan.a_group[b]))) and ka.unique().numel() == ka.numel()
        assert int(ka.min()) >= 0 and int(ka.max()) < La
        for kv in plan.v_keep[b]:
            assert kv.numel() == maskplan.len_keep(Lv, maskplan.group_ratio(int(plan.v_group[b]))) and kv.unique().numel() == kv.numel()
    back = maskplan.plan_to_arrays(plan)
    for k in ("a_group", "v_group", "a_keep", "v_keep"):
        assert np.array_equal(np.asarray(back[k]), d["plan_" + k]), k


def test_reference_drawn_mae_plan_is_a_permutation_split():
    d = np.load(os.path.join(GOLD, "m_w1_b4.npz"))
    plan = maskplan.plan_from_arrays({k[5:]: d[k] for k in d.files if k.startswith("plan_")})
    B, La = plan.ids_restore_a.shape
    assert plan.batch == B == 4 and plan.ids_keep_a.shape[1] == int(La * 0.25) and plan.ids_keep_v.shape[-1] == 49
    for b in range(B):
        assert sorted(plan.ids_restore_a[b].tolist()) == list(range(La))                       # a permutation
        # the kept tokens are the ones whose rank in the shuffle is below keep (:379-388): mask 0 there, 1 elsewhere
        assert torch.equal(torch.nonzero(plan.mask_a()[b] == 0).flatten(), plan.ids_keep_a[b].sort().values)
    assert float(plan.mask_a().mean()) == 0.75 and float(plan.mask_v().mean()) == 0.75
    back = maskplan.plan_to_arrays(plan)
    assert all(np.array_equal(np.asarray(back[k]), d["plan_" + k]) for k in back)


def test_host_plan_generator_has_the_reference_distribution_properties():
    """make_contrastive_plan / make_mae_plan (the host generator used where no device draw exists: oracle-side tests, fixtures): structured audio masks
    remove WHOLE time columns and frequency rows first (:415-422: int(t r 0.7) columns, int(f r 0.7) rows get noise 1.1 = sorted last), group 0 keeps
    everything, the MAE plan keeps a quarter."""
    cfg = AVSiamConfig(audio_tokens=512, frames=2)
    gen = torch.Generator().manual_seed(5)
    plan = maskplan.make_contrastive_plan(cfg, 10, gen, random.Random(7))
    t, f = cfg.audio_t, cfg.audio_f
    assert torch.bincount(plan.a_group).tolist() == [2, 2, 2, 2, 2]
    for b in range(10):
        g = int(plan.a_group[b]); r = maskplan.group_ratio(g)
        kept = torch.zeros(cfg.audio_tokens, dtype=torch.bool); kept[plan.a_keep[b]] = True
        grid = kept.reshape(f, t)
        assert int(kept.sum()) == maskplan.len_keep(cfg.audio_tokens, r)
        if g == 0:
            assert bool(kept.all())
        else:
            # the structured picks - c = int(t r 0.7) whole time columns, w = int(f r 0.7) whole frequency rows - are sorted last, i.e. removed first;
            # when they are MORE than the tokens to remove, the surplus (lowest token ids first: stable ties) stays, one token per picked column
            c, w = int(t * r * 0.7), int(f * r * 0.7)
            slack = max(0, c * f + w * (t - c) - (cfg.audio_tokens - int(kept.sum())))
            assert int((~grid.any(0)).sum()) >= c - slack and int((~grid.any(1)).sum()) >= w - slack
            if g == 4:                                                # no surplus at ratio 0.8: exactly the picked columns and rows are empty
                assert (int((~grid.any(0)).sum()), int((~grid.any(1)).sum())) == (c, w) == (35, 4)
        assert len(plan.v_keep[b]) == 2 and all(k.numel() == maskplan.len_keep(196, maskplan.group_ratio(int(plan.v_group[b]))) for k in plan.v_keep[b])
    mae = maskplan.make_mae_plan(cfg, 3, gen)
    assert mae.ids_keep_a.shape == (3, 128) and mae.ids_keep_v.shape == (3, 2, 49)
    assert all(sorted(mae.ids_restore_v[b, tt].tolist()) == list(range(196)) for b in range(3) for tt in range(2))
    # same generator state, same plan (the fixtures depend on it)
    p1 = maskplan.make_mae_plan(cfg, 2, torch.Generator().manual_seed(9)); p2 = maskplan.make_mae_plan(cfg, 2, torch.Generator().manual_seed(9))
    assert torch.equal(p1.ids_keep_a, p2.ids_keep_a) and torch.equal(p1.ids_restore_v, p2.ids_restore_v)


def test_algorithmic_flops_of_the_step():
    """flops.step_flops: 24 N D^2 + 4 N^2 D per block on KEPT tokens, training = 3 x forward except the patch embedding (2 x).
    The values bench.py prints for the headline shape and for the reference's one-frame shapes are pinned; the contrastive pass's count follows the
    group sizes of the batch."""
    assert flops.blk(10, 8) == 24 * 10 * 64 + 4 * 100 * 8
    head = AVSiamConfig(frames=10, audio_tokens=512)
    assert abs(flops.gflop_per_sample(head, 64) - 1856.56426368) < 1e-6                      # profiles/r05/a_bench.json config.gflop_per_sample
    one = AVSiamConfig(frames=1, audio_tokens=512)
    assert abs(flops.gflop_per_sample(one, 4) - 511.573945344) < 1e-6                        # secondary[1].gflop_per_sample
    s = flops.step_flops(head, 64)
    assert s["train_total"] == 3 * (s["fwd_pass1"] + s["fwd_pass2"]) + 2 * s["fwd_embed"]
    # pass 2 by hand: the modality blocks on the kept tokens of each modality (per frame), the joint layers on their union, decoder embedding, the decoder on ALL tokens, the two prediction heads
    D, Dd, T, ka, kv, La, Lv = 768, 512, 10, 128, 49, 512, 196
    n_enc = ka + T * kv
    per = head.depth * (flops.blk(ka, D) + T * flops.blk(kv, D)) + 2 * flops.blk(n_enc, D) + n_enc * 2 * D * Dd + head.dec_depth * flops.blk(La + T * Lv, Dd) \
        + La * 2 * Dd * 256 + T * Lv * 2 * Dd * 768
    assert s["fwd_pass2"] == 64 * per
    # the contrastive pass depends on the batch through its group sizes only: per-sample cost is not constant in B
    assert flops.step_flops(head, 5)["fwd_pass1"] * 2 == flops.step_flops(head, 10)["fwd_pass1"]
    assert math.isclose(flops.gflop_per_sample(head, 64), flops.gflop_per_sample(head, 65), rel_tol=0.02)


def test_parameter_arena_layout():
    """arena.ParamArena: one flat fp32 buffer ordered [pass-1 only | both passes | pass-2 only | dead], so that each pass's gradients, its
    all-reduce and its fused Adam launch cover ONE contiguous range and the parameters the reference never updates (grad None under
    DDP(find_unused_parameters=True): torch.optim.Adam skips them) lie outside both.  Offsets are 64-element aligned; views alias the buffer;
    the bf16 transposed copies exist for the matrices the input-gradient GEMMs read."""
    from avsiam_amd.arena import ALIGN, ParamArena
    from avsiam_amd.param_spec import P1, P2, build_spec, live_names, state_dict_keys
    cfg = AVSiamConfig()
    spec = build_spec(cfg)
    a = ParamArena(cfg)
    (b1, e1), (b2, e2) = a.range[P1], a.range[P2]
    assert b1 == 0 and b1 < b2 < e1 < e2 == a.live_end < a.total                   # the two ranges overlap exactly in the shared parameters
    info = {s.name: s for s in spec}
    for name, off in a.offset.items():
        n = math.prod(info[name].shape)
        assert off % ALIGN == 0
        live = info[name].live
        inside1, inside2 = b1 <= off and off + n <= e1, b2 <= off and off + n <= e2
        assert inside1 == bool(live & P1) and inside2 == bool(live & P2), name
        if live == 0:
            assert off >= a.live_end
    # no two parameters overlap
    spans = sorted((off, off + math.prod(info[n].shape)) for n, off in a.offset.items())
    assert all(x[1] <= y[0] for x, y in zip(spans, spans[1:]))
    assert set(live_names(cfg, P1)) == {s.name for s in spec if s.live & P1} and len(live_names(cfg, P2)) == sum(1 for s in spec if s.live & P2)
    assert len(state_dict_keys(cfg)) == 963                                        # the reference's state_dict (aliases of shared tensors included)
    # views alias the flat buffer; matrices come as [N, K]
    w = a.w("blocks_u.0.mlp.fc1.weight") if "blocks_u.0.mlp.fc1.weight" in a.offset else a.w(next(n for n in a.names if n.endswith("mlp.fc1.weight")))
    assert w.dim() == 2 and w.shape == (3072, 768) and w.data_ptr() >= a.p.data_ptr()
    w.fill_(2.0)
    assert float(a.p.sum()) == 2.0 * w.numel()
    # a CPU arena gets its gradient buffer lazily (the gloo tests use it); the epoch counter opens with every zero-fill
    g = a.ensure_grads()
    assert g.numel() == a.live_end and a.zero_epoch == 0
    a.zero_grad_range(P2)
    assert a.zero_epoch == 1
    # inference arenas skip the transposed copies and the gradients
    inf = ParamArena(cfg, transposed=False, grads=False)
    assert inf.t_total == 0 and a.t_total > 0 and inf.total == a.total


@pytest.mark.parametrize("tile_rows", [64, 128])
def test_attention_tile_tables_partition_the_packed_rows(tile_rows):
    """ops.AttnTiles / ops.AttnSeqs (host tables the varlen attention kernels walk): every sequence longer than min_len is covered by ceil(L /
    tile_rows) tiles that name its first row, its length and their own first query; the shorter ones go to the fused backward's table exactly
    once; both tables count the same rows and sum of L^2 the roofline FLOPs are computed from."""
    from avsiam_amd import ops
    lens = [196] * 3 + [39, 128, 1, 129, 512, 64, 65, 2472]
    tiles = ops.AttnTiles(lens, "cpu", tile_rows=tile_rows)
    assert tiles.ntiles == sum(-(-L // tile_rows) for L in lens) and tiles.max_row == sum(lens)
    row, k = 0, 0
    for L in lens:
        for q0 in range(0, L, tile_rows):
            assert (int(tiles.start[k]), int(tiles.len[k]), int(tiles.q0[k])) == (row, L, q0)
            k += 1
        row += L
    assert tiles.rows == float(sum(lens)) and tiles.sum_sq == float(sum(L * L for L in lens))
    # the engine's split of a backward: sequences of at most 64 / 128 tokens to the fused kernel, the rest to the two-kernel path
    short, mid = ops.AttnSeqs(lens, "cpu", 0, 64), ops.AttnSeqs(lens, "cpu", 64, 128)
    long_ = ops.AttnTiles(lens, "cpu", tile_rows=tile_rows, min_len=128)
    assert short.len.tolist() == [39, 1, 64] and mid.len.tolist() == [128, 65] and short.max_len == 64 and mid.max_len == 128
    starts = [sum(lens[:i]) for i in range(len(lens))]
    assert short.start.tolist() == [starts[3], starts[5], starts[8]] and mid.start.tolist() == [starts[4], starts[9]]
    assert long_.ntiles == sum(-(-L // tile_rows) for L in lens if L > 128) and long_.max_row == sum(lens)
    assert short.rows + mid.rows + long_.rows == float(sum(lens))
    assert short.sum_sq + mid.sum_sq + long_.sum_sq == tiles.sum_sq
    assert ops.pad_rows(95630) == 95744 and ops.pad_rows(128) == 128 and ops.pad_rows(1, 64) == 64
    assert abs(ops.attn_q_scale(64) - 64 ** -0.5 * math.log2(math.e)) < 1e-12


def test_entry_point_arguments_and_process_group_plumbing(monkeypatch, capsys):
    """run_cavmae_pretrain_base.build_parser keeps the reference's flags and defaults (/root/reference/src/run_cavmae_pretrain_base.py:44-100:
    what egs/audioset/run_pretrain_base.sh passes must parse) and adds the documented extensions; utils.init_distributed_mode reads the torchrun
    environment like utils.py:250-299 and stays single-process without it; the print gate of :216-229; AverageMeter of util.py:238-253."""
    import argparse
    from avsiam_amd import utils
    from avsiam_amd.run_cavmae_pretrain_base import build_parser
    p = build_parser()
    a = p.parse_args([])
    assert (a.masking_ratio, a.mask_mode, a.contrast_loss_weight, a.mae_loss_weight, a.lr, a.target_length) == (0.75, "unstructured", 0.01, 3.0, 0.001, 1024)
    assert (a.dataset_mean, a.dataset_std, a.n_epochs if hasattr(a, "n_epochs") else 1) [:2] == (-5.081, 4.4849)
    assert a.frames == 1 and a.graph_step is False and a.raw_input is False and a.pretrain_path == 'None'
    # every option of the reference's launch line (egs/audioset/run_pretrain_base.sh:75-87)
    b = p.parse_args("--model cav-mae --dataset audioset --data-train tr.json --data-val te.json --exp-dir ./exp --label-csv l.csv --n_class 527 "
                     "--lr 5e-5 --n-epochs 25 --batch-size 4 --save_model True --mixup 0.0 --bal None --lrscheduler_start 10 --lrscheduler_decay 0.5 "
                     "--lrscheduler_step 5 --dataset_mean -5.081 --dataset_std 4.4849 --target_length 1024 --noise True --warmup True --lr_adapt False "
                     "--norm_pix_loss True --pretrain_path None --mae_loss_weight 1.0 --contrast_loss_weight 0.01 --num_workers 6 --tr_pos False --masking_ratio 0.75 "
                     "--masking_ratio_a 0.75 --mask_mode unstructured --wandb 0 --model_name run1".split())
    assert b.masking_ratio_a == 0.75 and b.num_workers == 6 and b.batch_size == 4 and b.lr == 5e-5 and b.n_epochs == 25 and b.save_model is True and b.noise is True and b.norm_pix_loss is True and b.tr_pos is False
    c = p.parse_args(["--frames", "10", "--graph-step", "--raw-input", "--steps-per-epoch", "7"])
    assert c.frames == 10 and c.graph_step and c.raw_input and c.steps_per_epoch == 7
    assert (a.fp8, a.recompute, a.share_pass_buffers, a.depth) == (None, None, None, None)            # unset: the AVSIAM_* environment / defaults decide
    d = p.parse_args(["--model", "cav-mae-huge14", "--fp8", "3", "--recompute", "0.25", "--share-pass-buffers"])
    assert (d.model, d.fp8, d.recompute, d.share_pass_buffers) == ("cav-mae-huge14", "3", "0.25", True)
    with pytest.raises(SystemExit):
        p.parse_args(["--fp8", "4"])
    # no launcher: single process, rank 0, no process group
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK"):
        monkeypatch.delenv(k, raising=False)
    ns = argparse.Namespace()
    utils.init_distributed_mode(ns)
    assert (ns.rank, ns.world_size, ns.gpu, ns.distributed) == (0, 1, 0, False) and "Not using distributed mode" in capsys.readouterr().out
    # the print gate: silent on non-master ranks unless forced, restored afterwards
    utils.setup_for_distributed(False)
    try:
        print("hidden"); print("shown", force=True)
    finally:
        utils.restore_print()
    out = capsys.readouterr().out
    assert "hidden" not in out and "shown" in out
    m = utils.AverageMeter()
    m.update(2.0, 3); m.update(4.0, 1)
    assert (m.val, m.sum, m.count, m.avg) == (4.0, 10.0, 4, 2.5)
    utils.init_seeds(87)
    x = torch.rand(1).item(); utils.init_seeds(87)
    assert torch.rand(1).item() == x


def test_model_family_exports_and_per_model_options(monkeypatch):
    """The reference's export list (/root/reference/src/models/__init__.py:8-13): CAVMAE, CAVMAE_BASE, CAVMAE_LARGE, CAVMAE_HUGE and their CAVMAEFT_*
    classes exist with the reference constructor signature; the larger skeletons are parameterisations of the same class.  Precision / memory policy
    are options of ONE model (config.EngineOptions): the AVSIAM_* environment seeds defaults at construction, keywords win, two models differ."""
    import inspect
    from avsiam_amd import models
    from avsiam_amd.config import EngineOptions
    for name in ("CAVMAE", "CAVMAE_BASE", "CAVMAE_LARGE", "CAVMAE_HUGE", "CAVMAEFT", "CAVMAEFT_BASE", "CAVMAEFT_LARGE", "CAVMAEFT_HUGE"):
        assert hasattr(models, name), name
    assert issubclass(models.CAVMAE_LARGE, models.CAVMAE_BASE) and issubclass(models.CAVMAE_HUGE, models.CAVMAE_BASE) and models.CAVMAE is models.CAVMAE_BASE
    assert issubclass(models.CAVMAEFT_LARGE, models.CAVMAEFT_BASE) and issubclass(models.CAVMAEFT_HUGE, models.CAVMAEFT_BASE)
    ref_ctor = ["img_size", "audio_length", "patch_size", "in_chans", "embed_dim", "modality_specific_depth", "num_heads", "decoder_embed_dim",
                "decoder_depth", "decoder_num_heads", "mlp_ratio", "norm_layer", "norm_pix_loss", "tr_pos", "opt"]          # cav_mae_base.py:219-222
    sig = inspect.signature(models.CAVMAE_BASE.__init__)
    assert [p for p in sig.parameters if sig.parameters[p].kind == inspect.Parameter.POSITIONAL_OR_KEYWORD][1:] == ref_ctor
    for k in ("cfg", "fp8_mode", "recompute", "grad_stream", "share_pass_buffers", "options"):
        assert sig.parameters[k].kind == inspect.Parameter.KEYWORD_ONLY, k
    for k in ("AVSIAM_FP8", "AVSIAM_RECOMPUTE", "AVSIAM_GRAD_STREAM", "AVSIAM_DETERMINISTIC"):
        monkeypatch.delenv(k, raising=False)
    d = EngineOptions.from_env()
    assert (d.fp8, d.recompute, d.grad_stream, d.wgrad_stream, d.deterministic, d.prune_dead) == ("0", "0", "bf16", "auto", False, True)
    monkeypatch.setenv("AVSIAM_FP8", "2")
    monkeypatch.setenv("AVSIAM_RECOMPUTE", "")                                                   # empty = unset
    assert EngineOptions.from_env().fp8 == "2" and EngineOptions.from_env(fp8="3").fp8 == "3" and EngineOptions.from_env(fp8=None).recompute == "0"
    with pytest.raises(ValueError):
        EngineOptions.from_env(fp8="7")
    with pytest.raises(ValueError):
        EngineOptions.from_env(recompute="1.5")
    monkeypatch.delenv("AVSIAM_FP8")
    # two small models (depth 1; CPU construction only - forward needs the GPU): each owns its options object
    small = {"depth": 1, "dec_depth": 1}
    m0 = models.CAVMAE_BASE(cfg=AVSiamConfig(audio_tokens=128, **small), verbose=False)
    m3 = models.CAVMAE_HUGE(cfg=vit_huge14(frames=1, **small), verbose=False, fp8_mode="3", recompute="0.5")
    assert (m0.options.fp8, m3.options.fp8, m3.options.recompute) == ("0", "3", "0.5") and m0.options is not m3.options
    assert (m3.cfg.embed_dim, m3.cfg.head_dim, m3.cfg.st) == (1280, 80, 14)
    m0.set_options(deterministic=True)                                                           # a runtime field: no engine is dropped, m3 untouched
    assert m0.options.deterministic and not m3.options.deterministic
    shared = EngineOptions(fp8="1")                                                              # one options object handed to two models: each keeps a copy
    ma, mb = (models.CAVMAE_BASE(cfg=AVSiamConfig(audio_tokens=128, **small), verbose=False, options=shared) for _ in range(2))
    ma.set_options(fp8="2", deterministic=True)
    assert (ma.options.fp8, mb.options.fp8, shared.fp8) == ("2", "1", "1") and not mb.options.deterministic
    with pytest.raises(ValueError):
        models.CAVMAE_LARGE(cfg=AVSiamConfig(), verbose=False)
    with pytest.raises(ValueError):
        models.CAVMAE_BASE(cfg=AVSiamConfig(**small), verbose=False, fp8_mode="1", options=EngineOptions())
    assert models.CAVMAE_LARGE(cfg=vit_large(audio_tokens=128, **small), verbose=False).cfg.embed_dim == 1024
