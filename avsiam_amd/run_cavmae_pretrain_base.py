"""Entry point with the flag surface of the reference (/root/reference/src/run_cavmae_pretrain_base.py:47-104).

    torchrun --nnodes=1 --nproc_per_node=8 --master-addr 127.0.0.1 -m avsiam_amd.run_cavmae_pretrain_base \\
        --model cav-mae --dataset audioset --target_length 1024 --batch-size 64 --lr 2e-4 --n-epochs 1 ...

Flags the reference parses but its model ignores (masking ratios, mask_mode, norm_pix_loss, tr_pos: SURVEY.md section 5)
are accepted and equally inert.  Data flags: with --data-train '' (or 'synthetic') AudioSet-shaped synthetic
batches are used - the sqlite/wav/mp4 input pipeline (src/dataloader.py) is out of scope for this path; --data-val
'synthetic' adds a synthetic validation loader so validate() and the best-model bookkeeping run.
Extensions: --frames (T frames per sample), --steps-per-epoch / --val-steps (synthetic epoch lengths), --raw-input (the
loader hands over what the reference's dataset holds BEFORE normalisation - un-normalised fbank, uint8 frames - and the
dataloader's arithmetic, incl. --noise, is applied by the kernels that read the inputs: the patch gather of the embedding
and the target gather of the reconstruction loss).
"""
import argparse
import ast
import json
import os
import pickle
import time


# --model -> (class name in avsiam_amd.models, shape function in avsiam_amd.config, patch stride)
MODELS = {"cav-mae": ("CAVMAE_BASE", "vit_base", 16), "cav-mae-large": ("CAVMAE_LARGE", "vit_large", 16), "cav-mae-huge14": ("CAVMAE_HUGE", "vit_huge14", 14)}


def build_parser():
    parser = argparse.ArgumentParser(formatter_class=argparse.ArgumentDefaultsHelpFormatter)
    parser.add_argument("--data-train", type=str, default='', help="training data json ('' or 'synthetic': synthetic tensors)")
    parser.add_argument("--data-val", type=str, default='', help="validation data json")
    parser.add_argument("--data-eval", type=str, default=None, help="evaluation data json")
    parser.add_argument("--label-csv", type=str, default='', help="csv with class labels")
    parser.add_argument("--n_class", type=int, default=527, help="number of classes")
    parser.add_argument("--model", type=str, default='cav-mae', choices=list(MODELS),
                        help="the model used: cav-mae = models.CAVMAE_BASE (the only one the reference's entry point builds, :171-175); cav-mae-large / "
                             "cav-mae-huge14 = models.CAVMAE_LARGE / CAVMAE_HUGE, the skeletons the reference exports by name (src/models/__init__.py:9,13)")
    parser.add_argument("--dataset", type=str, default="audioset", choices=["audioset", "esc50", "speechcommands", "fsd50k", "vggsound", "epic", "k400", "msrvtt"])
    parser.add_argument("--dataset_mean", type=float, default=-5.081)
    parser.add_argument("--dataset_std", type=float, default=4.4849)
    parser.add_argument("--target_length", type=int, default=1024, help="the input length in frames")
    parser.add_argument("--noise", type=ast.literal_eval, default=False)
    parser.add_argument("--exp-dir", type=str, default="", help="directory to dump experiments")
    parser.add_argument('--lr', '--learning-rate', default=0.001, type=float, metavar='LR')
    parser.add_argument("--optim", type=str, default="adam", choices=["sgd", "adam"])
    parser.add_argument('-b', '--batch-size', default=12, type=int, metavar='N')
    parser.add_argument('-w', '--num_workers', default=6, type=int, metavar='NW')
    parser.add_argument("--n-epochs", type=int, default=1)
    parser.add_argument("--lr_patience", type=int, default=2)
    parser.add_argument("--lr_adapt", type=ast.literal_eval, default=False)
    parser.add_argument("--metrics", type=str, default="mAP", choices=["mAP", "acc"])
    parser.add_argument('--warmup', type=ast.literal_eval, default='True')
    parser.add_argument("--lrscheduler_start", default=10, type=int)
    parser.add_argument("--lrscheduler_step", default=5, type=int)
    parser.add_argument("--lrscheduler_decay", default=0.5, type=float)
    parser.add_argument("--n-print-steps", type=int, default=50)
    parser.add_argument('--save_model', type=ast.literal_eval, default=False)
    parser.add_argument("--mixup", type=float, default=0)
    parser.add_argument("--bal", type=str, default=None)
    parser.add_argument("--cont_model", type=str, default=None)
    parser.add_argument("--weight_file", type=str, default=None)
    parser.add_argument('--norm_pix_loss', type=ast.literal_eval, default=None)
    parser.add_argument("--pretrain_path", type=str, default='None')
    parser.add_argument("--contrast_loss_weight", type=float, default=0.01)
    parser.add_argument("--mae_loss_weight", type=float, default=3.0)
    parser.add_argument('--tr_pos', type=ast.literal_eval, default=None)
    parser.add_argument("--masking_ratio", type=float, default=0.75)
    parser.add_argument("--masking_ratio_a", type=float, default=None)
    parser.add_argument("--mask_mode", type=str, default='unstructured', choices=['unstructured', 'time', 'freq', 'tf'])
    parser.add_argument("--wandb", type=int, default=0)
    parser.add_argument('--model_name', type=str, default=None)
    parser.add_argument('--world_size', default=1, type=int)
    parser.add_argument('--local_rank', default=-1, type=int)
    parser.add_argument('--dist_url', default='env://')
    # extensions
    parser.add_argument('--fp8', default=None, choices=["0", "1", "2", "3"],
                        help="fp8 MFMA path of THIS run's model (BASELINE configs[4]): 1 = e4m3 forward GEMMs, 2 = + e5m2 x e4m3 input gradients, "
                             "3 = + fp8 weight gradients; default: AVSIAM_FP8 or 0 (bf16)")
    parser.add_argument('--recompute', default=None, metavar="FRACTION", help="per-layer activation recompute: 0 | 1 | the share of every stack's leading blocks")
    parser.add_argument('--share-pass-buffers', dest="share_pass_buffers", action="store_true", default=None,
                        help="both passes of the step take their activation buffers from one pool (the card holds the larger pass, not the sum)")
    parser.add_argument('--deterministic', action="store_true", default=None,
                        help="bit-reproducible steps (one writer per gradient element, one stream; slower) - for debugging, e.g. the first multi-GPU session")
    parser.add_argument('--depth', default=None, type=int, help="(tests) encoder depth override of the chosen skeleton")
    parser.add_argument('--frames', default=1, type=int, help="frames per sample (extension; reference pre-training uses 1)")
    parser.add_argument('--steps-per-epoch', dest="steps_per_epoch", default=20, type=int, help="synthetic-data epoch length")
    parser.add_argument('--val-steps', dest="val_steps", default=2, type=int, help="synthetic validation batches per epoch")
    parser.add_argument('--graph-step', dest="graph_step", action="store_true",
                        help="(extension) replay the training step from one captured hipGraph - single GPU; for small per-GPU batches such as the "
                             "reference's 4, where the eager step is bound by the host's launch rate")
    parser.add_argument('--raw-input', dest="raw_input", action="store_true",
                        help="feed un-normalised fbank + uint8 frames and normalise on the device (dataloader.py:505-513,461-462)")
    return parser


def make_input_xf(args, device):
    """The reference dataset's per-sample arithmetic as per-batch transform descriptors (ops.InputXf) that the input-reading
    kernels apply on the fly: (fbank - dataset_mean) / dataset_std, + noise and time roll in training when --noise
    (dataloader.py:505-513: amp = rand() / 10 and shift = randint(-T, T) per sample); uint8 frames -> (x / 255 - mean_c) / std_c
    (:461-462, 152-155).  Returns f(batch, train) -> (audio transform, frame transform)."""
    import numpy as np
    import torch
    from .ops import InputXf
    rank = getattr(args, "rank", getattr(args, "local_rank", 0)) or 0
    rng = np.random.default_rng(87 + rank)
    count = [0]
    frames = InputXf.frames()

    def make(batch, train=True):
        count[0] += 1
        if bool(args.noise) and train:
            amp = torch.from_numpy((rng.random(batch) / 10).astype(np.float32)).to(device)
            shift = torch.from_numpy(rng.integers(-args.target_length, args.target_length, batch).astype(np.int32)).to(device)
            return InputXf.audio(args.dataset_mean, args.dataset_std, shift, amp, seed=((87 + rank) << 32) | count[0]), frames    # the rank in the Philox key: every
            # data-parallel rank draws its own noise field (the reference: independent torch.rand per sample and worker)
        return InputXf.audio(args.dataset_mean, args.dataset_std), frames
    return make


def main(argv=None):
    import torch
    from . import models, utils
    from .config import AVSiamConfig
    from .traintest_cavmae_base import SyntheticAVLoader, train
    print("I am process %s, running on %s: starting (%s)" % (os.getpid(), os.uname()[1], time.asctime()))
    args = build_parser().parse_args(argv)
    if args.masking_ratio_a is None:
        args.masking_ratio_a = args.masking_ratio
    args.local_rank = int(os.environ.get("LOCAL_RANK", 0))
    utils.init_seeds(87 + args.local_rank)                                    # :113
    utils.init_distributed_mode(args)                                         # :114
    try:
        return _run(args, torch, models, AVSiamConfig, SyntheticAVLoader, train)
    finally:
        utils.restore_print()
        if args.distributed and torch.distributed.is_initialized():
            torch.distributed.destroy_process_group()


def _run(args, torch, models, AVSiamConfig, SyntheticAVLoader, train):
    print('current mae loss {:.3f}, and contrastive loss {:.3f}'.format(args.mae_loss_weight, args.contrast_loss_weight))
    if args.data_train not in ('', 'synthetic'):
        raise SystemExit("only synthetic AudioSet-shaped data is supported on this path (see module docstring)")
    if args.model not in MODELS:
        raise ValueError('model not supported')                                # :177
    from . import config as _config
    cls_name, shape_fn, stride = MODELS[args.model]
    freq = 128 // stride                                                       # frequency patches of the 128-bin fbank (8; 9 on the 14 x 14 grid)
    kw = {"audio_tokens": args.target_length // stride * freq, "frames": args.frames}
    if args.depth is not None:
        kw["depth"] = args.depth
    cfg = getattr(_config, shape_fn)(**kw)
    if args.model == 'cav-mae':
        print('pretrain a cav-mae model with 11 modality-specific layers and 1 modality-sharing layers')        # :172
    audio_model = getattr(models, cls_name)(audio_length=args.target_length, norm_pix_loss=args.norm_pix_loss,
                                            modality_specific_depth=23, tr_pos=args.tr_pos, opt=args, cfg=cfg,
                                            fp8_mode=args.fp8, recompute=args.recompute, share_pass_buffers=args.share_pass_buffers,
                                            deterministic=args.deterministic)        # :175
    if args.pretrain_path not in ('None', '', None):
        # resume from a checkpoint this loop (or the reference's, traintest_cavmae_base.py:223-234: 'module.'-prefixed keys) wrote - the
        # reference carries the same load commented out (:181-198).  With the fp8 mode the delayed-scaling state saved beside the weights
        # ('<path>.fp8') is restored too, so the resumed run quantises on the grids it stopped with instead of re-calibrating.
        sd = torch.load(args.pretrain_path, map_location='cpu')
        sd = {(k[len('module.'):] if k.startswith('module.') else k): v for k, v in sd.items()}
        miss, unexpected = audio_model.load_state_dict(sd, strict=False)
        print('now load pretrain model from {:s}, missing keys: {:d}, unexpected keys: {:d}'.format(args.pretrain_path, len(miss), len(unexpected)))
        # strict=False is what the reference's consumers use (run_cavmae_ft_base.py:245-248), but a checkpoint of ANOTHER model would then
        # train silently from the random start: refuse one that shares no key with this model, and say loudly what is left out
        total = len(audio_model.state_dict())
        if len(miss) >= total:
            raise SystemExit('--pretrain_path {:s}: none of the {:d} keys of CAVMAE_BASE is in this checkpoint - not a checkpoint of this model'
                             .format(args.pretrain_path, total))
        if miss or unexpected:
            print('WARNING: --pretrain_path is a PARTIAL match: {:d} of {:d} keys keep their initial value (first: {}), {:d} checkpoint keys '
                  'are ignored (first: {})'.format(len(miss), total, list(miss)[:3], len(unexpected), list(unexpected)[:3]), flush=True)
        # The optimizer: the loop saves the FIRST optimizer's state beside the weights (best_optim_state.pth, as traintest_cavmae_base.py:230);
        # when that file sits beside the checkpoint, Adam #1 continues from it (moments and step count).  Adam #2's state is not saved by the
        # reference's loop either, so the MAE pass's moments restart - without the file this is a weights-only warm start, and says so.
        from .param_spec import P1, P2
        opt_path = os.path.join(os.path.dirname(args.pretrain_path), 'best_optim_state.pth')
        if os.path.exists(opt_path):
            audio_model.to(torch.device("cuda", getattr(args, "gpu", 0)))      # (the optimizer state lives beside the weights; train() keeps this device)
            audio_model.load_optimizer_state_dict(P1, torch.load(opt_path, map_location='cpu'))
            opt2 = os.path.join(os.path.dirname(args.pretrain_path), 'best_optim_state_2.pth')     # written by this loop, not by the reference's
            if os.path.exists(opt2):
                audio_model.load_optimizer_state_dict(P2, torch.load(opt2, map_location='cpu'))
            print('restored the state of optimizer 1 (Adam moments, step {:d}) from {:s}; optimizer 2 {}'
                  .format(audio_model._opt_state[P1]['step'], opt_path,
                          'too (step {:d})'.format(audio_model._opt_state[P2]['step']) if os.path.exists(opt2) else 'restarts'))
        else:
            print('no best_optim_state.pth beside the checkpoint: weights-only warm start (both Adam states restart, bias correction from step 1)')
        if audio_model.options.fp8 != "0" and os.path.exists(args.pretrain_path + ".fp8"):
            audio_model.load_fp8_state(torch.load(args.pretrain_path + ".fp8", map_location='cpu'))
            print('restored the fp8 delayed-scaling state from {:s}.fp8'.format(args.pretrain_path))
    if args.exp_dir and args.rank == 0:
        os.makedirs("%s/models" % args.exp_dir, exist_ok=True)
        with open("%s/args.pkl" % args.exp_dir, "wb") as f:
            pickle.dump(args, f)
        with open(args.exp_dir + '/args.json', 'w') as f:
            json.dump(args.__dict__, f, indent=2)
    val_loader = None
    if args.data_val == 'synthetic':
        val_loader = SyntheticAVLoader(cfg, args.batch_size, args.val_steps, torch.device("cuda", args.gpu), 1087 + args.rank,
                                       raw=args.raw_input)
    if args.raw_input:
        args._input_xf = make_input_xf(args, torch.device("cuda", args.gpu))
    print('Now starting training for {:d} epochs.'.format(args.n_epochs))
    train(audio_model, None, [val_loader, None], [None, None], None, args, None)                                  # :212
    args.__dict__.pop("_input_xf", None)
    return audio_model


if __name__ == "__main__":
    main()
