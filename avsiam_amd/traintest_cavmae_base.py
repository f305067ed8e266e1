"""The pre-training step and loop of the reference (/root/reference/src/traintest_cavmae_base.py:29-264),
re-scheduled for the flat-arena model.

Reference step (:131-152):  pass 1 contrastive-only fwd/bwd -> Adam#1 -> pass 2 MAE-only fwd/bwd (sees the updated
weights) -> Adam#2, under fp16 autocast + GradScaler and DDP(find_unused_parameters=True).
Here: bf16 operands with fp32 accumulation (no loss scaling needed), gradients of each pass live in one contiguous
range -> one RCCL all-reduce + one fused Adam launch per pass; parameters a pass does not touch get no update,
exactly like torch.optim.Adam skipping grad=None parameters.
"""
import datetime
import os
import pickle
import time

import numpy as np
import torch

from .param_spec import P1, P2
from .utils import AverageMeter


def train_step(model, a_input, v_input, lr, plans=None, input_xf=None):
    """One reference step on one batch.  Returns the device scalars the reference logs (no host sync here):
    (loss_pass2, loss_mae_a, loss_mae_v, loss_c, c_acc).  input_xf: the batch's raw-input transforms (both passes see the SAME
    augmented batch, as with the reference's loader)."""
    pm, pc = plans if plans is not None else (None, None)     # None: masks are drawn on the device
    out = model(a_input, v_input, mae_loss_weight=0, contrast_loss_weight=1, mask_plan=pc, input_xf=input_xf)        # :132
    loss_c, c_acc = out[4], out[7]
    out[0].backward()                                                                             # :138 (+ DDP's all-reduce)
    model.allreduce_grads(P1, average=False)                  # no-op when backward reduced; the 1/W rides in the Adam kernel
    model.adam_step(P1, lr)                                                                       # :139
    out = model(a_input, v_input, mae_loss_weight=1, contrast_loss_weight=0, mask_plan=pm, input_xf=input_xf)        # :147
    out[0].backward()                                                                             # :150
    model.allreduce_grads(P2, average=False)
    model.adam_step(P2, lr)                                                                       # :151
    return out[0], out[2], out[3], loss_c, c_acc


class SyntheticAVLoader:
    """AudioSet-shaped synthetic batches (there is no dataset here): a ~ N(0,1) [B, target_length, 128],
    v ~ N(0,1) [B,(T,)3,224,224], generated on the device once and re-used.
    raw=True yields what the reference's dataset holds BEFORE its normalisation (/root/reference/src/dataloader.py:505-513,
    152-155,461-462): un-normalised fbank (mean/std of AudioSet) and uint8 frames; the consumer passes them through
    avsiam_amd.preprocess."""

    def __init__(self, cfg, batch_size, steps, device, seed=87, raw=False):
        from .weights import synth_inputs
        a, v = synth_inputs(cfg, batch_size, seed)
        if raw:
            a = a * 4.4849 - 5.081                            # (x - mean) / std of the loader (--dataset_mean/--dataset_std) gives ~N(0,1) back
            v = (v * 0.25 + 0.5).clamp(0, 1).mul(255).round().to(torch.uint8)
        self.a, self.v, self.steps, self.raw = a.to(device), v.to(device), steps, raw

    def __len__(self):
        return self.steps

    def __iter__(self):
        for _ in range(self.steps):
            yield self.a, self.v, None


def validate(audio_model, val_loader, val_sampler, args):
    """validate() of the reference (:381-424): no-grad forward with BOTH losses at the run's loss weights
    (``args.mae_loss_weight`` / ``args.contrast_loss_weight``, :401); per-batch losses are averaged over batches (:417-422).
    Returns (loss, loss_mae, loss_mae_a, loss_mae_v, loss_c, c_acc).  Masks stay random, as in the reference."""
    device = audio_model.arena.p.device
    rank = getattr(args, "rank", 0)
    raw = getattr(args, "_input_xf", None)
    acc = [[] for _ in range(6)]
    with torch.no_grad():
        for i, (a_input, v_input, _) in enumerate(val_loader):
            a_input, v_input = a_input.to(device), v_input.to(device)
            if rank == 0 and i % 50 == 0:
                print("Val index: {}/{}".format(i, len(val_loader)))
            out = audio_model(a_input, v_input, args.masking_ratio, args.masking_ratio, mae_loss_weight=args.mae_loss_weight,
                              contrast_loss_weight=args.contrast_loss_weight, mask_mode=args.mask_mode,
                              input_xf=raw(a_input.size(0), train=False) if raw is not None else None)
            for lst, j in zip(acc, (0, 1, 2, 3, 4, 7)):
                lst.append(out[j].detach())
    if not acc[0]:
        return (float("nan"),) * 6
    host = torch.stack([torch.stack(l) for l in acc]).float().cpu().numpy()           # ONE device->host copy
    return tuple(float(np.mean(row)) for row in host)


def _save_checkpoint(audio_model, path):
    """DDP-style 'module.'-prefixed keys, so the reference's consumers (run_cavmae_ft_base.py:245-248, linear_val :269-278)
    can load it."""
    sd = {"module." + k: v.detach().cpu() for k, v in audio_model.state_dict().items()}
    torch.save(sd, path)
    f8 = audio_model.fp8_state() if hasattr(audio_model, "fp8_state") else {}
    if f8:                                                  # fp8 mode: the delayed-scaling state, beside the weights (model.load_fp8_state)
        torch.save(f8, path + ".fp8")


def train(audio_model, train_sampler, test_loader, test_sampler, train_loader_linear, args, audio_conf):
    """Signature of the reference ``train`` (:29).  ``train_sampler`` may be a DataLoader-like iterable yielding
    (a_input, v_input, label); with args.data_train in ('', 'synthetic') a SyntheticAVLoader is used.  The per-epoch
    MLP probe (linear_val, broken in the reference: SURVEY.md quick facts) is out of scope."""
    rank = getattr(args, "rank", 0)
    device = torch.device("cuda", getattr(args, "gpu", 0))
    audio_model = audio_model.to(device)
    audio_model.set_distributed(getattr(args, "world_size", 1), rank, getattr(args, "_comm", None))
    audio_model.publish_grads = False                       # gradients stay in the flat arena
    print('Total parameter number is : {:.3f} million'.format(sum(p.numel() for p in audio_model.parameters()) / 1e6))
    per_sample_time, per_sample_dnn_time = AverageMeter(), AverageMeter()
    loss_av_meter, loss_a_meter, loss_v_meter, loss_c_meter = AverageMeter(), AverageMeter(), AverageMeter(), AverageMeter()
    meters = (loss_av_meter, loss_a_meter, loss_v_meter, loss_c_meter)
    progress, result = [], np.zeros([args.n_epochs, 10])
    exp_dir = args.exp_dir
    global_step, epoch, best_loss, best_epoch = 0, 1, np.inf, 0
    start_time = time.time()
    lr = args.lr
    milestones = set(range(args.lrscheduler_start, 1000, args.lrscheduler_step))      # MultiStepLR (:73-74)
    if train_sampler is None or not hasattr(train_sampler, "__iter__"):
        train_loader = SyntheticAVLoader(audio_model.cfg, args.batch_size, getattr(args, "steps_per_epoch", 20), device, 87 + rank,
                                         raw=getattr(args, "raw_input", False))
    else:
        train_loader = train_sampler
    raw = getattr(args, "_input_xf", None)                  # raw inputs (--raw-input): per-batch transforms, applied inside the kernels
    test_loader = test_loader[0] if isinstance(test_loader, (list, tuple)) else test_loader
    val_sampler = test_sampler[0] if isinstance(test_sampler, (list, tuple)) else test_sampler
    # The reference updates its loss meters every step with four .item() host syncs (:160-163).  Here the four losses are
    # accumulated ON THE DEVICE every step (sum and count), and read back on print steps and at the end of the epoch - same
    # epoch means in result.csv, same NaN guard, no per-step sync.
    dsum = torch.zeros(4, device=device)
    dcount, dsynced = 0, 0

    def sync_meters(last=None):
        """Fold the device-side sums into the meters; `last` = this step's four values for the meters' .val."""
        nonlocal dcount, dsynced
        host = dsum.cpu().numpy() if last is None else torch.cat([dsum, last]).cpu().numpy()
        n = dcount - dsynced
        for k, m in enumerate(meters):
            m.sum, m.count = float(host[k]) * B_ref[0], dcount * B_ref[0]
            m.avg = m.sum / max(m.count, 1)
            if last is not None:
                m.val = float(host[4 + k])
        dsynced = dcount
        return n

    B_ref = [args.batch_size]
    # --graph-step (extension, single GPU): the step replayed from one captured hipGraph (avsiam_amd.graph_step) - for the reference's own
    # per-GPU batch of 4, where the eager step is bound by the host's launch rate.  Captured on the first batch (whose one warm-up step IS that
    # batch's training step) and again when the batch size or the learning rate changes; every other batch is copied into the captured buffers.
    use_graph = bool(getattr(args, "graph_step", False)) and getattr(args, "world_size", 1) == 1 and raw is None and not audio_model.share_pass_buffers
    if getattr(args, "graph_step", False) and not use_graph:
        print("--graph-step ignored: it needs one GPU, pre-normalised inputs and private pass buffers")
    gstep, gkey, gbuf = None, None, None
    while epoch < args.n_epochs + 1:
        begin_time = end_time = time.time()
        print('---------------'); print(datetime.datetime.now())
        print("current #epochs=%s, #steps=%s" % (epoch, global_step))
        for i, (a_input, v_input, _) in enumerate(train_loader):
            B = a_input.size(0)
            B_ref[0] = B
            a_input = a_input.to(device, non_blocking=True)
            v_input = v_input.to(device, non_blocking=True)
            dnn_start_time = time.time()
            if use_graph:
                if gstep is None or gkey != (tuple(a_input.shape), tuple(v_input.shape), lr):
                    from .graph_step import GraphedTrainStep
                    gbuf = (a_input.clone(), v_input.clone())
                    gstep, gkey = GraphedTrainStep(audio_model, gbuf[0], gbuf[1], lr, warmup=1), (tuple(a_input.shape), tuple(v_input.shape), lr)
                    loss, la, lv, lc, c_acc = gstep.warm_out
                else:
                    gbuf[0].copy_(a_input); gbuf[1].copy_(v_input)
                    loss, la, lv, lc, c_acc = gstep.step()
            else:
                loss, la, lv, lc, c_acc = train_step(audio_model, a_input, v_input, lr,
                                                     input_xf=raw(B, train=True) if raw is not None else None)
            step_vals = torch.stack([loss.detach(), la.detach(), lv.detach(), lc.detach()]).float()
            dsum += step_vals
            dcount += 1
            print_step = global_step % args.n_print_steps == 0
            if print_step or global_step == 0:                # host sync only here (the reference syncs 4x per step)
                sync_meters(step_vals)
                per_sample_time.update((time.time() - end_time) / B)
                per_sample_dnn_time.update((time.time() - dnn_start_time) / B)
                print('Epoch: [{0}][{1}/{2}]\t Per Sample Total Time {3:.5f}\t Per Sample DNN Time {4:.5f}\t Train Total Loss {5:.4f}\t'
                      'Train MAE Loss Audio {6:.4f}\t Train MAE Loss Visual {7:.4f}\t Train Contrastive Loss {8:.4f}\t Train Contrastive Acc {9:.3f}'
                      .format(epoch, i, len(train_loader), per_sample_time.avg, per_sample_dnn_time.avg, loss_av_meter.val,
                              loss_a_meter.val, loss_v_meter.val, loss_c_meter.val, c_acc.item()), flush=True)
                if np.isnan(loss_av_meter.avg):
                    print("training diverged...")
                    return
            end_time = time.time()
            global_step += 1
        sync_meters()                                        # epoch means over EVERY step, print step or not
        if np.isnan(loss_av_meter.avg):
            print("training diverged...")
            return
        ev = (0.0,) * 6
        if test_loader is not None:
            print('start validation')
            ev = validate(audio_model, test_loader, val_sampler, args)
            print("Eval Audio MAE Loss: {:.6f}".format(ev[2])); print("Eval Visual MAE Loss: {:.6f}".format(ev[3]))
            print("Eval Total MAE Loss: {:.6f}".format(ev[1])); print("Eval Contrastive Loss: {:.6f}".format(ev[4]))
            print("Eval Total Loss: {:.6f}".format(ev[0])); print("Eval Contrastive Accuracy: {:.6f}".format(ev[5]))
        print("Train Audio MAE Loss: {:.6f}".format(loss_a_meter.avg)); print("Train Visual MAE Loss: {:.6f}".format(loss_v_meter.avg))
        print("Train Contrastive Loss: {:.6f}".format(loss_c_meter.avg)); print("Train Total Loss: {:.6f}".format(loss_av_meter.avg))
        result[epoch - 1, :] = [loss_a_meter.avg, loss_v_meter.avg, loss_c_meter.avg, loss_av_meter.avg, ev[2], ev[3], ev[4], ev[0], ev[5], lr]
        if exp_dir:
            os.makedirs("%s/models" % exp_dir, exist_ok=True)
            if rank == 0:
                np.savetxt(exp_dir + '/result.csv', result, delimiter=',')
        # best model by evaluation loss (:221-230); without a validation loader there is nothing to rank epochs by
        if test_loader is not None and ev[0] < best_loss:
            best_loss, best_epoch = ev[0], epoch
        if rank == 0 and exp_dir:
            if best_epoch == epoch:
                _save_checkpoint(audio_model, "%s/models/best_audio_model.pth" % exp_dir)
                torch.save(audio_model.optimizer_state_dict(P1, lr), "%s/models/best_optim_state.pth" % exp_dir)    # the first optimizer, as :230
                # (extension: the reference saves only optimizer 1; with the second one beside it --pretrain_path is a full resume)
                torch.save(audio_model.optimizer_state_dict(P2, lr), "%s/models/best_optim_state_2.pth" % exp_dir)
            if getattr(args, "save_model", False):
                _save_checkpoint(audio_model, "%s/models/audio_model.%d.pth" % (exp_dir, epoch))
        if epoch in milestones:
            lr *= args.lrscheduler_decay
        print('Epoch-{0} lr: {1}'.format(epoch, lr))
        if rank == 0 and exp_dir:
            progress.append([epoch, global_step, best_epoch, best_loss, time.time() - start_time])
            with open("%s/progress.pkl" % exp_dir, "wb") as f:
                pickle.dump(progress, f)
        print('epoch {:d} training time: {:.3f}'.format(epoch, time.time() - begin_time))
        epoch += 1
        for m in (per_sample_time, per_sample_dnn_time) + meters:
            m.reset()
        dsum.zero_()
        dcount = dsynced = 0
