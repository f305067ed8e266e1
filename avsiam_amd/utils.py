"""Process-group setup, mirroring ``utils.init_distributed_mode`` of the reference
(/root/reference/src/utils.py:250-299): RANK / WORLD_SIZE / LOCAL_RANK from the torchrun environment,
``torch.cuda.set_device``, ``init_process_group`` + barrier.  backend "nccl" is RCCL on ROCm (xGMI inside a node);
"gloo" is used by the CPU tests."""
import os
import random

import numpy as np
import torch
import torch.distributed as dist


def init_seeds(seed=0):
    """run_cavmae_pretrain_base.py:31-41 (python / numpy / torch RNGs); called with 87 + local_rank (:113)."""
    random.seed(seed)
    np.random.seed(seed)
    torch.manual_seed(seed)


def init_distributed_mode(args, backend=None):
    """utils.py:250-299 of the reference: rank / world size / GPU from the torchrun environment, one process per GPU,
    ``init_process_group`` + barrier.  Like the reference, a launch with RANK and WORLD_SIZE in the environment ALWAYS forms
    the process group (also at world size 1: ``args.distributed = True``, :288); a plain ``python -m ...`` start without
    them runs single-process (the reference's commented-out 'Not using distributed mode' branch, :276-279)."""
    launched = "RANK" in os.environ and "WORLD_SIZE" in os.environ
    args.rank = int(os.environ.get("RANK", 0))
    args.world_size = int(os.environ.get("WORLD_SIZE", 1))
    args.gpu = int(os.environ.get("LOCAL_RANK", 0))
    args.distributed = launched
    if torch.cuda.is_available():
        torch.cuda.set_device(args.gpu)
    if not launched:
        print('Not using distributed mode')
        return
    args.dist_backend = backend or ("nccl" if torch.cuda.is_available() else "gloo")
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    os.environ.setdefault("MASTER_PORT", "29500")
    print('| distributed init (rank {}): {}, gpu {}'.format(args.rank, getattr(args, "dist_url", "env://"), args.gpu), flush=True)
    kw = {}
    if args.dist_backend == "nccl":
        kw["device_id"] = torch.device("cuda", args.gpu)
    dist.init_process_group(backend=args.dist_backend, init_method=getattr(args, "dist_url", "env://") or "env://",
                            world_size=args.world_size, rank=args.rank, **kw)
    dist.barrier()
    setup_for_distributed(args.rank == 0)


_builtin_print = None


def setup_for_distributed(is_master):
    """utils.py:216-229 of the reference: printing is disabled on every rank but the master (``force=True`` overrides)."""
    import builtins
    global _builtin_print
    if _builtin_print is None:
        _builtin_print = builtins.print

    def print(*a, **k):
        force = k.pop("force", False)
        if is_master or force:
            _builtin_print(*a, **k)

    builtins.print = print


def restore_print():
    import builtins
    if _builtin_print is not None:
        builtins.print = _builtin_print


class AverageMeter:
    """src/utilities/util.py:238-253"""

    def __init__(self):
        self.reset()

    def reset(self):
        self.val = self.avg = self.sum = self.count = 0

    def update(self, val, n=1):
        self.val = val
        self.sum += val * n
        self.count += n
        self.avg = self.sum / self.count
