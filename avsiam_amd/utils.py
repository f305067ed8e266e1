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
    args.rank = int(os.environ.get("RANK", 0))
    args.world_size = int(os.environ.get("WORLD_SIZE", 1))
    args.gpu = int(os.environ.get("LOCAL_RANK", 0))
    args.distributed = args.world_size > 1
    if torch.cuda.is_available():
        torch.cuda.set_device(args.gpu)
    if not args.distributed:
        return
    args.dist_backend = backend or ("nccl" if torch.cuda.is_available() else "gloo")
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    os.environ.setdefault("MASTER_PORT", "29500")
    print('| distributed init (rank {}): env://, gpu {}'.format(args.rank, args.gpu), flush=True)
    kw = {}
    if args.dist_backend == "nccl":
        kw["device_id"] = torch.device("cuda", args.gpu)
    dist.init_process_group(backend=args.dist_backend, init_method="env://", world_size=args.world_size, rank=args.rank, **kw)
    dist.barrier()


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
