"""The training step replayed from ONE captured hipGraph - for the shapes where the eager step is bound by the host's launch rate.

The reference pre-trains at a per-GPU batch of 4 (/root/reference/egs/audioset/run_pretrain_base.sh:30-31): one step of
/root/reference/src/traintest_cavmae_base.py:131-152 is then ~800 kernel launches of a few microseconds each, and the Python /
ctypes / HIP launch path (several microseconds per launch) - not the GPU - sets the step time.  `GraphedTrainStep` captures
`traintest_cavmae_base.train_step` once (torch.cuda.CUDAGraph = hipGraph on ROCm: every kernel of libavsiam_hip.so is launched on
torch's current stream, so stream capture records them - both HIP streams of the backward, joined before Adam) and replays it.

What changes from step to step must not be a kernel ARGUMENT (frozen at capture):
  * the mask plans are still drawn per step: the Philox key lives in device memory (`avs_mask_plan_dev`) and a node of the graph
    advances it; the host part of the contrastive draw (two batch permutations, the structured time / frequency picks:
    cav_mae_base.py:533-538, 415-422 - numpy + three small host-to-device copies into fixed buffers) runs in front of each replay;
  * Adam's step count lives in device memory (`avs_adam_dev`), advanced by a node of the graph; the bias corrections are evaluated in
    the kernel (double, like torch.optim.Adam);
  * the batch: `a` / `v` are FIXED device buffers - copy each new batch into them (`.copy_`) before `step()`.
The counters are re-written from the model's host-side state and the host part of the contrastive draw is made in front of EVERY replay
(three fill kernels, three small copies), so eager steps and replayed steps may be mixed freely; results equal the eager step's (same kernels, same order, same keys -
tests/test_train_gpu.py::test_graphed_step_equals_the_eager_step).

Single GPU only: the data-parallel reducer issues its collectives from the host as the backward proceeds."""
import torch

from . import _lib
from .param_spec import P1, P2
from .traintest_cavmae_base import train_step


class GraphedTrainStep:
    def __init__(self, model, a, v, lr, warmup=2):
        if model._dp:
            raise RuntimeError("GraphedTrainStep: single GPU only (the gradient all-reduce is issued from the host during backward)")
        if model.publish_grads:
            raise RuntimeError("GraphedTrainStep: set model.publish_grads = False (the fused step does not hand .grad views to autograd)")
        if model.share_pass_buffers:
            raise RuntimeError("GraphedTrainStep: not with share_pass_buffers (the pool's ownership checks are host-side state)")
        self.model, self.lr, self.a, self.v = model, float(lr), a, v
        B = a.shape[0]
        cur = torch.cuda.current_stream()
        side = torch.cuda.Stream(device=a.device)
        side.wait_stream(cur)
        with torch.cuda.stream(side):                     # eager warm-up: engines, optimizer state, kernel attributes, shadows
            for _ in range(max(1, warmup)):               # (REAL training steps on the contents of a / v - a loop that captures on its first
                self.warm_out = train_step(model, a, v, self.lr)      #  batch asks for warmup=1 and takes this step's losses as that batch's)
        cur.wait_stream(side)
        torch.cuda.synchronize()
        dev = a.device
        self.seed_dev = torch.zeros(1, dtype=torch.int64, device=dev)
        self.steps_dev = {P1: torch.zeros(1, dtype=torch.int32, device=dev), P2: torch.zeros(1, dtype=torch.int32, device=dev)}
        self.eng_c = model._engine("contrastive", B)
        self._sync_counters()
        self.graph = torch.cuda.CUDAGraph()
        n0 = _lib.calls
        model._graph = {"seed": self.seed_dev, "steps": self.steps_dev}
        try:
            # Nothing is drawn on the host here: the capture records the kernels with whatever the descriptor buffers hold (the warm-up's
            # draw), and EVERY replay is preceded by its own draw_host() in step() - so an eager draw on the same engine between the capture
            # and a replay (validate() at an epoch end, a recapture after a learning-rate change) cannot hand its permutation to the
            # replayed step, and the numpy generator advances exactly once per step whichever way the step runs.
            # capture_error_mode "thread_local": a HIP call from ANOTHER thread during the capture window (a DataLoader's pin-memory thread,
            # a checkpoint writer) must not invalidate the capture; the step's own two streams belong to this thread.
            with torch.cuda.graph(self.graph, capture_error_mode="thread_local"):
                self.out = train_step(model, a, v, self.lr)
        finally:
            model._graph = None
        self.kernel_nodes = _lib.calls - n0               # launches of libavsiam_hip.so held by the graph (torch's own few nodes come on top)
        torch.cuda.synchronize()

    def _key(self):
        m = self.model
        m._rngs()
        k = ((m._seed_base & 0xFFFFFFFF) << 32) | ((getattr(m, "_draws", 0) + 1) & 0xFFFFFFFF)       # the key the NEXT draw takes (model._next_seed)
        return k - (1 << 64) if k >= (1 << 63) else k                                                # as the int64 with the same bits

    def _sync_counters(self):
        m = self.model
        self.seed_dev.fill_(self._key())
        for w in (P1, P2):
            self.steps_dev[w].fill_(int(m._opt_state[w]["step"]))

    def step(self):
        """one training step on the current contents of `a` / `v`; returns the step's device scalars
        (loss_pass2, loss_mae_a, loss_mae_v, loss_c, c_acc) - the SAME tensors every call (copy them to keep a history)"""
        m = self.model
        self._sync_counters()
        self.eng_c.draw_host(m._np_rng())                 # this replay's contrastive permutations / structured picks (host part of the draw)
        self.graph.replay()
        m._draws = getattr(m, "_draws", 0) + 2            # two plans were drawn (contrastive, MAE)
        for w in (P1, P2):
            m._opt_state[w]["step"] += 1
        return self.out
