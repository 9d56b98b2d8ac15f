"""Adam on the hand-written HIP kernel (csrc/adam.hip) -- the optimizer step of the training loop (train_place_recognition.py:386-392 builds
``torch.optim.Adam(model.parameters(), lr)``).

Same hyper-parameters, same update rule and the same ``state_dict`` layout as ``torch.optim.Adam`` (per parameter: ``step``, ``exp_avg``,
``exp_avg_sq``), so checkpoints move both ways (``state_dict()`` writes an own copy of the counter per parameter, ``load_state_dict()`` drops
the device scalars of the steps taken before it); ``amsgrad`` / ``maximize`` / ``foreach`` variants are not built (the reference uses none).
The step counter and the learning rate live on the device (one scalar each per parameter group; the counter is shared by its parameters'
``state['step']``) and the tensor list travels in the kernel arguments: an optimizer step is ``1 + ceil(n / 84)`` launches with nothing
host-side that changes between steps in the arithmetic, so ``train.GraphedTrainer`` captures it like torch's ``capturable=True`` optimizers --
and a learning-rate scheduler (train_place_recognition.py:531-568 steps one per epoch) still takes effect on the captured graph:
``sync_hyperparameters()`` writes a changed ``param_groups[i]['lr']`` into the group's device scalar (``step()`` calls it when it is not
being captured, ``GraphedTrainer.step`` calls it before every replay).  betas / eps / weight_decay are launch constants: changing them after
a capture needs a new capture (``GraphedTrainer`` checks).  CPU parameters are refused (no fallback)."""
import ctypes

import torch

from ._lib import call, ptr


class Adam(torch.optim.Optimizer):
    def __init__(self, params, lr=1e-3, betas=(0.9, 0.999), eps=1e-8, weight_decay=0.0, capturable=True, fused=True):
        if lr < 0 or eps < 0 or not 0 <= betas[0] < 1 or not 0 <= betas[1] < 1 or weight_decay < 0:
            raise ValueError("Adam: invalid hyper-parameter")
        # capturable / fused are accepted for signature compatibility with torch.optim.Adam call sites: this optimizer is always both
        super().__init__(params, dict(lr=lr, betas=betas, eps=eps, weight_decay=weight_decay, capturable=True, fused=True))
        self._step = {}          # group index -> the group's device step counter (kept out of param_groups: state_dict() copies those)
        self._lr = {}            # group index -> (device scalar, the host value it holds)
        self._keep = {}          # group index -> gradient tensors of the launches in flight

    def state_dict(self):
        """torch.optim.Adam's layout with ONE step tensor PER PARAMETER: the live state shares a single device scalar per group, and that aliasing
        would survive torch.save / torch.load into a torch.optim.Adam, whose ``_foreach_add_`` then advances the shared counter once per
        parameter per step (bias correction wrong from the first resumed step)."""
        sd = super().state_dict()
        for st in sd["state"].values():
            if torch.is_tensor(st.get("step")):
                st["step"] = st["step"].detach().clone()
        return sd

    def load_state_dict(self, state_dict):
        """The loaded counters / learning rates replace whatever this optimizer held: the per-group device scalars of earlier steps are
        dropped, so the next step adopts the checkpoint's counter (``_init_group``) and re-creates the learning-rate scalar from the loaded
        ``param_groups``.  A hipGraph captured before the load holds the OLD scalars' addresses and must be re-captured (GraphedTrainer does
        not survive a load, like torch's capturable optimizers)."""
        super().load_state_dict(state_dict)
        self._step.clear()
        self._lr.clear()
        self._keep.clear()

    def _init_group(self, gi, group):
        ps = [p for p in group["params"] if p.grad is not None]
        if not ps:
            return None
        step = self._step.get(gi)
        for p in ps:
            if not p.is_cuda or p.dtype != torch.float32 or not p.is_contiguous():
                raise RuntimeError("patchaugnet_amd.optim.Adam updates contiguous fp32 parameters on the MI355X only (no CPU path)")
            if p.grad.is_sparse:
                raise RuntimeError("Adam does not support sparse gradients")
            st = self.state[p]
            if not st:
                if step is None:
                    step = self._step[gi] = torch.zeros((), dtype=torch.float32, device=p.device)
                st["step"] = step                                  # one device scalar per group, shared
                st["exp_avg"] = torch.zeros_like(p, memory_format=torch.preserve_format)
                st["exp_avg_sq"] = torch.zeros_like(p, memory_format=torch.preserve_format)
            elif step is None:                                     # state restored by load_state_dict: adopt the first parameter's counter
                step = self._step[gi] = st["step"].to(device=p.device, dtype=torch.float32).reshape(()).clone()
            if st["step"] is not step:
                st["step"] = step
        return ps

    @torch.no_grad()
    def sync_hyperparameters(self):
        """Write every group's current ``lr`` into its device scalar if it changed (a scheduler stepped).  A no-op launch-wise otherwise; must not
        be called while a hipGraph is being captured (the fill would be baked into the graph as a constant)."""
        for gi, group in enumerate(self.param_groups):
            ent = self._lr.get(gi)
            if ent is None:
                ps = [p for p in group["params"] if p.is_cuda]
                if not ps:
                    continue
                ent = self._lr[gi] = [torch.full((), float(group["lr"]), dtype=torch.float32, device=ps[0].device), float(group["lr"])]
            elif ent[1] != float(group["lr"]):
                ent[0].fill_(float(group["lr"]))
                ent[1] = float(group["lr"])

    @torch.no_grad()
    def step(self, closure=None):
        loss = None
        if closure is not None:
            with torch.enable_grad():
                loss = closure()
        if not torch.cuda.is_current_stream_capturing():
            self.sync_hyperparameters()
        for gi, group in enumerate(self.param_groups):
            ps = self._init_group(gi, group)
            if ps is None:
                continue
            if gi not in self._lr:
                raise RuntimeError("patchaugnet_amd.optim.Adam: first step of a parameter group inside a hipGraph capture; run one eager step (or "
                                   "sync_hyperparameters()) before capturing")
            n = len(ps)
            grads = [p.grad if p.grad.is_contiguous() else p.grad.contiguous() for p in ps]
            arr = lambda ts: (ctypes.c_void_p * n)(*[t.data_ptr() for t in ts])
            numel = (ctypes.c_long * n)(*[p.numel() for p in ps])
            b1, b2 = group["betas"]
            with torch.cuda.device(ps[0].device):
                step = self._step[gi]
                call("pa_adam_tick", ptr(step))
                call("pa_adam_step", n, arr(ps), arr(grads), arr([self.state[p]["exp_avg"] for p in ps]), arr([self.state[p]["exp_avg_sq"] for p in ps]),
                     numel, ptr(step), ptr(self._lr[gi][0]), float(b1), float(b2), float(group["eps"]), float(group["weight_decay"]))
            self._keep[gi] = grads                                 # the launches read these buffers asynchronously
        return loss
