"""Adam on the hand-written HIP kernel (csrc/adam.hip) -- the optimizer step of the training loop (train_place_recognition.py:386-392 builds
``torch.optim.Adam(model.parameters(), lr)``).

Same hyper-parameters, same update rule and the same ``state_dict`` layout as ``torch.optim.Adam`` (per parameter: ``step``, ``exp_avg``,
``exp_avg_sq``), so checkpoints move both ways; ``amsgrad`` / ``maximize`` / ``foreach`` variants are not built (the reference uses none).
The step counter lives on the device (one scalar per parameter group, shared by its parameters' ``state['step']``) and the tensor list
travels in the kernel arguments: an optimizer step is ``1 + ceil(n / 84)`` launches with nothing host-side in the arithmetic, so
``train.GraphedTrainer`` captures it like torch's ``capturable=True`` optimizers.  CPU parameters are refused (no fallback)."""
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
        self._keep = {}          # group index -> gradient tensors of the launches in flight

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
    def step(self, closure=None):
        loss = None
        if closure is not None:
            with torch.enable_grad():
                loss = closure()
        for gi, group in enumerate(self.param_groups):
            ps = self._init_group(gi, group)
            if ps is None:
                continue
            n = len(ps)
            grads = [p.grad if p.grad.is_contiguous() else p.grad.contiguous() for p in ps]
            arr = lambda ts: (ctypes.c_void_p * n)(*[t.data_ptr() for t in ts])
            numel = (ctypes.c_long * n)(*[p.numel() for p in ps])
            b1, b2 = group["betas"]
            with torch.cuda.device(ps[0].device):
                step = self._step[gi]
                call("pa_adam_tick", ptr(step))
                call("pa_adam_step", n, arr(ps), arr(grads), arr([self.state[p]["exp_avg"] for p in ps]), arr([self.state[p]["exp_avg_sq"] for p in ps]),
                     numel, ptr(step), float(group["lr"]), float(b1), float(b2), float(group["eps"]), float(group["weight_decay"]))
            self._keep[gi] = grads                                 # the launches read these buffers asynchronously
        return loss
