"""Optimizers whose dense update also runs through the HIP library.

`FusedSGD` is a torch.optim.SGD (so the reference loop's zero_grad / step / lr_scheduler calls and the
embedding update hook treat it as such) whose dense step is ONE `dlrm_sgd_dense_multi` launch per param group
instead of torch's foreach kernels.  `FusedRWSAdagrad` mirrors optim/rwsadagrad.py.  Embedding tables never carry a `.grad` (fused sparse update), so
they are skipped here exactly as torch.optim.SGD skips them."""
from __future__ import annotations

import torch

from . import ops


class FusedSGD(torch.optim.SGD):
    def __init__(self, params, lr: float = 1e-3):
        super().__init__(params, lr=lr, momentum=0, dampening=0, weight_decay=0, nesterov=False)

    @torch.no_grad()
    def step(self, closure=None):
        loss = None
        if closure is not None:
            with torch.enable_grad():
                loss = closure()
        for group in self.param_groups:
            lr = ops.device_lr(group, float(group["lr"]))      # (a device scalar while a whole-step graph captures: graph.py)
            ws, gs = [], []
            for p in group["params"]:
                g = p.grad
                if g is None:
                    continue
                if g.is_sparse:
                    raise RuntimeError("FusedSGD: sparse gradients are handled by the fused embedding update")
                if not g.is_contiguous():
                    g = g.contiguous()
                ws.append(p.data)
                gs.append(g)
            ops.sgd_dense_multi(ws, gs, lr)       # one launch per param group (pointers by value in the kernarg)
        return loss


class FusedRWSAdagrad(torch.optim.Optimizer):
    """Row-wise sparse Adagrad with the reference's hyper-parameters and state layout (optim/rwsadagrad.py:19-152):
    `state[p]["step"]`, `state[p]["sum"]` for dense parameters, `state[p]["momentum"]` ([rows] fp32) for embedding
    tables.  Dense parameters step through `dlrm_adagrad_dense`; embedding tables never carry a `.grad` here — the
    optimizer-step pre-hook of DLRM_Net applies the fused backward + row-wise update (`dlrm_emb_bwd_rowwise_adagrad`)
    with this optimizer's lr / lr_decay / eps and keeps the row-wise state in `state[p]["momentum"]`."""

    def __init__(self, params, lr=1e-2, lr_decay=0.0, weight_decay=0.0, initial_accumulator_value=0.0, eps=1e-10):
        if not 0.0 <= lr:
            raise ValueError("Invalid learning rate: {}".format(lr))
        if not 0.0 <= lr_decay:
            raise ValueError("Invalid lr_decay value: {}".format(lr_decay))
        if weight_decay != 0.0:
            raise ValueError("FusedRWSAdagrad: weight_decay is not supported (the reference rejects it for sparse gradients)")
        if not 0.0 <= initial_accumulator_value:
            raise ValueError("Invalid initial_accumulator_value value: {}".format(initial_accumulator_value))
        if not 0.0 <= eps:
            raise ValueError("Invalid epsilon value: {}".format(eps))
        defaults = dict(lr=lr, lr_decay=lr_decay, eps=eps, weight_decay=weight_decay,
                        initial_accumulator_value=initial_accumulator_value)
        super().__init__(params, defaults)
        for group in self.param_groups:
            for p in group["params"]:
                self.state[p]["step"] = 0

    @torch.no_grad()
    def step(self, closure=None):
        loss = None
        if closure is not None:
            with torch.enable_grad():
                loss = closure()
        for group in self.param_groups:
            for p in group["params"]:
                g = p.grad
                if g is None:
                    continue
                if g.is_sparse:
                    raise RuntimeError("FusedRWSAdagrad: sparse gradients are handled by the fused embedding update")
                state = self.state[p]
                if "sum" not in state:
                    state["sum"] = torch.full_like(p.data, self.defaults["initial_accumulator_value"], dtype=torch.float32)
                state["step"] += 1
                clr = group["lr"] / (1.0 + (state["step"] - 1.0) * group["lr_decay"])
                if group["lr_decay"] == 0:
                    clr = ops.device_lr(group, clr)          # (clr == lr: a device scalar while a whole-step graph captures)
                ops.adagrad_dense(p.data, state["sum"], g if g.is_contiguous() else g.contiguous(), clr, group["eps"])
        return loss
