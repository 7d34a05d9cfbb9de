"""Optimizers whose dense update also runs through the HIP library.

`FusedSGD` is a torch.optim.SGD (so the reference loop's zero_grad / step / lr_scheduler calls and the
embedding update hook treat it as such) whose dense step launches `dlrm_sgd_dense` per parameter
instead of torch's foreach kernels.  Embedding tables never carry a `.grad` (fused sparse update), so
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
            lr = float(group["lr"])
            for p in group["params"]:
                g = p.grad
                if g is None:
                    continue
                if g.is_sparse:
                    raise RuntimeError("FusedSGD: sparse gradients are handled by the fused embedding update")
                if not g.is_contiguous():
                    g = g.contiguous()
                ops.sgd_dense(p.data, g, lr)
        return loss
