"""Multi-threaded CPU port of the reference training step — TEST INFRASTRUCTURE / cpu_baseline ONLY.

The reference cannot travel to the GPU box (it is Python under /root/reference), so bench.py's
`cpu_baseline` ("kind": "port") times THIS file there: the same PyTorch CPU operator call sites the
reference's hot path makes, on all host cores —
    F.embedding_bag(mode="sum", sparse=True)        dlrm_s_pytorch.py:277,452-457
    F.linear + relu / sigmoid                        dlrm_s_pytorch.py:216,238-241
    cat -> bmm -> Z[:, li, lj] -> cat                dlrm_s_pytorch.py:483-504
    binary_cross_entropy(mean) / mse                 dlrm_s_pytorch.py:386-393
    backward() and torch.optim.SGD.step (sparse + dense)   dlrm_s_pytorch.py:1611-1621
written as plain functions over a parameter dict (state_dict names).  tests/test_oracle_golden.py pins it
bit-for-bit against the golden vectors of the reference.
"""
from __future__ import annotations

from typing import Dict, List

import torch
import torch.nn.functional as F


class TorchPortDLRM:
    def __init__(self, params: Dict[str, torch.Tensor], sigmoid_top: int, self_interaction: bool = False,
                 loss: str = "bce", lr: float = 0.1, interaction: str = "dot", loss_threshold: float = 0.0, loss_ws=None):
        # cat interaction :505-507, --loss-threshold clamp :607-610, wbce :388-391 + loss_fn_wrap :150-156
        self.interaction, self.loss_threshold = interaction, float(loss_threshold)
        self.loss_ws = None if loss_ws is None else torch.as_tensor(loss_ws, dtype=torch.float64)
        self.p = {k: v.detach().clone().requires_grad_(True) for k, v in params.items()}
        self.T = sum(1 for k in self.p if k.startswith("emb_l.") and k.endswith(".weight"))
        self.nbot = sum(1 for k in self.p if k.startswith("bot_l.") and k.endswith(".weight"))
        self.ntop = sum(1 for k in self.p if k.startswith("top_l.") and k.endswith(".weight"))
        self.sigmoid_top, self.self_interaction, self.loss = sigmoid_top, self_interaction, loss
        self.opt = torch.optim.SGD(list(self.p.values()), lr=lr)
        self._pairs = None

    def _tower(self, x, name, n, sig):
        for i in range(n):
            x = F.linear(x, self.p[f"{name}.{2 * i}.weight"], self.p[f"{name}.{2 * i}.bias"])
            x = torch.sigmoid(x) if i == sig else torch.relu(x)
        return x

    def forward(self, X, lS_o: List[torch.Tensor], lS_i: List[torch.Tensor]):
        x = self._tower(X, "bot_l", self.nbot, -1)
        # --weighted-pooling=learned: per_sample_weights = v_W_l[k].gather(0, indices)   dlrm_s_pytorch.py:425-426
        ly = [F.embedding_bag(lS_i[k], self.p[f"emb_l.{k}.weight"], lS_o[k], mode="sum", sparse=True,
                              per_sample_weights=(self.p[f"v_W_l.{k}"].gather(0, lS_i[k]) if f"v_W_l.{k}" in self.p else None))
              for k in range(self.T)]
        B, D = x.shape
        if self.interaction == "cat":
            p = self._tower(torch.cat([x] + ly, dim=1), "top_l", self.ntop, self.sigmoid_top)
            return self._clamp(p)
        Tm = torch.cat([x] + ly, dim=1).view(B, -1, D)
        Z = torch.bmm(Tm, Tm.transpose(1, 2))
        if self._pairs is None:
            nf = Tm.size(1)
            off = 1 if self.self_interaction else 0
            self._pairs = (torch.tensor([i for i in range(nf) for _ in range(i + off)]),
                           torch.tensor([j for i in range(nf) for j in range(i + off)]))
        R = torch.cat([x, Z[:, self._pairs[0], self._pairs[1]]], dim=1)
        return self._clamp(self._tower(R, "top_l", self.ntop, self.sigmoid_top))

    def _clamp(self, p):
        if 0.0 < self.loss_threshold < 1.0:
            return torch.clamp(p, min=self.loss_threshold, max=1.0 - self.loss_threshold)
        return p

    def train_step(self, X, lS_o, lS_i, target):
        Z = self.forward(X, lS_o, lS_i)
        if self.loss == "wbce":
            E = (self.loss_ws[target.view(-1).long()].view_as(target) * F.binary_cross_entropy(Z, target, reduction="none")).mean()
        else:
            E = F.binary_cross_entropy(Z, target) if self.loss == "bce" else F.mse_loss(Z, target)
        self.opt.zero_grad()
        E.backward()
        self.opt.step()
        return float(E.detach()), Z.detach()
