"""The model semantics of the reference's torchrec trainer (BASELINE.json configs[4]) on the same HIP kernels.

`torchrec_dlrm/dlrm_main.py:598-653` builds `torchrec.models.dlrm.DLRM` and wraps it in `DLRMTrain`.  torchrec itself is a
third-party dependency that is NOT vendored in the reference (requirements.txt:9, unpinned nightly) and is not installed here,
so this module restates its published model (torchrec/models/dlrm.py: SparseArch, DenseArch, InteractionArch, OverArch,
DLRMTrain) — parity for THIS variant is therefore "unpinned": it is checked against the CPU oracle's restatement and against
torch operators, not against torchrec output.  What differs from `dlrm_s_pytorch.DLRM_Net` (same math otherwise):
  * InteractionArch keeps the strictly UPPER triangle in `torch.triu_indices(F, F, offset=1)` order — the same pairwise dots
    as the reference's tril order, permuted columns (interaction mode 2 of `dlrm_interact_fwd`);
  * OverArch ends with a bare Linear: the model returns LOGITS (no sigmoid);
  * DLRMTrain.forward(batch) -> (loss, (loss.detach(), logits.detach(), labels)) with BCEWithLogitsLoss (`dlrm_bce_logits_loss`);
  * the embedding optimizer is applied in backward (`apply_optimizer_in_backward`, dlrm_main.py:647-651): here the fused sparse
    update (row-wise Adagrad / SGD) launched by the optimizer-step hook — or, with `overlap_streams`, during backward itself.
Inputs are the multi-hot batches of `dlrm_amd.multihot.Multihot` (int32 ids, per-table local offsets).
"""
from __future__ import annotations

from typing import Sequence

import numpy as np
import torch
import torch.nn as nn

from .dlrm_net import DLRM_Net, FusedMLP
from .functional import BCEWithLogitsLossFunction


class FusedBCEWithLogitsLoss(nn.Module):
    def forward(self, logits, target):
        return BCEWithLogitsLossFunction.apply(logits, target)


class DLRM(DLRM_Net):
    """torchrec.models.dlrm.DLRM(embedding_bag_collection, dense_in_features, dense_arch_layer_sizes, over_arch_layer_sizes):
    the embedding bag collection is given by its table sizes and the embedding dimension."""

    def __init__(self, num_embeddings_per_feature: Sequence[int], embedding_dim: int, dense_in_features: int,
                 dense_arch_layer_sizes: Sequence[int], over_arch_layer_sizes: Sequence[int]):
        if list(dense_arch_layer_sizes)[-1] != embedding_dim:
            raise ValueError("dense_arch_layer_sizes[-1] must equal the embedding dimension (torchrec DLRM asserts the same)")
        T = len(num_embeddings_per_feature)
        F = T + 1
        ln_bot = np.asarray([dense_in_features] + list(dense_arch_layer_sizes))
        ln_top = np.asarray([embedding_dim + F * (F - 1) // 2] + list(over_arch_layer_sizes))
        super().__init__(embedding_dim, np.asarray(list(num_embeddings_per_feature)), ln_bot, ln_top, arch_interaction_op="dot",
                         arch_interaction_itself=False, sigmoid_bot=-1, sigmoid_top=-1, loss_function="bce")
        # OverArch: MLP(ReLU) over all but the last size, then a bare Linear -> logits
        arith = self.top_l.arith
        self.top_l = FusedMLP(*list(self.top_l.children())[:-1])
        self.top_l.arith = arith
        self.interaction_order = "triu"
        self.loss_fn = FusedBCEWithLogitsLoss()


class DLRMTrain(nn.Module):
    """torchrec.models.dlrm.DLRMTrain: forward(dense_features, lS_o, lS_i, labels) -> (loss, (loss.detach(), logits.detach(),
    labels)); `labels` [B] or [B, 1]."""

    def __init__(self, dlrm_module: DLRM):
        super().__init__()
        self.model = dlrm_module
        self.loss_fn = FusedBCEWithLogitsLoss()

    def forward(self, dense_features, lS_o, lS_i, labels):
        logits = self.model(dense_features, lS_o, lS_i)
        lab = labels.to(torch.float32).reshape(logits.shape)
        loss = self.loss_fn(logits, lab)
        return loss, (loss.detach(), logits.detach(), labels)
