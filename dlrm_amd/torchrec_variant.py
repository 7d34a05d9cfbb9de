"""The model semantics of the reference's torchrec trainer (BASELINE.json configs[4]) on the same HIP kernels.

`torchrec_dlrm/dlrm_main.py:598-653` builds `torchrec.models.dlrm.DLRM` and wraps it in `DLRMTrain`.  torchrec itself is a
third-party dependency that is NOT vendored in the reference (requirements.txt:9, unpinned nightly) and is not installed here,
so this module restates its published model (torchrec/models/dlrm.py: SparseArch, DenseArch, InteractionArch, OverArch,
DLRMTrain) — parity for THIS variant is therefore "unpinned": it is checked against the test suite's CPU restatement and against
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


class ShardedDLRM(DLRM_Net):
    """the same model on N ranks with PLANNED sharding (dlrm_amd.sharding.plan) and NON-replicated inputs.

      * every rank feeds only its batch slice: dense [B/N, 13] and key-major ids (`values`: for table t the B/N * hot[t] ids of
        its samples — the per-rank KJT of the reference's torchrec loader, multi_hot_criteo.py:200-214);
      * `ext_dist.kjt_input_dist`: one all-to-all of ids brings every table-wise table's whole-batch ids to its owner, one
        all-gather gives every rank the whole-batch ids of the row-wise tables;
      * table-wise tables: pooled for the whole batch by the owner, pooled rows to the sample owners by the existing all-to-all;
      * row-wise tables: every rank pools the rows of ITS contiguous row range (ids of other ranges are skipped by the kernel:
        `BagBatch.ignore_oob`), the partial sums meet in a reduce-scatter over the batch; backward all-gathers the gradient rows
        and every rank updates its own rows;
      * interaction on the local batch slice reads the all-to-all blocks and the reduce-scatter block IN PLACE in global table
        order through the kernel's {pointer, stride} table (`InteractFunction(order=...)`).
    Embedding gradients are not divided by N (the reference's behaviour in both trainers: the loss is the mean over the LOCAL
    batch; DDP averages only the dense parameters)."""

    def __init__(self, num_embeddings_per_feature: Sequence[int], multi_hot_sizes: Sequence[int], embedding_dim: int,
                 dense_in_features: int, dense_arch_layer_sizes: Sequence[int], over_arch_layer_sizes: Sequence[int],
                 global_batch: int, plan=None):
        from . import ext_dist, sharding
        from .dlrm_net import EmbeddingUpdateHook
        super().__init__()                                   # empty shell: flags, pending list, stream state
        N, me = max(ext_dist.my_size, 1), max(ext_dist.my_rank, 0)
        rows, hot, D = [int(n) for n in num_embeddings_per_feature], [int(h) for h in multi_hot_sizes], int(embedding_dim)
        self.plan = plan if plan is not None else sharding.plan(rows, hot, D, N, global_batch)
        self.rows, self.hot, self.m_spa, self.world, self.rank = rows, hot, D, N, me
        self.arch_interaction_op, self.arch_interaction_itself, self.interaction_order = "dot", False, "triu"
        self.loss_threshold, self.weighted_pooling, self.quantize_emb, self.ndevices = 0.0, None, False, -1
        self.tw_owner = [-1] * len(rows)
        for s_ in self.plan.shards:
            if s_.kind == "table":
                self.tw_owner[s_.table] = s_.rank
        self.rw_tables = self.plan.row_wise()
        self.tw_mine = self.plan.table_wise(me)
        self.tw_per_rank = self.plan.tables_per_rank()
        if N > 1 and min(self.tw_per_rank) == 0:
            raise ValueError("ShardedDLRM: every rank needs at least one table-wise table (plan %s)" % self.tw_per_rank)
        self.rw_range = {s_.table: s_.row_ranges[me] for s_ in self.plan.shards if s_.kind == "row"}
        T = len(rows)
        F = T + 1
        ln_bot = np.asarray([dense_in_features] + list(dense_arch_layer_sizes))
        ln_top = np.asarray([D + F * (F - 1) // 2] + list(over_arch_layer_sizes))
        local_rows = [rows[t] for t in self.tw_mine] + [self.rw_range[t][1] - self.rw_range[t][0] for t in self.rw_tables]
        saved = ext_dist.my_size, ext_dist.force_distributed
        ext_dist.my_size, ext_dist.force_distributed = 1, False     # create_emb: build exactly the listed (local) tables
        try:
            self.emb_l, self.v_W_l = self.create_emb(D, np.asarray(local_rows), None)
        finally:
            ext_dist.my_size, ext_dist.force_distributed = saved
        self.bot_l = self.create_mlp(ln_bot, -1)
        top = self.create_mlp(ln_top, -1)
        self.top_l = FusedMLP(*list(top.children())[:-1])    # bare last Linear -> logits
        self.loss_fn = FusedBCEWithLogitsLoss()
        # canonical feature f (0 = dense, 1 + t = table t) -> position in the block list [x | a2a block of rank 0.. | rw block]
        pos, k = {}, 1
        for r in range(N):
            for t in self.plan.table_wise(r):
                pos[t] = k
                k += 1
        for t in self.rw_tables:
            pos[t] = k
            k += 1
        self.feature_order = [0] + [pos[t] for t in range(T)]
        self._offs = {}
        EmbeddingUpdateHook.register(self)

    def load_full_state(self, full: dict) -> None:
        """copy this rank's shards out of a full (single-process) state_dict: emb_l.{t}.weight, bot_l.*, top_l.*"""
        with torch.no_grad():
            j = 0
            for t in self.tw_mine:
                self.emb_l[j].weight.copy_(torch.as_tensor(full[f"emb_l.{t}.weight"]))
                j += 1
            for t in self.rw_tables:
                lo, hi = self.rw_range[t]
                self.emb_l[j].weight.copy_(torch.as_tensor(full[f"emb_l.{t}.weight"])[lo:hi])
                j += 1
            for name, p in list(self.bot_l.named_parameters()):
                p.copy_(torch.as_tensor(full[f"bot_l.{name}"]))
            for name, p in list(self.top_l.named_parameters()):
                p.copy_(torch.as_tensor(full[f"top_l.{name}"]))

    def _bag_starts(self, B: int, h: int, like: torch.Tensor) -> torch.Tensor:
        key = (B, h, like.device, like.dtype)
        if key not in self._offs:
            self._offs[key] = torch.arange(B, device=like.device, dtype=like.dtype) * h
        return self._offs[key]

    def forward(self, dense_x, values):
        from . import ext_dist, ops
        from .functional import EmbeddingBagsFunction, InteractFunction
        N, D = self.world, self.m_spa
        Bl = dense_x.size(0)
        B = Bl * N
        ops.check_index_errors()
        dist_on = N > 1 or ext_dist.is_distributed()         # (a forced one-rank RCCL group runs every collective as a self-exchange)
        if dist_on:
            tw, rw = ext_dist.kjt_input_dist(values, self.hot, self.tw_owner, self.rw_tables)
        else:                                                # one rank: every table is "mine", nothing is exchanged
            seg, tw, rw = 0, {}, {}
            for t, h in enumerate(self.hot):
                (rw if t in self.rw_tables else tw)[t] = values[seg:seg + Bl * h]
                seg += Bl * h
        n_tw = len(self.tw_mine)
        w_tw = [self.emb_l[j].weight for j in range(n_tw)]
        w_rw = [self.emb_l[n_tw + j].weight for j in range(len(self.rw_tables))]
        bags = ops.BagBatch([self._bag_starts(B, self.hot[t], values) for t in self.tw_mine], [tw[t] for t in self.tw_mine])
        E_tw = EmbeddingBagsFunction.apply(self._stash_embedding_grad, bags, None, *w_tw)          # [B, n_tw * D]
        blocks = []
        if dist_on:
            req = ext_dist.alltoall([E_tw], self.tw_per_rank, emb_dim=D)
        if self.rw_tables:
            ids = []
            for t in self.rw_tables:                         # ids of other ranks' rows become -1: skipped by the kernels
                lo, hi = self.rw_range[t]
                v = rw[t]
                ids.append(torch.where((v >= lo) & (v < hi), v - lo, torch.full_like(v, -1)))
            bags_rw = ops.BagBatch([self._bag_starts(B, self.hot[t], values) for t in self.rw_tables], ids)
            bags_rw.ignore_oob = True
            E_rw = EmbeddingBagsFunction.apply(self._stash_embedding_grad, bags_rw, None, *w_rw)   # partial sums, whole batch
            E_rw = ext_dist.reduce_scatter_rows(E_rw)                                              # [B/N, n_rw * D]
        x = self.apply_mlp(dense_x, self.bot_l)
        blocks = [x] + (list(req.wait()) if dist_on else [E_tw]) + ([E_rw] if self.rw_tables else [])
        z = InteractFunction.apply(D, self._interaction_mode(), True, list(self.feature_order), *blocks)
        return self.apply_mlp(z, self.top_l)


class LowRankCrossNet(nn.Module):
    """torchrec.modules.crossnet.LowRankCrossNet(in_features, num_layers, low_rank): parameters V_kernels[l] [low_rank, in],
    W_kernels[l] [in, low_rank] (xavier-normal), bias[l] [in] (zeros); x_{l+1} = x_0 * (W_l (V_l x_l) + b_l) + x_l."""

    arith = "f32"

    def __init__(self, in_features: int, num_layers: int, low_rank: int):
        super().__init__()
        self.V_kernels, self.W_kernels, self.bias = nn.ParameterList(), nn.ParameterList(), nn.ParameterList()
        std = np.sqrt(2.0 / (in_features + low_rank))
        for _ in range(num_layers):
            self.V_kernels.append(nn.Parameter(torch.tensor(np.random.normal(0.0, std, size=(low_rank, in_features)).astype(np.float32))))
            self.W_kernels.append(nn.Parameter(torch.tensor(np.random.normal(0.0, std, size=(in_features, low_rank)).astype(np.float32))))
            self.bias.append(nn.Parameter(torch.zeros(in_features, dtype=torch.float32)))

    def forward(self, x0):
        from . import ops
        from .functional import LowRankCrossNetFunction
        flat = [p for l in range(len(self.bias)) for p in (self.V_kernels[l], self.W_kernels[l], self.bias[l])]
        return LowRankCrossNetFunction.apply(ops.arith_code(self.arith), x0, *flat)


class DLRM_DCN(DLRM_Net):
    """torchrec.models.dlrm.DLRM_DCN (the MLPerf-v2 model, torchrec_dlrm/dlrm_main.py:608-619): dense arch and pooled embeddings
    are concatenated to [B, F*D] — the feature buffer the bottom tower and the embedding kernel write side by side — and passed
    through a DCN-v2 low-rank cross network; the over-arch takes its [B, F*D] output and ends with a bare Linear (logits)."""

    def __init__(self, num_embeddings_per_feature: Sequence[int], embedding_dim: int, dense_in_features: int,
                 dense_arch_layer_sizes: Sequence[int], over_arch_layer_sizes: Sequence[int], dcn_num_layers: int,
                 dcn_low_rank_dim: int):
        if list(dense_arch_layer_sizes)[-1] != embedding_dim:
            raise ValueError("dense_arch_layer_sizes[-1] must equal the embedding dimension")
        F = len(num_embeddings_per_feature) + 1
        ln_bot = np.asarray([dense_in_features] + list(dense_arch_layer_sizes))
        ln_top = np.asarray([F * embedding_dim] + list(over_arch_layer_sizes))
        super().__init__(embedding_dim, np.asarray(list(num_embeddings_per_feature)), ln_bot, ln_top, arch_interaction_op="cat",
                         sigmoid_bot=-1, sigmoid_top=-1, loss_function="bce")
        arith = self.top_l.arith
        self.top_l = FusedMLP(*list(self.top_l.children())[:-1])
        self.top_l.arith = arith
        self.crossnet = LowRankCrossNet(F * embedding_dim, dcn_num_layers, dcn_low_rank_dim)
        self.loss_fn = FusedBCEWithLogitsLoss()

    def set_mlp_arith(self, name: str) -> None:
        super().set_mlp_arith(name)
        self.crossnet.arith = name

    def interact_features(self, x, ly):
        return self.crossnet(super().interact_features(x, ly))

    def sequential_forward(self, dense_x, lS_o, lS_i):
        from . import ops
        from .functional import CatFunction, OutSlot
        ops.check_index_errors()
        B, T, D = dense_x.size(0), len(self.emb_l), self.m_spa
        feat = torch.empty((B, (1 + T) * D), dtype=torch.float32, device=dense_x.device)
        x = self.apply_mlp(dense_x, self.bot_l, out_slot=OutSlot(feat[:, :D]))
        E = self._emb_packed(lS_o, lS_i, self.emb_l, self.v_W_l, out_slot=OutSlot(feat[:, D:]))
        z = self.crossnet(CatFunction.apply(OutSlot(feat), x, E))          # the feature buffer IS cat([dense, sparse]): no copy
        return self.apply_mlp(z, self.top_l)
