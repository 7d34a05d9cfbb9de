"""DLRM_Net for MI355X: the module surface of the reference's `DLRM_Net`
(dlrm_s_pytorch.py:207-612), every device operation a hand-written HIP kernel.

Drop-in contract:
  * constructor signature, `forward(dense_x, lS_o, lS_i)`, the public helpers `create_mlp`,
    `create_emb`, `apply_mlp`, `apply_emb`, `interact_features`, and the attributes
    `emb_l / v_W_l / bot_l / top_l / loss_fn / loss_ws / ndevices / loss_threshold /
    weighted_pooling / quantize_emb`;
  * `state_dict()` keys and shapes: `emb_l.{k}.weight`, `bot_l.{2i}.weight|bias`, `top_l.{2i}...`;
  * parameter initialisation consumes the numpy RNG in the reference's order (tables, bottom tower,
    top tower; weight then bias), so equal seeds give equal initial parameters;
  * `ext_dist.my_size > 1` selects the table-sharded / batch-split distributed forward;
  * configuration errors terminate through `sys.exit("ERROR: ...")` like the reference.
What differs by design: embedding parameters never receive a `.grad`; their sparse update is fused
into one kernel that runs when the optimizer steps (see `EmbeddingUpdateHook`), so the COO gradient
of the reference (nnz x D floats per table) is never written to HBM.
"""
from __future__ import annotations

import os
import sys
import weakref
from typing import List, Optional, Sequence

import numpy as np
import torch
import torch.nn as nn

from . import ext_dist, ops
from .functional import (BCEElementwiseFunction, BCELossFunction, CatFunction, ChunkPackFunction, ClampFunction,
                         EmbeddingBagsFunction, GatherInteractFunction, InteractFunction, MLPFunction, MSELossFunction,
                         OutSlot)
from . import functional as _functional
from .functional import MLP_CONSUMER_APPLIES_LAST_ACT, _side_stream

# one-lookup-per-bag verdict of untagged offsets: left on the device as a launch predicate (default) or waited for by the host (0)
DEVICE_PREDICATE = os.environ.get("DLRM_DEVICE_PREDICATE", "1") != "0"
from .ops import ACT_NONE, ACT_RELU, ACT_SIGMOID, BagBatch


_EMB_INIT_DEVICE = None


def set_embedding_init(device=None) -> None:
    """None (default): tables are drawn from numpy's global RNG exactly like the reference
    (dlrm_s_pytorch.py:280-284).  A torch device: tables are allocated and drawn U(-sqrt(1/n), sqrt(1/n))
    directly in that device's memory (needed for the 96 GB Criteo-Terabyte tables)."""
    global _EMB_INIT_DEVICE
    _EMB_INIT_DEVICE = device


class FusedMLP(nn.Sequential):
    """nn.Sequential of Linear / ReLU / Sigmoid children (same child names -> same state_dict keys as
    the reference tower) whose forward runs the whole tower through the fused GEMM kernels.
    Being an nn.Module it can be wrapped by DistributedDataParallel exactly like the reference's.

    `arith` ("f32" default | "bf16x6" | "bf16", see ops.arith_code) is the arithmetic of this tower's GEMMs; it is a
    property of the module and is handed to every C-ABI call — there is no process-wide switch."""

    arith = os.environ.get("DLRM_MLP_ARITH", "f32")

    def _layers(self):
        mods = list(self.children())
        params, acts = [], []
        i = 0
        while i < len(mods):
            lin = mods[i]
            if not isinstance(lin, nn.Linear):
                raise RuntimeError("FusedMLP expects Linear layers each followed by ReLU or Sigmoid")
            act = ACT_NONE
            if i + 1 < len(mods) and isinstance(mods[i + 1], (nn.ReLU, nn.Sigmoid)):
                act = ACT_RELU if isinstance(mods[i + 1], nn.ReLU) else ACT_SIGMOID
                i += 1
            params += [lin.weight, lin.bias]
            acts.append(act)
            i += 1
        return params, tuple(acts)

    def forward(self, x, out_slot: Optional[OutSlot] = None, consumer_applies_last_act: bool = False):
        """consumer_applies_last_act: the caller promises that the ONLY consumer of the output multiplies the gradient it sends back
        by the derivative of this tower's last ReLU (the interaction backward kernels do, ops.INTERACT_RELU_X) — the tower's
        backward then skips that pass.  Ignored (False) for towers that do not end in a ReLU."""
        params, acts = self._layers()
        flag = MLP_CONSUMER_APPLIES_LAST_ACT if (consumer_applies_last_act and acts and acts[-1] == ACT_RELU) else 0
        return MLPFunction.apply(x, acts, out_slot, ops.arith_code(self.arith) | flag, *params)

    def ends_in_relu(self) -> bool:
        mods = list(self.children())
        return bool(mods) and isinstance(mods[-1], nn.ReLU)


class FusedBCELoss(nn.Module):
    """BCELoss(reduction="mean") computed by the fused loss kernel."""

    def forward(self, p, target):
        return BCELossFunction.apply(p, target, None)


class FusedBCELossNone(nn.Module):
    """BCELoss(reduction="none") by the elementwise loss kernel: what `loss_fn` is under --loss-function=wbce, where the
    reference's loss_fn_wrap multiplies it by loss_ws[T] and takes the mean (dlrm_s_pytorch.py:150-156)."""

    def forward(self, p, target):
        return BCEElementwiseFunction.apply(p, target)


class FusedMSELoss(nn.Module):
    def forward(self, p, target):
        return MSELossFunction.apply(p, target)


class EmbeddingUpdateHook:
    """Applies the fused sparse embedding update when an optimizer steps.

    `run()` in the reference builds `torch.optim.SGD(dlrm.parameters())` and calls
    zero_grad / backward / step (dlrm_s_pytorch.py:1343-1369, 1611-1621).  Embedding parameters keep
    `.grad is None` (SGD skips them); backward parks (weights, bags, d_out) here and a global
    optimizer step pre-hook launches `dlrm_emb_bwd_sgd` with the learning rate the optimizer holds at
    that moment — the same point in the step, with the same lr, as the reference's sparse update."""

    _models: "weakref.WeakSet" = weakref.WeakSet()
    _handle = None
    _post = None

    @classmethod
    def register(cls, model: "DLRM_Net") -> None:
        cls._models.add(model)
        if cls._handle is None:
            from torch.optim.optimizer import register_optimizer_step_post_hook, register_optimizer_step_pre_hook
            cls._handle = register_optimizer_step_pre_hook(cls._pre_step)
            cls._post = register_optimizer_step_post_hook(cls._post_step)

    @staticmethod
    def _pre_step(optimizer, args, kwargs):
        for model in list(EmbeddingUpdateHook._models):
            if (model.overlap_streams or model.update_in_backward) and model._bound_optimizer is None and model._owned_by(optimizer):
                model._bound_optimizer = weakref.ref(optimizer)      # (what lets the NEXT backward pass launch / take the sparse update itself)
            if model._pending_emb:
                model.apply_pending_embedding_updates(optimizer)

    @staticmethod
    def _post_step(optimizer, args, kwargs):
        # overlap mode: the embedding update runs on the side stream; once optimizer.step() has returned, everything the
        # caller enqueues on ITS stream (state_dict reads, evaluation, the next forward) is ordered after it
        for model in list(EmbeddingUpdateHook._models):
            model._join_side_stream()


class _NullCtx:
    def __enter__(self):
        return self

    def __exit__(self, *exc):
        return False


class DLRM_Net(nn.Module):
    # ---------------------------------------------------------------- parameter construction
    def create_mlp(self, ln, sigmoid_layer):
        """Tower with ln[i] -> ln[i+1] Linear layers; Sigmoid after layer `sigmoid_layer`, ReLU after
        the others.  W ~ N(0, sqrt(2/(m+n))), b ~ N(0, sqrt(1/m)) drawn from numpy's global RNG
        (dlrm_s_pytorch.py:208-246)."""
        mods = []
        for i in range(ln.size - 1):
            fan_in, fan_out = int(ln[i]), int(ln[i + 1])
            lin = nn.Linear(fan_in, fan_out, bias=True)
            w = np.random.normal(0.0, np.sqrt(2 / (fan_out + fan_in)), size=(fan_out, fan_in)).astype(np.float32)
            b = np.random.normal(0.0, np.sqrt(1 / fan_out), size=fan_out).astype(np.float32)
            lin.weight.data = torch.tensor(w, requires_grad=True)
            lin.bias.data = torch.tensor(b, requires_grad=True)
            mods.append(lin)
            mods.append(nn.Sigmoid() if i == sigmoid_layer else nn.ReLU())
        return FusedMLP(*mods)

    def create_emb(self, m, ln, weighted_pooling=None):
        """One EmbeddingBag(sum) parameter holder per LOCAL table, W ~ U(-sqrt(1/n), sqrt(1/n)) from
        numpy's global RNG (dlrm_s_pytorch.py:248-294).  The holders own the parameters (state_dict
        parity); lookups go through the batched HIP kernel, not through the holders' forward."""
        tables = nn.ModuleList()
        pool_w = []
        for i in range(ln.size):
            if ext_dist.is_distributed() and i not in self.local_emb_indices:
                continue
            n = int(ln[i])
            if getattr(self, "qr_flag", False) and n > self.qr_threshold:
                sys.exit("ERROR: QR embeddings are not supported by the MI355X DLRM_Net")
            if getattr(self, "md_flag", False) and n > self.md_threshold:
                sys.exit("ERROR: mixed-dimension embeddings are not supported by the MI355X DLRM_Net")
            bound = np.sqrt(1 / n)
            if _EMB_INIT_DEVICE is None:
                w = torch.tensor(np.random.uniform(low=-bound, high=bound, size=(n, m)).astype(np.float32))
            else:
                # benchmark-scale tables (tens of GB) cannot go through a float64 numpy temporary on the
                # host: same distribution, drawn on the device (see set_embedding_init)
                w = torch.empty((n, m), dtype=torch.float32, device=_EMB_INIT_DEVICE).uniform_(-bound, bound)
            holder = nn.EmbeddingBag(n, m, mode="sum", sparse=True, _weight=w)
            pool_w.append(None if weighted_pooling is None else torch.ones(n, dtype=torch.float32))
            tables.append(holder)
        return tables, pool_w

    def __init__(self, m_spa=None, ln_emb=None, ln_bot=None, ln_top=None, arch_interaction_op=None,
                 arch_interaction_itself=False, sigmoid_bot=-1, sigmoid_top=-1, sync_dense_params=True,
                 loss_threshold=0.0, ndevices=-1, qr_flag=False, qr_operation="mult", qr_collisions=0,
                 qr_threshold=200, md_flag=False, md_threshold=200, weighted_pooling=None,
                 loss_function="bce"):
        super().__init__()
        self._pending_emb: list = []
        self.emb_update_mode = ops.UPD_SORTED
        # False (or env DLRM_FUSED_EMB_UPDATE=0): backward materialises the reference's sparse COO gradient into
        # `emb.weight.grad` (dlrm_emb_bwd_coo) and ANY torch optimizer consumes it, exactly like the reference
        self.fused_emb_update = os.environ.get("DLRM_FUSED_EMB_UPDATE", "1") != "0"
        # True (or env DLRM_OVERLAP=1): the HBM-bound embedding kernels run on a second HIP stream beside the MFMA-bound MLP
        # GEMMs they do not depend on — forward: the pooled lookups beside the bottom MLP (the reference's own overlap idea,
        # dlrm_s_pytorch.py:563-568); backward: the fused sparse update beside the bottom-MLP backward and the dense
        # optimizer step.  Same kernels, same arithmetic, same results; optimizer.step() returns with the caller's stream
        # ordered after the update.
        self.overlap_streams = os.environ.get("DLRM_OVERLAP", "0") == "1"
        # Single-process forward with ONE lookup per bag (the Criteo data sets; DLRM_FUSE_EMB_INTERACT=0 turns it off), D = 128 and at most
        # 26 tables: the interaction kernels gather the embedding rows themselves (dlrm_interact_fwd_gather / _bwd_gather) and
        # apply_emb's pooled-embedding buffer is never written or re-read — bit-identical results, forward 0.21 ms instead of 0.34 +
        # 0.26 at Criteo-Terabyte shapes (profiles/round3).  Taken when every table has exactly B lookups AND the bag starts are proven
        # to be 0, 1, 2, ... (ops.offsets_are_iota: one device pass + one synchronisation per distinct offsets tensor object, cached;
        # a ragged batch with nnz == B — an empty bag next to a two-lookup bag — takes the two kernels like every other multi-hot
        # input).  The kernels still verify the bag starts themselves and report a violation through the index-error block.
        self.fuse_emb_interact = os.environ.get("DLRM_FUSE_EMB_INTERACT", "1") == "1"
        # True (or env DLRM_UPDATE_IN_BACKWARD=1; opt-in): once the model knows its plain-SGD optimizer (from that optimizer's first step), the
        # fused backward takes the SGD step of every embedding row that ONE lookup of the batch names — the row is staged in the interaction
        # kernel's LDS and its gradient is in registers, so the table row is written at once and neither the gradient row nor a second read of
        # the table row ever touches HBM (ABI 17: dlrm_emb_presort, dlrm_interact_bwd_gather_sgd, dlrm_emb_bwd_sgd_presorted; BASELINE
        # north_star "fused sparse SGD for the backward embedding update", torchrec's apply_optimizer_in_backward in dlrm_main.py).  Rows named
        # by several lookups are updated when the optimizer steps, as before.  After backward() + optimizer.step() the tables hold the bits the
        # step-time update writes (DLRM_UPD_SORTED); what differs is WHEN: single-lookup rows change during backward(), with the learning rate the
        # optimizer holds then (the reference loop changes it after step(): dlrm_s_pytorch.py:1620-1621).  A loop that calls backward() without
        # step(), or several times per step, must leave this off — as with overlap_streams, which moves the whole update into backward.
        self.update_in_backward = os.environ.get("DLRM_UPDATE_IN_BACKWARD", "0") == "1"
        self._bound_optimizer = None        # weakref to the optimizer that owns the tables (learnt at its first step)
        self._side_keep: list = []          # tensors the side stream still reads (released at the join)
        # > 1: the pooled-embedding all-to-all of the distributed forward is pipelined in that many batch chunks (opt-in)
        self.a2a_chunks = max(int(os.environ.get("DLRM_A2A_CHUNKS", "1")), 1)
        if m_spa is None or ln_emb is None or ln_bot is None or ln_top is None or arch_interaction_op is None:
            return  # empty shell, like the reference's guard (dlrm_s_pytorch.py:320-326)

        self.ndevices = ndevices
        self.output_d = 0
        self.parallel_model_batch_size = -1
        self.parallel_model_is_not_prepared = True
        self.arch_interaction_op = arch_interaction_op
        self.arch_interaction_itself = arch_interaction_itself
        self.sync_dense_params = sync_dense_params
        self.loss_threshold = loss_threshold
        self.loss_function = loss_function
        self.weighted_pooling = ("learned" if weighted_pooling is not None and weighted_pooling != "fixed"
                                 else weighted_pooling)
        self.qr_flag = qr_flag
        if qr_flag:
            self.qr_collisions, self.qr_operation, self.qr_threshold = qr_collisions, qr_operation, qr_threshold
        self.md_flag = md_flag
        if md_flag:
            self.md_threshold = md_threshold
        self.m_spa = m_spa

        if ext_dist.is_distributed():
            n_emb = len(ln_emb)
            if n_emb < ext_dist.my_size:
                sys.exit("only (%d) sparse features for (%d) devices, table partitions will fail"
                         % (n_emb, ext_dist.my_size))
            self.n_global_emb = n_emb
            self.n_local_emb, self.n_emb_per_rank = ext_dist.get_split_lengths(n_emb)
            self.local_emb_slice = ext_dist.get_my_slice(n_emb)
            self.local_emb_indices = list(range(n_emb))[self.local_emb_slice]

        if ndevices > 1:
            sys.exit("ERROR: single-process multi-GPU (ndevices=%d) is not supported; launch one process per "
                     "GPU (torchrun) to use table-sharded embeddings over RCCL" % ndevices)
        self.emb_l, pool_w = self.create_emb(m_spa, ln_emb, weighted_pooling)
        if self.weighted_pooling == "learned":
            # dlrm_s_pytorch.py:370-375: the per-row pooling weights become parameters (state_dict keys v_W_l.{k})
            self.v_W_l = nn.ParameterList([nn.Parameter(w) for w in pool_w])
        else:
            self.v_W_l = pool_w
        self.bot_l = self.create_mlp(ln_bot, sigmoid_bot)
        self.top_l = self.create_mlp(ln_top, sigmoid_top)

        self.quantize_emb = False
        self.emb_l_q = []
        self.quantize_bits = 32

        if loss_function == "mse":
            self.loss_fn = FusedMSELoss()
        elif loss_function == "bce":
            self.loss_fn = FusedBCELoss()
        elif loss_function == "wbce":
            import __main__ as _m  # the reference reads the CLI global `args.loss_weights` (:391)
            lw = getattr(getattr(_m, "args", None), "loss_weights", "1.0-1.0")
            self.loss_ws = torch.tensor(np.fromstring(lw, dtype=float, sep="-"))
            self.loss_fn = FusedBCELossNone()
        else:
            sys.exit("ERROR: --loss-function=" + str(loss_function) + " is not supported")
        EmbeddingUpdateHook.register(self)

    def _interaction_mode(self) -> int:
        """0: strictly lower triangle, reference order (dlrm_s_pytorch.py:499-501); 1: with the diagonal (--arch-interaction-itself);
        2: torchrec's torch.triu_indices(F, F, 1) order (dlrm_amd.torchrec_variant)"""
        if getattr(self, "interaction_order", "tril") == "triu":
            if self.arch_interaction_itself:
                sys.exit("ERROR: the torchrec (triu) interaction order has no self-interaction")
            return 2
        return 1 if self.arch_interaction_itself else 0

    def set_mlp_arith(self, name: str) -> None:
        """Arithmetic of both towers' GEMMs (FusedMLP.arith): "f32" | "bf16x6" | "bf16"."""
        ops.arith_code(name)
        for tower in (self.bot_l, self.top_l):
            getattr(tower, "module", tower).arith = name        # DDP-wrapped towers keep the FusedMLP in .module

    # ---------------------------------------------------------------- operators
    def apply_mlp(self, x, layers, out_slot: Optional[OutSlot] = None, consumer_applies_last_act: bool = False):
        if consumer_applies_last_act:
            return layers(x, out_slot=out_slot, consumer_applies_last_act=True)
        if out_slot is not None:
            return layers(x, out_slot=out_slot)
        return layers(x)

    def _relu_x(self) -> int:
        """ops.INTERACT_RELU_X when the dot interaction's backward may apply the derivative of the bottom tower's last ReLU to its
        feature-0 gradient (and the tower is told to expect that: apply_mlp(..., consumer_applies_last_act=True)), else 0."""
        tower = getattr(self.bot_l, "module", self.bot_l)          # DDP / FlatDDP keep the FusedMLP in .module
        ok = _functional.FUSE_ACT_BWD and self.arch_interaction_op == "dot" and isinstance(tower, FusedMLP) and tower.ends_in_relu()
        return ops.INTERACT_RELU_X if ok else 0

    def _bags(self, lS_o, lS_i, v_W_l) -> BagBatch:
        return BagBatch(lS_o, lS_i, None)        # pooling weights, if any, are gathered on the device inside EmbeddingBagsFunction

    def _pool_weights(self, v_W_l, device) -> list:
        if v_W_l is None or not any(w is not None for w in v_W_l):
            return []
        if any(w is None for w in v_W_l):
            sys.exit("ERROR: pooling weights must be given for all tables or for none")
        out = []
        for k, w in enumerate(v_W_l):
            if w.device != device:                       # "fixed" weights are plain tensors the reference moves by hand (:1324-1326)
                w = w.to(device)
                if not isinstance(v_W_l, nn.ParameterList):
                    v_W_l[k] = w
            out.append(w)
        return out

    def _emb_weights(self, emb_l) -> List[torch.Tensor]:
        return [e.weight for e in emb_l]

    def _emb_packed(self, lS_o, lS_i, emb_l, v_W_l, out_slot: Optional[OutSlot] = None):
        """[B, T*D] pooled embeddings of all given tables, one kernel launch."""
        bags = self._bags(lS_o, lS_i, v_W_l)
        ws = self._emb_weights(emb_l)
        return EmbeddingBagsFunction.apply(self._stash_embedding_grad, bags, out_slot, *ws, *self._pool_weights(v_W_l, ws[0].device))

    def apply_emb(self, lS_o, lS_i, emb_l, v_W_l):
        """Reference-shaped result: a list with one [B, D] tensor per table (dlrm_s_pytorch.py:407-462)."""
        if self.quantize_emb:
            sys.exit("ERROR: quantized embeddings are a CPU-only inference option of the reference")
        packed = self._emb_packed(lS_o, lS_i, emb_l, v_W_l)
        D = emb_l[0].weight.size(1)
        return list(packed.split(D, dim=1))

    def weighted_bce(self, Z, T):
        """The whole wbce loss of the reference's loss_fn_wrap (mean of loss_ws[T.long()] * BCE(Z, T), dlrm_s_pytorch.py:
        150-156) in ONE fused kernel pass — for loops that call the model directly instead of through loss_fn_wrap."""
        ws = self.loss_ws.tolist()
        return BCELossFunction.apply(Z, T, None, (ws[0], ws[1] if len(ws) > 1 else ws[0]))

    def interact_features(self, x, ly):
        if self.arch_interaction_op == "dot":
            D = x.size(1)
            return InteractFunction.apply(D, self._interaction_mode(), False, x, *ly)
        if self.arch_interaction_op == "cat":
            return CatFunction.apply(None, x, *ly)
        sys.exit("ERROR: --arch-interaction-op=" + str(self.arch_interaction_op) + " is not supported")

    def quantize_embedding(self, bits):
        sys.exit("ERROR: quantized embeddings are a CPU-only inference option of the reference")

    # ---------------------------------------------------------------- fused sparse update
    def _owned_by(self, optimizer) -> bool:
        mine = {id(e.weight) for e in self.emb_l}
        return any(id(p) in mine for g in optimizer.param_groups for p in g["params"])

    def _join_side_stream(self) -> None:
        if self._side_keep:
            dev = self._side_keep[0].device
            ops.timer_mark()
            torch.cuda.current_stream(dev).wait_stream(_side_stream(dev))
            self._side_keep = []

    def _presort_for_backward(self, weights, bags, pred):
        """`bags.presort` of the fused lookup + interaction path (GatherInteractFunction.backward): ops.Presorted when this backward pass may
        take the SGD step of single-lookup rows itself (update_in_backward, a bound plain-SGD optimizer, the sorted update, nothing parked,
        not inside a distributed forward), else None."""
        if not (self.update_in_backward and self.fused_emb_update and self.emb_update_mode == ops.UPD_SORTED) or self._pending_emb:
            return None
        opt = self._bound_optimizer() if self._bound_optimizer is not None else None
        if opt is None or not _is_plain_sgd(opt) or not ops.presort_ok(weights, bags):
            return None
        plan = _embedding_update_plan(opt, weights)
        if plan is None or plan[0] != "sgd":
            return None
        return ops.emb_presort(weights, bags, plan[1], pred)

    def _stash_embedding_grad(self, weights, bags, dout, presorted=None):
        if presorted is not None:
            # the fused backward already applied the single-lookup rows with presorted.lr; the rest follows from the same sorted workspace —
            # on the side stream now (overlap mode) or when the optimizer steps
            if self.overlap_streams and self._bound_optimizer is not None and self._bound_optimizer() is not None:
                cur, side = torch.cuda.current_stream(dout.device), _side_stream(dout.device)
                if cur != side:
                    ops.timer_mark()
                    side.wait_stream(cur)
                with torch.cuda.stream(side):
                    ops.emb_bwd_sgd_presorted(weights, bags, dout, presorted)
                self._side_keep += [dout, presorted.ws, presorted.mask] + bags.keep
                return
            self._pending_emb.append((weights, bags, dout, presorted))
            return
        if not self.fused_emb_update:
            self._materialize_coo_grads(weights, bags, dout)
            return
        opt = self._bound_optimizer() if (self.overlap_streams and self._bound_optimizer is not None) else None
        if opt is not None and not self._pending_emb:
            # overlap mode, optimizer known from its earlier steps: the sparse SGD update is launched NOW — this backward
            # node runs on the side stream its forward ran on, so the update overlaps the bottom-MLP backward that autograd
            # runs next on the main stream — with the learning rate the optimizer holds at this moment (the reference reads
            # it at step(), a few microseconds later in the same loop body: dlrm_s_pytorch.py:1611-1621)
            plan = _embedding_update_plan(opt, weights) if _is_plain_sgd(opt) else None     # (the Adagrad plan advances state["step"]: only built where it is applied)
            if plan is not None and plan[0] == "sgd":
                cur, side = torch.cuda.current_stream(dout.device), _side_stream(dout.device)
                if cur != side:                      # (distributed forward: the lookups ran on the main stream)
                    ops.timer_mark()
                    side.wait_stream(cur)
                with torch.cuda.stream(side):
                    ops.emb_bwd_sgd(weights, bags, dout, plan[1], self.emb_update_mode)
                self._side_keep += [dout] + bags.keep
                return
        self._pending_emb.append((weights, bags, dout, None))
        if len(self._pending_emb) == 65:
            print("WARNING: dlrm_amd: 65 embedding gradients are parked and no optimizer that owns the tables has stepped; "
                  "every backward() keeps its [B, T*D] gradient buffer alive until then", file=sys.stderr)

    @staticmethod
    def _materialize_coo_grads(weights, bags, dout):
        """emb.weight.grad (+)= the reference's sparse COO gradient: indices = the lookup indices verbatim (uncoalesced),
        values = dout[bag(i)] * psw[i] (EmbeddingBagBackward, dlrm_s_pytorch.py:1613)."""
        if getattr(bags, "ignore_oob", False):
            sys.exit("ERROR: fused_emb_update = False (sparse COO gradients) cannot be combined with row-wise table shards, whose "
                     "out-of-range ids are other ranks' rows")
        ops.check_index_errors(sync=True)        # the reference raises in forward; never hand an out-of-range id to an optimizer's scatter
        D = weights[0].size(1)
        values = ops.emb_bwd_coo(bags, dout, D)
        for k, (w, v) in enumerate(zip(weights, values)):
            idx = ops.bag_index_tensor(bags, k).reshape(1, -1).long()
            g = torch.sparse_coo_tensor(idx, v, size=tuple(w.shape), check_invariants=False)
            w.grad = g if w.grad is None else w.grad + g

    def apply_pending_embedding_updates(self, optimizer=None, lr: Optional[float] = None) -> None:
        """Launch the fused backward+update for every stashed embedding gradient: sparse SGD for torch.optim.SGD,
        row-wise sparse Adagrad for RWSAdagrad (the reference's optim/rwsadagrad.py or dlrm_amd.optim.FusedRWSAdagrad)."""
        pending, self._pending_emb = self._pending_emb, []
        side = None
        if self.overlap_streams and pending and pending[0][2].is_cuda:
            # launched from the step pre-hook: on the side stream, beside the dense optimizer step (joined in the post-hook)
            dev = pending[0][2].device
            side = _side_stream(dev)
            ops.timer_mark()
            side.wait_stream(torch.cuda.current_stream(dev))
            self._side_keep += [p_[2] for p_ in pending] + [t for p_ in pending for t in p_[1].keep]
            self._side_keep += [t for p_ in pending if len(p_) > 3 and p_[3] is not None for t in (p_[3].ws, p_[3].mask)]
            if optimizer is not None and self._bound_optimizer is None and self._owned_by(optimizer):
                self._bound_optimizer = weakref.ref(optimizer)
        with torch.cuda.stream(side) if side is not None else _NullCtx():
            self._apply_pending(pending, optimizer, lr)

    def _apply_pending(self, pending, optimizer, lr):
        for entry in pending:
            weights, bags, dout = entry[:3]
            pre = entry[3] if len(entry) > 3 else None        # ops.Presorted: the backward pass already took the single-lookup rows
            if pre is not None:
                # (the single-lookup rows were updated in backward with pre.lr — the optimizer's lr then; the rest takes the same step size)
                ops.emb_bwd_sgd_presorted(weights, bags, dout, pre)
                continue
            if lr is not None:
                ops.emb_bwd_sgd(weights, bags, dout, lr, self.emb_update_mode)
                continue
            # parked backward passes over THESE tables (`weights` is a fresh tuple per forward call: compare the tables themselves)
            key = tuple(id(w) for w in weights)
            plan = _embedding_update_plan(optimizer, weights, count=sum(1 for p_ in pending if tuple(id(w) for w in p_[0]) == key))
            if plan is None:
                self._pending_emb.append((weights, bags, dout, None))   # another optimizer owns these tables
            elif plan[0] == "coo":
                self._materialize_coo_grads(weights, bags, dout)  # the optimizer's own step consumes .grad right after this hook
            elif plan[0] == "sgd":
                ops.emb_bwd_sgd(weights, bags, dout, plan[1], self.emb_update_mode)
            else:
                _, clr, eps, states = plan
                ops.emb_bwd_rowwise_adagrad(weights, states, bags, dout, clr, eps)

    # ---------------------------------------------------------------- forward paths
    def forward(self, dense_x, lS_o, lS_i):
        if ext_dist.is_distributed():        # more than one rank (or a forced one-rank RCCL group: ext_dist.force_distributed)
            return self.distributed_forward(dense_x, lS_o, lS_i)
        return self.sequential_forward(dense_x, lS_o, lS_i)

    def _clamp(self, p):
        if 0.0 < self.loss_threshold < 1.0:
            return ClampFunction.apply(p, float(self.loss_threshold), float(1.0 - self.loss_threshold))
        return p

    def sequential_forward(self, dense_x, lS_o, lS_i):
        """bottom MLP -> embeddings -> interaction -> top MLP (dlrm_s_pytorch.py:587-612), with the
        bottom tower and the embedding kernel writing directly into the [B, (1+T)*D] feature buffer."""
        if self.arch_interaction_op not in ("dot", "cat"):
            sys.exit("ERROR: --arch-interaction-op=" + str(self.arch_interaction_op) + " is not supported")
        ops.check_index_errors()          # host memory read, no synchronisation: bad indices of earlier steps surface here
        B = dense_x.size(0)
        T = len(self.emb_l)
        D = self.emb_l[0].weight.size(1)
        n_out = self.bot_l[-2].out_features if isinstance(self.bot_l[-2], nn.Linear) else D
        if self.arch_interaction_op == "dot" and n_out != D:
            sys.exit("ERROR: bottom MLP output (%d) and embedding dimension (%d) differ" % (n_out, D))
        # the last ReLU of the bottom tower is differentiated inside the interaction backward (which has x staged): see _relu_x
        rx = self._relu_x() if dense_x.is_cuda else 0
        if (self.fuse_emb_interact and self.arch_interaction_op == "dot" and dense_x.is_cuda and ops.gather_ok(1 + T, D)
                and not any(w is not None for w in (self.v_W_l or []))):
            bags = self._bags(lS_o, lS_i, None)
            # nnz == B does not prove one lookup per bag (an empty bag next to a two-lookup bag is legal EmbeddingBag input and the
            # reference computes it): ops.offsets_are_iota proves offsets == arange(B) on the device, once per offsets tensor
            # object (None = undecided, only while a HIP graph is being captured: GraphedTrainStep proves every incoming batch).
            if all(n == B for n in bags.nnz) and all(e.weight.data_ptr() % 16 == 0 for e in self.emb_l):
                # True / None (capturing): the fused kernels alone.  A tensor nobody vouched for: its proof is a device pass whose verdict
                # STAYS on the device (round 6; until then the host waited for it, which ended the host's run-ahead once per step) —
                # GatherInteractFunction enqueues the fused kernels and the two-kernel form, each behind the launch predicate, and the
                # host goes on.  False (this tensor object was proven ragged before): the two kernels below.
                # (DLRM_DEVICE_PREDICATE=0: the host-side proof of rounds 4-6 — started on its own stream, the bottom tower enqueued, then the
                # host waits for the verdict — kept for A/B and for the tests of that path)
                proof = None
                if DEVICE_PREDICATE:
                    state = ops.offsets_iota_state(lS_o)
                else:
                    proof = ops.offsets_are_iota_start(lS_o)
                    state = proof if (proof is None or isinstance(proof, bool)) else "pending"
                if state is not False:
                    x = self.apply_mlp(dense_x, self.bot_l, consumer_applies_last_act=bool(rx))
                    if proof is not None and state == "pending":
                        state = ops.offsets_are_iota_finish(proof)
                    if state is not False:
                        bags.iota_flag = state if isinstance(state, torch.Tensor) else None
                        bags.presort = self._presort_for_backward if self.update_in_backward else None
                        z = GatherInteractFunction.apply(self._stash_embedding_grad, D, self._interaction_mode() | rx, bags, x,
                                                         *self._emb_weights(self.emb_l))
                        return self._clamp(self.apply_mlp(z, self.top_l))
                    del x        # a ragged batch with nnz == B after all: the two kernels below (the bottom tower runs again, into its slot)
        feat = torch.empty((B, n_out + T * D), dtype=torch.float32, device=dense_x.device)
        if self.overlap_streams and dense_x.is_cuda:
            # pooled lookups (HBM-bound) on the side stream beside the bottom-MLP GEMMs (MFMA-bound): they only meet at the
            # interaction.  The side stream first waits for everything already enqueued (inputs, the previous update).
            main, side = torch.cuda.current_stream(dense_x.device), _side_stream(dense_x.device)
            ops.timer_mark()
            side.wait_stream(main)
            with torch.cuda.stream(side):
                E = self._emb_packed(lS_o, lS_i, self.emb_l, self.v_W_l, out_slot=OutSlot(feat[:, n_out:]))
            x = self.apply_mlp(dense_x, self.bot_l, out_slot=OutSlot(feat[:, :n_out]), consumer_applies_last_act=bool(rx))
            ops.timer_mark()
            main.wait_stream(side)
        else:
            x = self.apply_mlp(dense_x, self.bot_l, out_slot=OutSlot(feat[:, :n_out]), consumer_applies_last_act=bool(rx))
            E = self._emb_packed(lS_o, lS_i, self.emb_l, self.v_W_l, out_slot=OutSlot(feat[:, n_out:]))
        if self.arch_interaction_op == "cat":
            z = CatFunction.apply(OutSlot(feat), x, E)      # the feature buffer IS cat([x] + ly, 1): nothing is copied
        else:
            z = InteractFunction.apply(D, self._interaction_mode() | rx, True, x, E)   # [B, round4(width)], zero padded
        return self._clamp(self.apply_mlp(z, self.top_l))

    def distributed_forward(self, dense_x, lS_o, lS_i):
        """Table-wise sharded embeddings + batch-split MLPs (dlrm_s_pytorch.py:528-585): every rank pools
        the WHOLE batch for its tables, one all-to-all turns table-split into batch-split, the bottom
        MLP runs while the exchange is in flight."""
        batch_size = dense_x.size(0)
        if batch_size < ext_dist.my_size:
            sys.exit("ERROR: batch_size (%d) must be larger than number of ranks (%d)" % (batch_size, ext_dist.my_size))
        if batch_size % ext_dist.my_size != 0:
            sys.exit("ERROR: batch_size %d can not split across %d ranks evenly" % (batch_size, ext_dist.my_size))
        ops.check_index_errors()
        dense_x = dense_x[ext_dist.get_my_slice(batch_size)]
        lS_o = lS_o[self.local_emb_slice]
        lS_i = lS_i[self.local_emb_slice]
        if len(self.emb_l) != len(lS_o) or len(self.emb_l) != len(lS_i):
            sys.exit("ERROR: corrupted model input detected in distributed_forward call")
        D = self.emb_l[0].weight.size(1)
        E = self._emb_packed(lS_o, lS_i, self.emb_l, self.v_W_l)           # [B, T_loc*D] == packed send buffer
        C = self.a2a_chunks
        if C > 1 and self.arch_interaction_op == "dot" and (batch_size // ext_dist.my_size) % C == 0:
            return self._pipelined_exchange_forward(dense_x, E, D, batch_size, C)
        req = ext_dist.alltoall([E], self.n_emb_per_rank, emb_dim=D)
        rx = self._relu_x() if dense_x.is_cuda else 0                       # (see sequential_forward)
        x = self.apply_mlp(dense_x, self.bot_l, consumer_applies_last_act=bool(rx))     # overlaps the exchange
        ly = list(req.wait())                                               # N x [B/N, T_s*D], read in place
        if self.arch_interaction_op == "dot":
            z = InteractFunction.apply(D, self._interaction_mode() | rx, True, x, *ly)
        else:
            z = self.interact_features(x, ly)
        return self._clamp(self.apply_mlp(z, self.top_l))

    def _pipelined_exchange_forward(self, dense_x, E, D, batch_size, C):
        """Distributed forward with the all-to-all split into C batch chunks (opt-in, DLRM_A2A_CHUNKS / model.a2a_chunks).

        The exchange moves B*T_loc*D*4*(N-1)/N bytes per rank and direction over N-1 xGMI links (218 MB over ONE link at
        N = 2) while only the bottom MLP overlaps it in the reference schedule (dlrm_s_pytorch.py:563-568).  Here all C chunk
        exchanges are issued up front (they queue on RCCL's stream); interaction + top MLP of chunk c run while chunks
        c+1.. are still on the wire, and in backward the reverse exchange of chunk c overlaps the top-MLP backward of the
        chunks before it (the autograd engine reaches them in reverse order).  Weight gradients of the C top-MLP passes are
        summed by autograd; results equal the unchunked schedule up to fp32 summation order of those C partial gradients."""
        N = ext_dist.my_size
        Bc = batch_size // N // C
        sends = ChunkPackFunction.apply(E, N, C)
        reqs = [ext_dist.alltoall([sends[c]], self.n_emb_per_rank, emb_dim=D) for c in range(C)]
        rx = self._relu_x() if dense_x.is_cuda else 0                           # every chunk's backward masks its own rows of dx
        x = self.apply_mlp(dense_x, self.bot_l, consumer_applies_last_act=bool(rx))     # overlaps the first exchange
        outs = []
        for c in range(C):
            ly = list(reqs[c].wait())                                           # N x [Bc, T_s*D]
            z = InteractFunction.apply(D, self._interaction_mode() | rx, True, x[c * Bc:(c + 1) * Bc], *ly)
            outs.append(self.apply_mlp(z, self.top_l))
        return self._clamp(torch.cat(outs, dim=0))


def _is_plain_sgd(optimizer) -> bool:
    return isinstance(optimizer, torch.optim.SGD) and not _is_rwsadagrad(optimizer)


def _is_rwsadagrad(optimizer) -> bool:
    """The reference's RWSAdagrad (optim/rwsadagrad.py) or our FusedRWSAdagrad: recognised by name + hyper-parameters so
    that the reference class needs no import here."""
    d = getattr(optimizer, "defaults", {})
    return type(optimizer).__name__ in ("RWSAdagrad", "FusedRWSAdagrad") and \
        all(k in d for k in ("lr", "lr_decay", "eps", "initial_accumulator_value"))


def _embedding_update_plan(optimizer, weights, count=1):
    """None if `optimizer` does not own the tables; ("sgd", lr) for torch.optim.SGD; ("rwsadagrad", clr, eps, states)
    for RWSAdagrad — `states` are the per-table row-wise accumulators kept in optimizer.state[p]["momentum"] exactly
    where the reference keeps them (created lazily with initial_accumulator_value, rwsadagrad.py:89-95), the step count
    in state[p]["step"] (clr = lr / (1 + (step-1)*lr_decay), :113-115); ("coo",) for every other optimizer (and SGD with
    momentum / weight decay): the sparse COO gradient is materialised and the optimizer's own step consumes it, as in
    the reference.  `count` = parked backward passes of these tables (gradient accumulation)."""
    if optimizer is None:
        return None
    owner = {}
    for group in optimizer.param_groups:
        for p in group["params"]:
            owner[id(p)] = group
    groups = [owner.get(id(w)) for w in weights]
    if all(g is None for g in groups):
        return None
    if any(g is None for g in groups):
        sys.exit("ERROR: optimizer holds only some of the embedding tables")
    if _is_rwsadagrad(optimizer):
        if count > 1:
            # the reference coalesces the ACCUMULATED gradient and applies one non-linear update / one step increment;
            # several separate fused updates would not equal that
            sys.exit("ERROR: gradient accumulation (more than one backward per optimizer step) with the fused row-wise "
                     "Adagrad update is not supported; set model.fused_emb_update = False (DLRM_FUSED_EMB_UPDATE=0)")
        clrs, states = [], []
        for w, g in zip(weights, groups):
            if g.get("weight_decay", 0) != 0:
                sys.exit("ERROR: weight_decay option is not compatible with sparse gradients")
            st = optimizer.state[w]
            if "momentum" not in st or st["momentum"].device != w.device:
                st["momentum"] = torch.full([w.shape[0]], float(optimizer.defaults["initial_accumulator_value"]),
                                            dtype=torch.float32, device=w.device)
            st["step"] = st.get("step", 0) + 1
            clrs.append(float(g["lr"]) / (1.0 + (st["step"] - 1.0) * float(g["lr_decay"])))
            states.append(st["momentum"])
        if len(set(clrs)) != 1 or len({float(g["eps"]) for g in groups}) != 1:
            sys.exit("ERROR: embedding tables in param groups with different learning rates are not supported")
        clr = clrs[0]
        if float(groups[0]["lr_decay"]) == 0.0:
            clr = ops.device_lr(groups[0], clr)          # clr == lr: the group's device scalar while a whole-step graph captures (graph.py)
        return ("rwsadagrad", clr, float(groups[0]["eps"]), states)
    if not isinstance(optimizer, torch.optim.SGD):
        return ("coo",)
    for g in groups:
        if g.get("momentum", 0) != 0 or g.get("weight_decay", 0) != 0 or g.get("nesterov", False) or g.get("maximize", False):
            return ("coo",)
    lrs = [float(g["lr"]) for g in groups]
    if len(set(lrs)) != 1:
        sys.exit("ERROR: embedding tables in param groups with different learning rates are not supported")
    return ("sgd", ops.device_lr(groups[0], lrs[0]))     # (the group's device scalar while a whole-step graph captures: graph.py)
