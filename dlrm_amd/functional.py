"""Autograd glue: torch.autograd.Function wrappers around the HIP kernels (dlrm_amd.ops).

The Functions exchange STRIDED VIEWS of shared buffers instead of copies:
  * the bottom tower writes its output straight into slot 0 of the [B, F*D] interaction buffer and
    the embedding kernel into slots 1..T (no torch.cat),
  * interaction backward writes per-input gradient chunks that the embedding update kernel, the
    bottom tower backward and (distributed) the reverse all-to-all consume in place.
Backward runs on the autograd engine's thread: streams/devices are looked up at call time.
"""
from __future__ import annotations

from typing import List, Optional, Sequence

import torch
from torch.autograd import Function

from . import ops
from .ops import ACT_NONE, ACT_RELU, ACT_SIGMOID


def _round4(n: int) -> int:
    return (n + 3) & ~3


def alloc2d(M: int, N: int, like: torch.Tensor, zero: bool = False) -> torch.Tensor:
    """[M, N] view of an [M, round_up(N, 4)] buffer: every row starts 16-byte aligned."""
    ldn = _round4(N)
    buf = (torch.zeros if zero else torch.empty)((M, ldn), dtype=torch.float32, device=like.device)
    return buf if ldn == N else buf[:, :N]


def _rowmajor(t: torch.Tensor) -> torch.Tensor:
    return t if (t.dim() == 2 and t.stride(1) == 1) or t.numel() == 0 else t.contiguous()


class OutSlot:
    """A caller-provided destination view handed to a Function without being an autograd input."""

    def __init__(self, view: torch.Tensor):
        self._view = view

    def get(self) -> torch.Tensor:
        return self._view[:]  # fresh tensor object sharing the storage (gets its own grad_fn)


class MLPFunction(Function):
    """nn.Sequential(Linear, act, Linear, act, ...) as one chain of fused GEMM(+bias+act) kernels.

    forward(x, acts, out_slot, W0, b0, W1, b1, ...) -> activated output of the last layer.
    Reference: DLRM_Net.create_mlp / apply_mlp (dlrm_s_pytorch.py:208-246, 399-405)."""

    @staticmethod
    def forward(ctx, x, acts, out_slot, *params):
        x = _rowmajor(x)
        L = len(acts)
        M = x.size(0)
        cur = x
        outs = []
        for i in range(L):
            W, b = params[2 * i], params[2 * i + 1]
            N = W.size(0)
            if i == L - 1 and out_slot is not None:
                y = out_slot.get()
            else:
                y = alloc2d(M, N, x)
            ops.linear_fwd(cur, W, b, acts[i], y)
            outs.append(y)
            cur = y
        ctx.acts = acts
        ctx.save_for_backward(x, *params, *outs)
        return outs[-1]

    @staticmethod
    def backward(ctx, dY):
        acts = ctx.acts
        L = len(acts)
        saved = ctx.saved_tensors
        x = saved[0]
        params = saved[1:1 + 2 * L]
        outs = saved[1 + 2 * L:]
        M = x.size(0)
        dY = _rowmajor(dY)
        grads: List[Optional[torch.Tensor]] = [None] * (2 * L)

        # last layer: activation backward + bias gradient (dY comes from outside, e.g. the loss)
        N_last = params[2 * (L - 1)].size(0)
        db = torch.zeros(N_last, dtype=torch.float32, device=x.device)
        dZ = alloc2d(M, N_last, x)
        ops.act_bwd(dY, outs[L - 1], acts[L - 1], dZ, db)
        grads[2 * (L - 1) + 1] = db
        dX = None
        for i in range(L - 1, -1, -1):
            W = params[2 * i]
            X_i = x if i == 0 else outs[i - 1]
            dW = torch.empty_like(W)
            ops.linear_bwd_weight(dZ, X_i, dW)
            grads[2 * i] = dW
            if i > 0:
                K = W.size(1)
                dprev = alloc2d(M, K, x)
                dbp = torch.zeros(K, dtype=torch.float32, device=x.device)
                # dgrad GEMM with the previous layer's activation derivative and bias-grad fused in
                ops.linear_bwd_data(dZ, W, X_i, acts[i - 1], dprev, dbp)
                grads[2 * (i - 1) + 1] = dbp
                dZ = dprev
            elif ctx.needs_input_grad[0]:
                dX = alloc2d(M, W.size(1), x)
                ops.linear_bwd_data(dZ, W, None, ACT_NONE, dX, None)
        return (dX, None, None, *grads)


class EmbeddingBagsFunction(Function):
    """All T EmbeddingBag(sum) lookups in one kernel launch; the backward pass does NOT materialise a
    gradient: it hands (bags, d_out) to `sink`, which applies the fused sparse update when the
    optimizer steps.  Reference: DLRM_Net.apply_emb (dlrm_s_pytorch.py:407-462)."""

    @staticmethod
    def forward(ctx, sink, bags, out_slot, *weights):
        T = len(weights)
        D = weights[0].size(1)
        out = out_slot.get() if out_slot is not None else alloc2d(bags.B, T * D, weights[0])
        ops.emb_fwd(weights, bags, out)
        ctx.sink = sink
        ctx.bags = bags
        ctx.weights = weights  # parameters (leaves) — kept by reference, not via save_for_backward
        return out

    @staticmethod
    def backward(ctx, dout):
        if ctx.sink is None:
            raise RuntimeError("dlrm_amd: embedding backward needs a gradient sink (fused update)")
        ctx.sink(ctx.weights, ctx.bags, _rowmajor(dout))
        return (None, None, None) + (None,) * len(ctx.weights)


class InteractFunction(Function):
    """R = [x | strictly-lower-triangular pairwise dots of the F feature vectors].

    forward(D, self_interaction, block0, block1, ...): each block is [B, k*D] and contributes k
    features (block0 = bottom-MLP output).  Reference: interact_features (dlrm_s_pytorch.py:483-504)."""

    @staticmethod
    def forward(ctx, D, self_interaction, *blocks):
        blocks = tuple(_rowmajor(b) for b in blocks)
        B = blocks[0].size(0)
        F = sum(b.size(1) // D for b in blocks)
        Wd = ops.interact_out_width(F, D, self_interaction)
        ldr = _round4(Wd)
        Rfull = torch.empty((B, ldr), dtype=torch.float32, device=blocks[0].device)
        ops.interact_fwd(blocks, D, self_interaction, Rfull)
        ctx.D, ctx.self_interaction, ctx.width = D, self_interaction, Wd
        ctx.save_for_backward(*blocks)
        return Rfull if ldr == Wd else Rfull[:, :Wd]

    @staticmethod
    def backward(ctx, dR):
        blocks = ctx.saved_tensors
        dR = _rowmajor(dR)
        B = blocks[0].size(0)
        total = sum(b.size(1) for b in blocks)
        # one flat buffer, one contiguous [B, k*D] chunk per input block, in input order: the chunks of the
        # all-to-all outputs are therefore already the packed send buffer of the reverse exchange
        flat = torch.empty(B * total, dtype=torch.float32, device=dR.device)
        dblocks, o = [], 0
        for b in blocks:
            n = B * b.size(1)
            dblocks.append(flat[o:o + n].view(B, b.size(1)))
            o += n
        ops.interact_bwd(blocks, ctx.D, ctx.self_interaction, dR, dblocks)
        return (None, None, *dblocks)


class BCELossFunction(Function):
    """BCELoss(reduction='mean'); loss and dL/dp are produced by one kernel pass."""

    @staticmethod
    def forward(ctx, p, target, weights):
        loss, dp = ops.bce_loss(p.contiguous(), target.contiguous(), weights, 1.0, want_grad=True)
        ctx.save_for_backward(dp)
        ctx.shape = p.shape
        return loss.reshape(())

    @staticmethod
    def backward(ctx, g):
        (dp,) = ctx.saved_tensors
        return dp.view(ctx.shape) * g, None, None


class MSELossFunction(Function):
    @staticmethod
    def forward(ctx, p, target):
        loss, dp = ops.mse_loss(p.contiguous(), target.contiguous(), 1.0, want_grad=True)
        ctx.save_for_backward(dp)
        ctx.shape = p.shape
        return loss.reshape(())

    @staticmethod
    def backward(ctx, g):
        (dp,) = ctx.saved_tensors
        return dp.view(ctx.shape) * g, None
