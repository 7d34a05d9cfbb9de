"""Autograd glue: torch.autograd.Function wrappers around the HIP kernels (dlrm_amd.ops).

The Functions exchange STRIDED VIEWS of shared buffers instead of copies:
  * the bottom tower writes its output straight into slot 0 of the [B, F*D] interaction buffer and
    the embedding kernel into slots 1..T (no torch.cat),
  * interaction backward writes per-input gradient chunks that the embedding update kernel, the
    bottom tower backward and (distributed) the reverse all-to-all consume in place.
Backward runs on the autograd engine's thread: streams/devices are looked up at call time.
"""
from __future__ import annotations

import os
from typing import List, Optional, Sequence

import weakref

import torch
from torch.autograd import Function

from . import ops
from .ops import ACT_NONE, ACT_RELU, ACT_SIGMOID


# optional: weight-gradient GEMMs on a side stream, concurrent with the data-gradient chain (env DLRM_OVERLAP_WGRAD=1).
# Measured on MI355X (profiles/r01, g02): 11.83 ms/step with the overlap vs 11.30 ms without — two chip-filling GEMMs
# sharing the CUs thrash each other's L2 panels — so it is OFF by default.
OVERLAP_WGRAD = os.environ.get("DLRM_OVERLAP_WGRAD", "0") == "1"
# ... except for SMALL batches inside a HIP-graph capture: rows at or below this run weight and data gradient concurrently (0 = never)
SMALL_BATCH_OVERLAP = int(os.environ.get("DLRM_SMALL_BATCH_OVERLAP", "8192"))
SMALL_BATCH_BITS = int(os.environ.get("DLRM_SMALL_BATCH_BITS", "8192"))      # no ReLU sign bits at or below this many rows (fp32 towers)
# hidden ReLU layers store 1 sign bit per activation for the next layer's data-gradient epilogue (DLRM_RELU_BITS=0: the
# epilogue re-reads the fp32 activation instead)
RELU_BITS = os.environ.get("DLRM_RELU_BITS", "1") == "1"
# arith "bf16": activations / weights are also kept as bf16 copies and the GEMMs read those (dlrm_gemm_bf16) instead of rounding fp32
# operands inside the k-loop (DLRM_BF16_STORAGE=0: the in-loop rounding of rounds 1-2; results are bit-identical)
BF16_STORAGE = os.environ.get("DLRM_BF16_STORAGE", "1") == "1"
# bf16 storage, LEAN: hidden activations / gradients exist only as bf16 (+ ReLU sign bits) wherever every consumer reads bf16 (MLPFunction)
BF16_LEAN = os.environ.get("DLRM_BF16_LEAN", "1") == "1"
# DCN-v2 (bf16 storage): the elementwise half of a cross layer inside the epilogue of its second product (dlrm_gemm_bf16_cross); 0 = two kernels
CROSS_FUSE = os.environ.get("DLRM_CROSS_FUSE", "1") == "1"      # (1: the forward x_{l+1} is bit-identical to the two kernels, but the backward reads bf16(u) instead of the fp32 u: gradients differ at that rounding — INTEGRATION.md)
# small batches: a whole fp32 tower per launch (csrc/tower.hip) for up to DLRM_TOWER_ROWS rows (0 = never: the per-layer GEMMs) while the
# weight traffic of its 16-row workgroups stays under DLRM_TOWER_L2_MB; see _tower_applies.  Criteo-Kaggle graph: 44 -> 19 kernels per step,
# 0.367 -> 0.357 ms (profiles/round5/kaggle_towers.md)
TOWER_ROWS = int(os.environ.get("DLRM_TOWER_ROWS", "4096"))
TOWER_L2_BYTES = int(os.environ.get("DLRM_TOWER_L2_MB", "384")) << 20
# ... and which half of a tower they take: the BACKWARD launches always (data-gradient chain + grouped weight gradients: 2-3 launches instead
# of 3 per layer), the forward launch only with DLRM_TOWER_FWD=1 — the per-layer forward GEMMs spread every layer over the whole chip and are
# faster than one 128-workgroup tower launch (Criteo-Kaggle graph: 0.358 ms with the tower forward, 0.319 without, 0.362 with no tower kernels)
TOWER_FWD = os.environ.get("DLRM_TOWER_FWD", "0") == "1"
# flag bit of MLPFunction's `arith` argument (see MLPFunction.forward); DLRM_FUSE_ACT_BWD=0 makes DLRM_Net never set it (A/B)
MLP_CONSUMER_APPLIES_LAST_ACT = 0x100
FUSE_ACT_BWD = os.environ.get("DLRM_FUSE_ACT_BWD", "1") == "1"
# bf16 towers: the bf16 copies of ALL weights of a tower (W16 for the forward GEMMs, W^T16 for the data gradients) in one launch at the start of
# the forward pass (dlrm_cast_bf16_multi) instead of one launch per layer and direction; 0 = per-layer casts
MULTI_CAST = os.environ.get("DLRM_BF16_MULTI_CAST", "1") == "1"
# arith "bf16x6": activations / gradients / weights of the GEMM layers travel as three bf16 planes (split once by their producer) and the GEMMs
# are the planes form of the bf16-shaped kernel (dlrm_gemm_bf16x6), wherever its shapes hold (DLRM_BF16X6_PLANES=0: every GEMM splits its fp32
# operands inside its k-loop, the kernels of rounds 1-3; the k-contiguous products are bit-identical either way)
BF16X6_PLANES = os.environ.get("DLRM_BF16X6_PLANES", "1") == "1"
# the N == 1 head of a tower (256 -> 1 + sigmoid): its whole backward in one pass over its input (DLRM_HEAD_FUSED=0: the three calls)
HEAD_FUSED = os.environ.get("DLRM_HEAD_FUSED", "1") == "1"
_side_streams = {}


class _Bf16Store:
    """how MLPFunction keeps the reduced-width copies of arith "bf16": one bf16 matrix per tensor"""
    planes = False
    kround = staticmethod(ops.round_bf16_k)
    cast = staticmethod(ops.cast_bf16)
    cast_t = staticmethod(ops.cast_bf16_transposed)
    cast_multi = staticmethod(ops.cast_bf16_multi)           # several (copy, transposed copy) pairs in one launch
    gemm = staticmethod(ops.gemm_bf16)
    wgrad = staticmethod(ops.linear_bwd_weight_bf16)
    wgrad_ok = staticmethod(ops.linear_bwd_weight_bf16_ok)

    @staticmethod
    def empty(M, N, device):
        return torch.empty((M, N), dtype=torch.bfloat16, device=device)

    @staticmethod
    def fwd_ok(M, N, K):        # (dlrm_gemm_bf16 keeps the fp32-shaped kernel for shapes the bf16-shaped one does not take)
        return True


class _PlaneStore:
    """... and of arith "bf16x6": the three bf16 planes of the fp32 tensor, [3, rows, cols] (ops.split_bf16x3)"""
    planes = True
    kround = staticmethod(ops.round_x6_k)
    cast = staticmethod(ops.split_bf16x3)
    cast_t = staticmethod(ops.split_bf16x3_transposed)
    gemm = staticmethod(ops.gemm_bf16x6)
    wgrad = staticmethod(ops.linear_bwd_weight_bf16x6)

    @staticmethod
    def wgrad_ok(M, N, K, dZ3, X3):
        return ops.linear_bwd_weight_bf16_ok(M, N, K, dZ3[0], X3[0])

    @staticmethod
    def empty(M, N, device):
        return torch.empty((3, M, N), dtype=torch.bfloat16, device=device)

    @staticmethod
    def fwd_ok(M, N, K):        # no other kernel reads planes: exactly the preconditions of dlrm_gemm_bf16x6
        return ops.gemm_bf16x6_ok(M, N, K)


def _side_stream(device) -> "torch.cuda.Stream":
    st = _side_streams.get(device)
    if st is None:
        st = torch.cuda.Stream(device=device)
        _side_streams[device] = st
    return st


# A parameter wrapped by ext_dist.FlatDDP carries `p._dlrm_grad_arena = (weakref to the flat fp32 gradient buffer, element offset)`:
# the weight-gradient GEMMs of MLPFunction write such a parameter's gradient straight into the flat buffer that FlatDDP all-reduces
# (no bucket copy in, none out).  The slot lives ON the Parameter object (not in a table keyed by its address): a tower that is
# moved with .to(), re-created, or whose wrapper is discarded can never be redirected into a stale buffer.
ARENA_BUSY = set()      # id(parameter) of slots handed out in the running backward pass whose gradient autograd has not accumulated yet


def set_grad_arena(p: torch.Tensor, flat: Optional[torch.Tensor], offset: int = 0) -> None:
    if flat is None:
        if hasattr(p, "_dlrm_grad_arena"):
            del p._dlrm_grad_arena
        ARENA_BUSY.discard(id(p))
        return
    p._dlrm_grad_arena = (weakref.ref(flat), int(offset))


def _grad_out(p: torch.Tensor) -> torch.Tensor:
    a = getattr(p, "_dlrm_grad_arena", None)
    flat = a[0]() if a is not None else None
    if flat is None:
        return torch.empty_like(p)
    off = a[1]
    if (flat.device != p.device or flat.dtype != p.dtype or off < 0 or off + p.numel() > flat.numel()
            or (flat.data_ptr() + 4 * off) % 16 != 0):
        raise RuntimeError("dlrm_amd: the flat gradient buffer of this parameter does not match it any more (device %s vs %s, "
                           "%d + %d of %d elements); re-wrap the tower in ext_dist.FlatDDP after moving it"
                           % (flat.device, p.device, off, p.numel(), flat.numel()))
    v = flat[off:off + p.numel()].view(p.shape)
    key = id(p)
    # one writer per slot: a second use of the tower in the same backward pass (the pipelined exchange applies the top tower once per
    # chunk) or the gradient of an earlier backward pass still living there (accumulation) get a tensor of their own — autograd sums
    # them and FlatDDP moves the sum into the flat buffer
    if key in ARENA_BUSY or (p.grad is not None and p.grad.data_ptr() == v.data_ptr()):
        return torch.empty_like(p)
    ARENA_BUSY.add(key)
    return v


def _ld(t: torch.Tensor) -> int:
    return t.stride(0) if t.size(0) > 1 else max(t.stride(0), t.size(1))


def _round4(n: int) -> int:
    return (n + 3) & ~3


def alloc2d(M: int, N: int, like: torch.Tensor, zero: bool = False) -> torch.Tensor:
    """[M, N] view of an [M, round_up(N, 4)] buffer: every row starts 16-byte aligned.  A single column stays [M, 1]
    contiguous (the matrix-vector kernels take any pitch, and the loss kernel reads the predictions in place)."""
    ldn = 1 if N == 1 else _round4(N)
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


def _tower_applies(x, arith, params, L) -> bool:
    """the small-batch tower kernels take this MLP: native fp32, a batch of at most TOWER_ROWS rows, widths the kernels hold in LDS, and
    little enough weight traffic — every 16-row workgroup streams ALL weights, (M / 16) x parameters x 4 bytes through L2, which is what
    bounds the path to small batches x small towers (Criteo-Kaggle: 128 x 1.3 MB; the Criteo-Terabyte towers at 2048 rows would move 1.1 GB)"""
    if TOWER_ROWS <= 0 or not x.is_cuda or arith != ops.arith_code("f32"):
        return False
    M = x.size(0)
    K0 = params[0].size(1)
    if M < 1 or M > TOWER_ROWS or x.size(1) not in (K0, _round4(K0)) or L > ops.TOWER_MAX_LAYERS:
        return False
    widths = [x.size(1)] + [params[2 * i].size(0) for i in range(L)]
    if not ops.tower_ok(M, widths):
        return False
    if any(params[2 * i].size(1) != (K0 if i == 0 else widths[i]) for i in range(L)):
        return False
    weight_bytes = 4 * sum(widths[i] * widths[i + 1] for i in range(L))
    return ((M + 15) // 16) * weight_bytes <= TOWER_L2_BYTES


class MLPFunction(Function):
    """nn.Sequential(Linear, act, Linear, act, ...) as one chain of fused GEMM(+bias+act) kernels.

    forward(x, acts, out_slot, arith, W0, b0, W1, b1, ...) -> activated output of the last layer (`arith`: the MLP
    arithmetic of every GEMM of the tower, forward and backward — a per-call argument of the C ABI).
    Reference: DLRM_Net.create_mlp / apply_mlp (dlrm_s_pytorch.py:208-246, 399-405).

    Input widths that are not a multiple of 4 floats (13 dense features, 479 interaction outputs) would
    force 4-byte loads in the first GEMM: the first layer then runs on a zero-padded copy of its weight
    [N, round4(K)] and on a zero-padded input.  `x` may already BE that padded input (width round4(K) with
    zero padding columns, which is what InteractFunction(padded=True) emits); otherwise a padded copy of
    `x` is made (cheap: it only happens for the narrow dense-feature input)."""

    @staticmethod
    def forward(ctx, x, acts, out_slot, arith, *params):
        x = _rowmajor(x)
        # MLP_CONSUMER_APPLIES_LAST_ACT OR-ed into `arith`: the gradient this tower receives is already multiplied by the derivative of its
        # LAST activation (a ReLU) — the interaction backward does that to its feature-0 gradient on request (ops.INTERACT_RELU_X), so
        # backward starts from dY as it comes instead of with an act_bwd pass over [M, N_last]
        ctx.consumer_applies_last_act = bool(int(arith) & MLP_CONSUMER_APPLIES_LAST_ACT)
        arith = int(arith) & ~MLP_CONSUMER_APPLIES_LAST_ACT
        if ctx.consumer_applies_last_act and acts[-1] != ACT_RELU:
            raise RuntimeError("dlrm_amd: MLP_CONSUMER_APPLIES_LAST_ACT needs a tower that ends in a ReLU")
        ctx.arith = arith
        L = len(acts)
        M = x.size(0)
        W0 = params[0]
        K0, Kp = W0.size(1), _round4(W0.size(1))
        ctx.in_width = x.size(1)
        ctx.tower = False
        if _tower_applies(x, arith, params, L):
            return MLPFunction._tower_forward(ctx, x, acts, out_slot, params)
        W0p = None
        if Kp != K0:
            if x.size(1) != Kp:
                if x.size(1) != K0:
                    raise RuntimeError("dlrm_amd: MLP input width %d does not match the first layer (%d)" % (x.size(1), K0))
                x = ops.pad_cols(x, Kp)
            W0p = ops.pad_cols(W0, Kp)
        elif x.size(1) != K0:
            raise RuntimeError("dlrm_amd: MLP input width %d does not match the first layer (%d)" % (x.size(1), K0))
        cur = x
        need_grad = any(ctx.needs_input_grad)
        # sign bits: a forward that will be differentiated (Function.forward itself runs grad-free), and not a SMALL batch — there the GEMMs run
        # on 64 x 64 tiles that neither write nor read the bits (the bits would come from a stand-alone kernel: one more launch per layer of a
        # launch-bound step) and the data gradient reads the fp32 activation, which small batches keep in cache anyway
        need_bits = RELU_BITS and need_grad and (M > SMALL_BATCH_BITS or (BF16_STORAGE and arith == ops.arith_code("bf16")) or
                                                 (BF16X6_PLANES and arith == ops.arith_code("bf16x6")))
        # arith "bf16" with bf16 STORAGE (default; DLRM_BF16_STORAGE=0 restores the in-loop rounding of rounds 1-2): every GEMM layer reads a
        # bf16 copy of its input and of its weight (dlrm_gemm_bf16: no conversion in the k-loop).  LEAN (default, DLRM_BF16_LEAN=0 turns it
        # off): a hidden activation is written ONLY as bf16 + ReLU sign bits when every consumer reads those — the next layer's forward
        # GEMM, its weight gradient (dlrm_linear_bwd_weight_bf16 reads bf16 operands k-strided) and the data-gradient mask (sign bits); the
        # fp32 copy (4 of the 6 bytes written per element) exists only where something reads it: the tower's output, the input of a
        # matrix-vector layer, layers whose shapes the bf16 weight gradient does not take.  Same operand rounding and accumulation
        # order as the in-loop path for every product.
        # arith "bf16x6" with PLANES (default): the same plan with every bf16 copy replaced by the three planes of the fp32 tensor (always lean)
        st = _Bf16Store
        store16 = BF16_STORAGE and arith == ops.arith_code("bf16")
        lean = store16 and BF16_LEAN
        if BF16X6_PLANES and arith == ops.arith_code("bf16x6") and M >= 256 and RELU_BITS and ops.BF16_PHASED:
            st, store16, lean = _PlaneStore, True, True
        ctx.st = st
        widths = [params[2 * i].size(0) for i in range(L)]                                   # N_i
        kin = [W0p.size(1) if (i == 0 and W0p is not None) else params[2 * i].size(1) for i in range(L)]
        out_last = out_slot.get() if out_slot is not None else None
        kb = [st.kround(kin[i]) for i in range(L)]
        use16 = [store16 and widths[i] % 4 == 0 and widths[i] != 1 and st.fwd_ok(M, widths[i], kb[i]) for i in range(L)]
        if out_last is not None and use16[L - 1] and not (_ld(out_last) % 4 == 0 and out_last.data_ptr() % 16 == 0):
            use16[L - 1] = False
        # weight gradient of layer i from bf16 operands as stored (X16_i = the bf16 input of its forward GEMM, dZ16_i)
        wg16 = [lean and need_grad and use16[i] and M >= 256 and M % 64 == 0 and widths[i] % 8 == 0 and widths[i] >= 64 and kin[i] >= 64
                and ops.BF16_PHASED for i in range(L)]
        # data gradient of layer i (dX = dZ . W) on bf16 operands: reduction over widths[i]
        # (its epilogue applies the derivative of the activation BELOW it: none, or ReLU through that layer's sign bits)
        dg16 = [store16 and use16[i] and widths[i] % 32 == 0 and kin[i] % 4 == 0 and st.fwd_ok(M, kin[i], widths[i]) and
                (i == 0 or acts[i - 1] == ACT_NONE or (acts[i - 1] == ACT_RELU and need_bits)) for i in range(L)]
        need32 = []
        for i in range(L):
            n32 = (not lean) or i == L - 1 or not use16[i] or not use16[i + 1] or kb[i + 1] != widths[i]
            if not n32 and need_grad:
                # backward consumers of the fp32 activation: the next layer's weight gradient when it is not the bf16 one; the ReLU
                # derivative when there are no sign bits; any other activation's derivative
                # ... and the next layer's DATA gradient when it is not the bf16 one (widths the bf16 / plane GEMM refuses, e.g. 200 or 72):
                # the fp32-storage fallback masks with this activation (ops.linear_bwd_data), the sign bits alone would not do
                n32 = ((not wg16[i + 1]) or (not dg16[i + 1]) or (acts[i] == ACT_RELU and not need_bits)
                       or acts[i] not in (ACT_RELU, ACT_NONE))
            need32.append(n32)
        # every weight copy of the tower in ONE launch (bf16 storage; the planes form keeps its per-layer splits): W16 [N, kb] for the forward
        # GEMM of a use16 layer, W^T16 [K, N] for its data gradient where backward will take the bf16 path (weights do not change in between)
        pre16, preT16 = [None] * L, [None] * L
        if store16 and st is _Bf16Store and MULTI_CAST:
            items, where = [], []
            for i in range(L):
                W_i = W0p if (i == 0 and W0p is not None) else params[2 * i]
                want = kb[i] if use16[i] else None
                wantT = widths[i] if (need_grad and dg16[i] and (i > 0 or ctx.needs_input_grad[0])) else None
                if want or wantT:
                    items.append((W_i, want, wantT)); where.append(i)
            if len(items) > 1:
                for i, (d_, dT_) in zip(where, st.cast_multi(items, category="linear_fwd")):
                    pre16[i], preT16[i] = d_, dT_
        ctx.preT16 = preT16
        cur16 = None                                                  # bf16 copy of the current activation, [M, kb of the next layer]
        outs, in16, bits = [], [], []                                   # in16[i]: the reduced-width input of layer i's GEMM, kept for its weight gradient
        for i in range(L):
            W, b = params[2 * i], params[2 * i + 1]
            if i == 0 and W0p is not None:
                W = W0p
            N = widths[i]
            y = (out_last if (i == L - 1 and out_last is not None) else alloc2d(M, N, x)) if need32[i] else None
            # hidden ReLU layers also store their sign bits (1 bit per element): the data-gradient GEMM of the NEXT layer reads
            # those instead of this fp32 activation for its fused ReLU derivative
            rb = ops.relu_bits_alloc(M, N, x.device) if (i < L - 1 and acts[i] == ACT_RELU and need_bits) else None
            y16 = None
            if use16[i]:
                a16 = cur16 if (cur16 is not None and cur16.size(-1) == kb[i]) else st.cast(cur, kb[i], category="linear_fwd")
                in16.append(a16 if (lean and need_grad and wg16[i]) else None)
                w16 = pre16[i] if pre16[i] is not None else st.cast(W, kb[i], category="linear_fwd")
                # the bf16 copy of this activation, if the NEXT layer is a GEMM that can take it as it is
                nxt = i + 1 < L and use16[i + 1] and kb[i + 1] == N
                y16 = st.empty(M, N, x.device) if (nxt or not need32[i]) else None
                st.gemm(a16, w16, b, acts[i], y, y16, relu_bits_out=rb, category="linear_fwd")
            else:
                in16.append(None)
                ops.linear_fwd(cur, W, b, acts[i], y, arith, relu_bits=rb)
            cur16 = y16
            outs.append(y)
            bits.append(rb)
            cur = y
        ctx.bits = bits
        ctx.acts = acts
        ctx.padded = W0p is not None
        ctx.plan = (use16, wg16, dg16, need32, kb) if store16 else None
        ctx.in16 = in16
        ctx.have32 = [o is not None for o in outs]
        ctx.save_for_backward(x, *params, *[o for o in outs if o is not None], *([W0p] if W0p is not None else []))
        return outs[-1]

    @staticmethod
    def _tower_forward(ctx, x, acts, out_slot, params):
        """small batches, native fp32: this tower's BACKWARD pass will run on the whole-tower kernels of csrc/tower.hip (one launch for the
        data-gradient chain, one + one for all weight / bias gradients), so every layer's fp32 output is kept.  The forward pass is the
        per-layer GEMMs by default, or (TOWER_FWD) one tower launch — activations of 16 batch rows stay in LDS from layer to layer, nothing
        is padded: an input that arrives unpadded (13 dense features) is read with element loads; an input that arrives padded to a
        multiple of 4 (the interaction's 367 -> 368 columns) meets a zero-padded copy of the first weight, as in the per-layer path."""
        L = len(acts)
        M = x.size(0)
        W0 = params[0]
        K0 = W0.size(1)
        W0p = ops.pad_cols(W0, x.size(1)) if x.size(1) != K0 else None
        outs = [(out_slot.get() if (i == L - 1 and out_slot is not None) else alloc2d(M, params[2 * i].size(0), x)) for i in range(L)]
        if TOWER_FWD:
            Ws = [W0p if (i == 0 and W0p is not None) else params[2 * i] for i in range(L)]
            ops.tower_fwd(x, Ws, [params[2 * i + 1] for i in range(L)], acts, outs)
        else:
            # default: the forward pass on the per-layer GEMMs (which spread a layer over the whole chip), the backward pass on the tower
            # kernels (which only need every layer's fp32 output); unaligned inputs are padded as the per-layer path does
            if x.size(1) % 4 != 0:
                Kp = _round4(x.size(1))
                x, W0p = ops.pad_cols(x, Kp), ops.pad_cols(W0, Kp)
            cur = x
            for i in range(L):
                ops.linear_fwd(cur, W0p if (i == 0 and W0p is not None) else params[2 * i], params[2 * i + 1], acts[i], outs[i],
                               ops.arith_code("f32"), relu_bits=None)
                cur = outs[i]
        ctx.tower, ctx.acts, ctx.padded = True, acts, W0p is not None
        ctx.save_for_backward(x, *params, *outs, *([W0p] if W0p is not None else []))
        return outs[-1]

    @staticmethod
    def _tower_backward(ctx, dY):
        acts = ctx.acts
        L = len(acts)
        saved = ctx.saved_tensors
        x, params, outs = saved[0], saved[1:1 + 2 * L], list(saved[1 + 2 * L:1 + 3 * L])
        W0p = saved[1 + 3 * L] if ctx.padded else None
        M = x.size(0)
        dY = _rowmajor(dY)
        Ws = [W0p if (i == 0 and W0p is not None) else params[2 * i] for i in range(L)]
        dZs = [alloc2d(M, params[2 * i].size(0), x) for i in range(L)]
        dX = alloc2d(M, x.size(1), x) if ctx.needs_input_grad[0] else None
        ops.tower_bwd(dY, Ws, acts, outs, dZs, dX, last_act_applied=ctx.consumer_applies_last_act)
        dWs = [_grad_out(params[2 * i]) for i in range(L)]              # (the first layer's at the parameter's true width)
        dbs = [_grad_out(params[2 * i + 1]) for i in range(L)]
        ops.tower_wgrad(dZs, [x] + outs[:-1], dWs, dbs)
        grads = []
        for i in range(L):
            grads += [dWs[i], dbs[i]]
        if dX is not None and dX.size(1) != ctx.in_width:
            dX = dX[:, :ctx.in_width]
        return (dX, None, None, None, *grads)

    @staticmethod
    def backward(ctx, dY):
        if ctx.tower:
            return MLPFunction._tower_backward(ctx, dY)
        acts, arith = ctx.acts, ctx.arith
        L = len(acts)
        saved = ctx.saved_tensors
        x = saved[0]
        params = saved[1:1 + 2 * L]
        n32 = sum(ctx.have32)
        it32 = iter(saved[1 + 2 * L:1 + 2 * L + n32])
        outs = [next(it32) if h else None for h in ctx.have32]          # fp32 activations (None: the layer kept bf16 + sign bits only)
        W0p = saved[1 + 2 * L + n32] if ctx.padded else None
        M = x.size(0)
        dY = _rowmajor(dY)
        grads: List[Optional[torch.Tensor]] = [None] * (2 * L)
        store16 = ctx.plan is not None
        st = ctx.st
        use16, wg16, dg16, need32, kb = ctx.plan if store16 else ([False] * L,) * 4 + ([0] * L,)

        # last layer: activation backward (its dY comes from outside, e.g. the loss or the interaction)
        N_last = params[2 * (L - 1)].size(0)
        first = L - 1                                      # the layer the loop below starts with
        dZ = None
        if (HEAD_FUSED and N_last == 1 and L >= 2 and not store16 and not ctx.consumer_applies_last_act and arith == ops.arith_code("f32")
                and outs[L - 1] is not None and outs[L - 2] is not None and not OVERLAP_WGRAD
                and not (0 < SMALL_BATCH_OVERLAP and M <= SMALL_BATCH_OVERLAP and torch.cuda.is_current_stream_capturing())):
            # the 256 -> 1 head of the top tower: act_bwd + weight gradient + data gradient of an N == 1 layer are ONE pass over its input
            # (dlrm_linear_head_bwd: X read once instead of twice, one launch + the finish instead of four: 48 -> ~29 us at B = 65536; the
            # bits of the three calls).  Outside its fast path nothing is launched and the three calls below run.
            dW_h, db_h = _grad_out(params[2 * (L - 1)]), _grad_out(params[2 * (L - 1) + 1])
            dprev = alloc2d(M, params[2 * (L - 1)].size(1), x)
            if ops.linear_head_bwd(dY, outs[L - 1], acts[L - 1], outs[L - 2], params[2 * (L - 1)], acts[L - 2], dprev, dW_h, db_h):
                grads[2 * (L - 1)], grads[2 * (L - 1) + 1] = dW_h, db_h
                dZ, first = dprev, L - 2
        if dZ is not None:
            pass
        elif ctx.consumer_applies_last_act and _ld(dY) % 4 == 0 and dY.data_ptr() % 16 == 0:
            dZ = dY                                        # already dL/dz of the last layer (see forward)
        else:
            dZ = alloc2d(M, N_last, x)
            if ctx.consumer_applies_last_act:
                dZ.copy_(dY)                               # (an unaligned gradient view: the GEMMs want 16-byte rows)
            else:
                ops.act_bwd(dY, outs[L - 1], acts[L - 1], dZ, None)
        dX = None
        # The weight-gradient GEMM of layer i and the data-gradient GEMM that feeds layer i-1 both consume dZ_i and
        # are independent: wgrad goes to a side HIP stream so the two kernels share the chip (their epilogue
        # store bursts and tail rounds interleave with the other's MFMA phases instead of idling the matrix cores).
        main = torch.cuda.current_stream()
        # (small batches — Criteo-Kaggle's 2048 rows — are the opposite regime: every GEMM of the step occupies a fraction of the chip, so the
        # independent weight-gradient launch runs BESIDE the data-gradient launch for free; the fork / join are events a graph capture records)
        # — measured at Criteo-Kaggle shapes: graphed step 0.515 -> 0.497 ms; in the EAGER step the two extra event records per layer cost the
        # host more than the overlap saves (1.51 -> 1.74 ms), so only a capturing stream forks
        side = _side_stream(x.device) if (OVERLAP_WGRAD or (0 < SMALL_BATCH_OVERLAP and M <= SMALL_BATCH_OVERLAP
                                                          and torch.cuda.is_current_stream_capturing())) else None
        keep = []                                          # tensors the side stream reads stay alive until the join
        dZ16 = None                                        # bf16 copy of dZ when the previous data-gradient GEMM produced one
        for i in range(first, -1, -1):
            W = params[2 * i] if not (i == 0 and W0p is not None) else W0p
            X_i = x if i == 0 else outs[i - 1]
            X16_i = ctx.in16[i] if wg16[i] else None
            # first layer on a zero-padded input: the gradient is written at the parameter's true width (the padding
            # columns of X are dropped inside the kernel) — contiguous, no slicing / re-packing afterwards
            dW = _grad_out(params[2 * i])
            db = _grad_out(params[2 * i + 1])
            N_i, K_i = W.size(0), W.size(1)
            do16 = wg16[i] and X16_i is not None
            if do16 and not st.wgrad_ok(M, N_i, dW.size(1), dZ16 if dZ16 is not None else X16_i, X16_i):
                # forward already dropped the fp32 operands on the strength of this plan: a run-time disagreement (DLRM_BF16_PHASED changed
                # between forward and backward, an unaligned gradient slot) must not fall into the fp32 branch with operands that no longer exist
                raise RuntimeError("dlrm_amd: the bf16 weight gradient planned at forward time for layer %d (M=%d, N=%d, K=%d) is refused at "
                                   "backward time (environment switched between forward and backward, or unaligned gradient storage)"
                                   % (i, M, N_i, dW.size(1)))
            if do16 and dZ16 is None:
                dZ16 = st.cast(dZ, N_i, category="linear_bwd_weight")                # (the tower's last layer: its dZ comes from act_bwd in fp32)

            def wgrad():
                if do16:
                    st.wgrad(dZ16, X16_i, dW, db)                                 # bf16 operands (or planes) as stored, k-strided reads
                else:
                    ops.linear_bwd_weight(dZ, X_i, dW, db, arith=arith)           # dW and db (row sums of dZ^T) in one GEMM
            if side is not None:
                side.wait_event(main.record_event())
                with torch.cuda.stream(side):
                    wgrad()
                keep += [dZ, dZ16]
            else:
                wgrad()
            grads[2 * i], grads[2 * i + 1] = dW, db
            need_dx = i > 0 or ctx.needs_input_grad[0]
            if not need_dx:
                continue
            mask_act = acts[i - 1] if i > 0 else ACT_NONE
            rbits = ctx.bits[i - 1] if i > 0 else None
            # what the layer below reads of this gradient: its bf16 weight / data gradient take the bf16 copy; fp32 is written only when
            # something reads it (the tower's input gradient, an fp32-storage weight / data gradient below)
            want16 = i > 0 and store16 and (wg16[i - 1] or (dg16[i - 1] and (i - 1 > 0 or ctx.needs_input_grad[0]))) and K_i % 32 == 0
            want32 = i == 0 or not store16 or not (wg16[i - 1] and (dg16[i - 1] or not (i - 1 > 0 or ctx.needs_input_grad[0]))) or not want16
            ok16 = (store16 and dg16[i] and (mask_act == ACT_NONE or (mask_act == ACT_RELU and rbits is not None)))
            dprev = alloc2d(M, K_i, x) if (want32 or not ok16) else None
            if ok16 and dprev is not None and _ld(dprev) % 4 != 0:
                ok16 = False
            if ok16:
                # bf16 storage: dX = dZ . W as a <k-contiguous, k-contiguous> GEMM over a transposed bf16 copy of W
                a16 = dZ16 if dZ16 is not None else st.cast(dZ, N_i, category="linear_bwd_data")
                wT16 = ctx.preT16[i] if ctx.preT16[i] is not None else st.cast_t(W, N_i, category="linear_bwd_data")
                d16 = st.empty(M, K_i, x.device) if want16 else None
                st.gemm(a16, wT16, None, ACT_NONE, dprev, d16, relu_bits_in=rbits, category="linear_bwd_data")
                dZ16 = d16
            else:
                # dgrad GEMM with the previous layer's activation derivative fused into the epilogue
                if dZ is None:
                    raise RuntimeError("dlrm_amd: internal: fp32 gradient missing for an fp32-storage data gradient")
                ops.linear_bwd_data(dZ, W, X_i if i > 0 else None, mask_act, dprev, arith, relu_bits=rbits)
                dZ16 = None
            if i > 0:
                dZ = dprev
            else:
                dX = dprev
                if dX.size(1) != ctx.in_width:
                    dX = dX[:, :ctx.in_width]
        if side is not None:
            main.wait_stream(side)
            del keep
        return (dX, None, None, None, *grads)


class EmbeddingBagsFunction(Function):
    """All T EmbeddingBag(sum) lookups in one kernel launch; the backward pass does NOT materialise a
    gradient: it hands (bags, d_out) to `sink`, which applies the fused sparse update when the
    optimizer steps.  Reference: DLRM_Net.apply_emb (dlrm_s_pytorch.py:407-462)."""

    @staticmethod
    def forward(ctx, sink, bags, out_slot, *tensors):
        # tensors = the T tables, optionally followed by T pooling-weight vectors [rows_t] (--weighted-pooling: `v_W_l`,
        # dlrm_s_pytorch.py:289-293,370-375): psw = v_W[idx] is gathered here and, when the vectors are Parameters
        # ("learned"), their dense gradient comes back from backward
        T = bags.T
        weights, vws = tensors[:T], tensors[T:]
        D = weights[0].size(1)
        if vws:
            ops.pool_weights_gather(vws, bags)
        out = out_slot.get() if out_slot is not None else alloc2d(bags.B, T * D, weights[0])
        ops.emb_fwd(weights, bags, out)
        ctx.sink = sink
        ctx.bags = bags
        ctx.weights = weights  # parameters (leaves) — kept by reference, not via save_for_backward
        ctx.vws = vws
        return out

    @staticmethod
    def backward(ctx, dout):
        if ctx.sink is None:
            raise RuntimeError("dlrm_amd: embedding backward needs a gradient sink (fused update)")
        dout = _rowmajor(dout)
        dvw = (None,) * len(ctx.vws)
        if ctx.vws and any(ctx.needs_input_grad[3 + len(ctx.weights):]):
            dvw = tuple(ops.emb_psw_grad(ctx.weights, ctx.bags, dout, ctx.vws))    # before the update touches the tables
        ctx.sink(ctx.weights, ctx.bags, dout)
        return (None, None, None) + (None,) * len(ctx.weights) + dvw


class InteractFunction(Function):
    """R = [x | strictly-lower-triangular pairwise dots of the F feature vectors].

    forward(D, self_interaction, padded, block0, block1, ...): each block is [B, k*D] and contributes k
    features (block0 = bottom-MLP output).  Reference: interact_features (dlrm_s_pytorch.py:483-504)."""

    @staticmethod
    def forward(ctx, D, self_interaction, padded, *blocks):
        order = None
        if blocks and isinstance(blocks[0], (list, tuple)):          # optional leading feature permutation (see ops.interact_fwd)
            order, blocks = list(blocks[0]), blocks[1:]
        blocks = tuple(_rowmajor(b) for b in blocks)
        B = blocks[0].size(0)
        F = sum(b.size(1) // D for b in blocks)
        # `self_interaction | ops.INTERACT_RELU_X` (backward only): block 0's first feature is a ReLU output whose derivative the backward
        # kernel applies (the producing MLPFunction was told so: MLP_CONSUMER_APPLIES_LAST_ACT) — never together with a permutation
        relu_x = int(self_interaction) & ops.INTERACT_RELU_X
        if relu_x and order is not None:
            raise RuntimeError("dlrm_amd: INTERACT_RELU_X names feature 0 of the block list; it cannot be combined with a feature permutation")
        mode = int(self_interaction) & 3
        Wd = ops.interact_out_width(F, D, mode)
        ldr = _round4(Wd)
        Rfull = torch.empty((B, ldr), dtype=torch.float32, device=blocks[0].device)
        ops.interact_fwd(blocks, D, mode, Rfull, order=order)
        ctx.order = order
        ctx.D, ctx.self_interaction, ctx.width = D, mode | relu_x, Wd
        ctx.save_for_backward(*blocks)
        # padded=True hands out the whole [B, round4(width)] buffer (zero padding columns) for MLPFunction
        return Rfull if (ldr == Wd or padded) else Rfull[:, :Wd]

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
        ops.interact_bwd(blocks, ctx.D, ctx.self_interaction, dR, dblocks, order=ctx.order)
        return (None, None, None, *dblocks) if ctx.order is None else (None, None, None, None, *dblocks)


class GatherInteractFunction(Function):
    """apply_emb + interact_features fused for one-lookup-per-bag batches (dlrm_s_pytorch.py:407-462 + 483-504):
    R = [x | lower-triangular dots of (x, E_1[idx_1], ..., E_T[idx_T])], the embedding rows fetched by the interaction kernel
    itself — the [B, T*D] pooled-embedding buffer is neither written nor read back.  Backward recomputes nothing: it gathers
    the same rows again, writes dx and the [B, T*D] gradient of the gathered rows, and hands the latter to `sink` (the fused
    sparse update), exactly like EmbeddingBagsFunction.backward."""

    @staticmethod
    def forward(ctx, sink, D, self_interaction, bags, x, *weights):
        # `bags.iota_flag` (a device int32, ops.offsets_iota_state): whether the batch really has ONE lookup per bag is known on the device
        # only — nnz == B does not prove it.  Both implementations are enqueued with that launch predicate (ABI 16): the fused kernel runs if
        # the flag is zero; dlrm_emb_fwd + the plain interaction (the reference's apply_emb + interact_features as two kernels, through a pooled
        # buffer that then has to live until backward) run if it is not.  The launches that do not run return at once; the host never waits.
        x = _rowmajor(x)
        F = 1 + len(weights)
        mode = int(self_interaction) & 3                      # (| ops.INTERACT_RELU_X: see InteractFunction.forward)
        Wd = ops.interact_out_width(F, D, mode)
        R = torch.empty((x.size(0), _round4(Wd)), dtype=torch.float32, device=x.device)
        flag = getattr(bags, "iota_flag", None)
        ly = None
        if flag is None:
            ops.interact_fwd_gather(x, weights, bags, D, mode, R)
        else:
            ops.interact_fwd_gather(x, weights, bags, D, mode, R, pred=(flag, 0))
            ly = alloc2d(x.size(0), len(weights) * D, x)
            ops.emb_fwd(weights, bags, ly, pred=(flag, 1))
            ops.interact_fwd((x, ly), D, mode, R, pred=(flag, 1))
        ctx.sink, ctx.bags, ctx.weights = sink, bags, weights
        ctx.D, ctx.self_interaction = D, int(self_interaction) & (3 | ops.INTERACT_RELU_X)
        ctx.flag = flag
        if ly is None:
            ctx.save_for_backward(x)
        else:
            ctx.save_for_backward(x, ly)
        return R                                   # [B, round4(width)], zero padding columns (what MLPFunction takes as is)

    @staticmethod
    def backward(ctx, dR):
        x = ctx.saved_tensors[0]
        dR = _rowmajor(dR)
        B, D, T = x.size(0), ctx.D, len(ctx.weights)
        flat = torch.empty(B * (1 + T) * D, dtype=torch.float32, device=dR.device)
        dx, dE = flat[:B * D].view(B, D), flat[B * D:].view(B, T * D)
        if ctx.sink is None:
            raise RuntimeError("dlrm_amd: embedding backward needs a gradient sink (fused update)")
        # `bags.presort` (DLRM_Net._presort_for_backward; None unless the model was told to update in backward and knows its SGD optimizer): the
        # sort of the sparse update runs NOW and the fused kernel takes the SGD step of every row a single lookup of the batch names (its table
        # row is staged in LDS anyway) — ops.Presorted travels to the sink, which applies the rest from the same sorted workspace
        pred = None if ctx.flag is None else (ctx.flag, 0)
        presort = getattr(ctx.bags, "presort", None)
        pre = presort(ctx.weights, ctx.bags, pred) if presort is not None else None
        ops.interact_bwd_gather(x, ctx.weights, ctx.bags, D, ctx.self_interaction, dR, dx, dE, pred=pred, presorted=pre)
        if ctx.flag is not None:
            ops.interact_bwd((x, ctx.saved_tensors[1]), D, ctx.self_interaction, dR, (dx, dE), pred=(ctx.flag, 1))
        if pre is None:
            ctx.sink(ctx.weights, ctx.bags, dE)
        else:
            ctx.sink(ctx.weights, ctx.bags, dE, presorted=pre)
        return (None, None, None, None, dx) + (None,) * T


class LowRankCrossNetFunction(Function):
    """DCN-v2 interaction (torchrec.modules.crossnet.LowRankCrossNet, the MLPerf-v2 default: torchrec_dlrm/dlrm_main.py:608-619):
        x_{l+1} = x_0 * (W_l (V_l x_l) + b_l) + x_l,   l = 0 .. L-1,   on the flattened feature buffer x_0 [B, F*D].
    forward(arith, x0, V_0, W_0, b_0, V_1, ...): V_l [r, in], W_l [in, r], b_l [in].  Both products run on the MLP GEMM kernels
    (act none), the elementwise halves on dlrm_cross_fwd / _bwd; backward is written out by hand so that the three gradient
    streams into x_0 (through every layer's Hadamard factor, and through x_l of layer 0) are accumulated by our kernels instead
    of autograd's ATen adds."""

    @staticmethod
    def _bf16_ok(arith, M, n_in, r) -> bool:
        """bf16 STORAGE form (arith "bf16"): every product reads bf16 operands as stored — x_l and du as the bf16 copies the elementwise
        kernels write beside (or instead of) their fp32 results, v and dv as bf16-only GEMM outputs; weight gradients from the stored
        operands (dlrm_linear_bwd_weight_bf16); the gradient sum g + dv.V inside the GEMM epilogue (addend).  Needs the shapes of the
        bf16-shaped kernels."""
        return (BF16_STORAGE and BF16_LEAN and arith == ops.arith_code("bf16") and M >= 256 and M % 64 == 0 and n_in % 64 == 0 and r % 64 == 0
                and n_in >= 192 and r >= 192 and ops.BF16_PHASED)

    @staticmethod
    def forward(ctx, arith, x0, *params):
        x0 = x0.contiguous()
        L = len(params) // 3
        M, n_in = x0.shape
        ctx.bf16 = LowRankCrossNetFunction._bf16_ok(arith, M, n_in, params[0].size(0))
        if ctx.bf16:
            xl, xl16 = x0, ops.cast_bf16(x0, n_in, category="linear_fwd")
            xs16, vs16, us = [], [], []
            for l in range(L):
                V, W, b = params[3 * l], params[3 * l + 1], params[3 * l + 2]
                v16 = torch.empty((M, V.size(0)), dtype=torch.bfloat16, device=x0.device)
                ops.gemm_bf16(xl16, ops.cast_bf16(V, n_in, category="linear_fwd"), None, ACT_NONE, None, v16, category="linear_fwd")
                W16 = ops.cast_bf16(W, V.size(0), category="linear_fwd")
                xs16.append(xl16); vs16.append(v16)
                # x_{l+1} = x0 * (v W^T + b) + x_l inside the GEMM's epilogue (DLRM_CROSS_FUSE=0: the two kernels of round 4): the fp32 u
                # never goes through HBM, its bf16 copy is all the backward pass reads
                fused = ops.gemm_bf16_cross(v16, W16, b, x0, xl, want16=(l + 1 < L)) if CROSS_FUSE else None
                if fused is not None:
                    xl, xl16, u = fused
                    us.append(u)
                    continue
                u = torch.empty((M, n_in), dtype=torch.float32, device=x0.device)
                ops.gemm_bf16(v16, W16, b, ACT_NONE, u, None, category="linear_fwd")
                us.append(u)
                if l + 1 < L:
                    xl, xl16 = ops.cross_fwd(x0, u, xl, want16=True)
                else:
                    xl = ops.cross_fwd(x0, u, xl)
            ctx.arith, ctx.L = arith, L
            ctx.save_for_backward(*params, x0, *xs16, *vs16, *us)
            return xl
        xs, vs, us = [x0], [], []
        for l in range(L):
            V, W, b = params[3 * l], params[3 * l + 1], params[3 * l + 2]
            v = alloc2d(M, V.size(0), x0)
            ops.linear_fwd(xs[-1], V, None, ACT_NONE, v, arith)
            u = torch.empty((M, n_in), dtype=torch.float32, device=x0.device)
            ops.linear_fwd(v, W, b, ACT_NONE, u, arith)
            xs.append(ops.cross_fwd(x0, u, xs[-1]))
            vs.append(v); us.append(u)
        ctx.arith, ctx.L = arith, L
        ctx.save_for_backward(*params, *xs[:-1], *vs, *us)
        return xs[-1]

    @staticmethod
    def backward(ctx, g):
        L, arith = ctx.L, ctx.arith
        sv = ctx.saved_tensors
        if ctx.bf16:
            params, x0 = sv[:3 * L], sv[3 * L]
            xs16, vs16, us = sv[3 * L + 1:4 * L + 1], sv[4 * L + 1:5 * L + 1], sv[5 * L + 1:6 * L + 1]
            M, n_in = x0.shape
            g = g.contiguous()
            dx0 = torch.empty_like(x0)
            grads = [None] * (3 * L)
            gsum = None                                                               # g + sum of the V-path gradients so far (own buffer)
            for l in range(L - 1, -1, -1):
                V, W = params[3 * l], params[3 * l + 1]
                r = V.size(0)
                gl = g if gsum is None else gsum
                du16 = ops.cross_bwd(gl, x0, us[l], dx0, accumulate=(l != L - 1), out="bf16")   # du = g * x0 (bf16 only);  dx0 (+)= g * u_l
                dW, db = torch.empty_like(W), torch.empty(W.size(0), dtype=torch.float32, device=g.device)
                ops.linear_bwd_weight_bf16(du16, vs16[l], dW, db)
                dv16 = torch.empty((M, r), dtype=torch.bfloat16, device=g.device)
                ops.gemm_bf16(du16, ops.cast_bf16_transposed(W, n_in, category="linear_bwd_data"), None, ACT_NONE, None, dv16,
                              category="linear_bwd_data")
                dV = torch.empty_like(V)
                ops.linear_bwd_weight_bf16(dv16, xs16[l], dV, None)
                # gradient reaching x_l = identity path + V path: g + dv . V, summed in the GEMM epilogue (the incoming g itself is never
                # written: the first layer processed gets a fresh buffer, later ones accumulate in place)
                # (layer 0: x_l IS x_0, so the accumulated Hadamard-path gradient dx0 joins the same epilogue: dx0 + g + dv . V, in place)
                out = dx0 if l == 0 else (torch.empty_like(x0) if gsum is None else gsum)
                ops.gemm_bf16(dv16, ops.cast_bf16_transposed(V, r, category="linear_bwd_data"), None, ACT_NONE, out, None,
                              category="linear_bwd_data", addend=gl, addend2=dx0 if l == 0 else None)
                gsum = out
                grads[3 * l], grads[3 * l + 1], grads[3 * l + 2] = dV, dW, db
            return (None, gsum, *grads)
        params, xs, vs, us = sv[:3 * L], sv[3 * L:4 * L], sv[4 * L:5 * L], sv[5 * L:6 * L]
        x0 = xs[0]
        M, n_in = x0.shape
        g = g.contiguous()
        dx0 = torch.empty_like(x0)
        grads = [None] * (3 * L)
        for l in range(L - 1, -1, -1):
            V, W = params[3 * l], params[3 * l + 1]
            du = ops.cross_bwd(g, x0, us[l], dx0, accumulate=(l != L - 1))        # du = g * x0;  dx0 (+)= g * u_l
            dW, db = torch.empty_like(W), torch.empty(W.size(0), dtype=torch.float32, device=g.device)
            ops.linear_bwd_weight(du, vs[l], dW, db, arith=arith)
            dv = alloc2d(M, V.size(0), x0)
            ops.linear_bwd_data(du, W, None, ACT_NONE, dv, arith)
            dV = torch.empty_like(V)
            ops.linear_bwd_weight(dv, xs[l], dV, None, arith=arith)
            dxl = torch.empty_like(x0)
            ops.linear_bwd_data(dv, V, None, ACT_NONE, dxl, arith)
            g = ops.add(g, dxl)                                                   # gradient reaching x_l: identity path + V path
            grads[3 * l], grads[3 * l + 1], grads[3 * l + 2] = dV, dW, db
        dx0 = ops.add(dx0, g)                                                     # x_l of layer 0 IS x_0
        return (None, dx0, *grads)


class ChunkPackFunction(Function):
    """Re-orders the rows of the pooled-embedding send buffer for a PIPELINED all-to-all.

    `E` is [B, W] in global-batch order (rank r's samples are rows r*Bl .. (r+1)*Bl, Bl = B/N).  Chunk c of the pipeline
    holds, for every destination rank r, the c-th sub-slice (Bc = Bl/C rows) of r's samples; the function returns C tensors
    [N*Bc, W] (views of one [C, N, Bc, W] buffer), each of which is a complete, smaller all-to-all send buffer.  Backward
    scatters the C gradient buffers back to global-batch order for the fused embedding update."""

    @staticmethod
    def forward(ctx, E, N, C):
        B, W = E.shape
        Bc = B // (N * C)
        if Bc * N * C != B:
            raise RuntimeError("dlrm_amd: batch %d does not split into %d ranks x %d chunks" % (B, N, C))
        packed = _rowmajor(E).reshape(N, C, Bc, W).permute(1, 0, 2, 3).contiguous()
        ctx.dims = (N, C, Bc, W)
        return tuple(packed[c].view(N * Bc, W) for c in range(C))

    @staticmethod
    def backward(ctx, *grads):
        N, C, Bc, W = ctx.dims
        dE = torch.empty((N, C, Bc, W), dtype=grads[0].dtype, device=grads[0].device)
        for c, g in enumerate(grads):
            dE[:, c].copy_(g.reshape(N, Bc, W))
        return dE.view(N * C * Bc, W), None, None


class CatFunction(Function):
    """R = torch.cat(blocks, dim=1) — the "cat" interaction (dlrm_s_pytorch.py:505-507).

    forward(shared, block0, block1, ...): `shared` is None (R is allocated and filled by ONE strided-copy launch), or an
    OutSlot over a buffer in which the blocks ALREADY sit side by side in order (the [B, (1+T)*D] feature buffer the
    bottom tower and the embedding kernel wrote into): then R is that buffer and nothing is copied.  Backward hands out
    column views of dR (consumers take row strides)."""

    @staticmethod
    def forward(ctx, shared, *blocks):
        ctx.widths = [b.size(1) for b in blocks]
        if shared is not None:
            R = shared.get()
            if R.size(1) != sum(ctx.widths):
                raise RuntimeError("dlrm_amd: CatFunction shared buffer width mismatch")
            return R
        blocks = [_rowmajor(b) for b in blocks]
        R = alloc2d(blocks[0].size(0), sum(ctx.widths), blocks[0])
        dsts, o = [], 0
        for w in ctx.widths:
            dsts.append(R[:, o:o + w])
            o += w
        ops.copy_blocks(blocks, dsts)
        return R

    @staticmethod
    def backward(ctx, dR):
        outs, o = [], 0
        for w in ctx.widths:
            outs.append(dR[:, o:o + w])
            o += w
        return (None, *outs)


class ClampFunction(Function):
    """torch.clamp(p, lo, hi) of the predictions (--loss-threshold, dlrm_s_pytorch.py:580-583,607-610)."""

    @staticmethod
    def forward(ctx, p, lo, hi):
        p = p.contiguous()
        ctx.save_for_backward(p)
        ctx.lo, ctx.hi = lo, hi
        return ops.clamp(p, lo, hi)

    @staticmethod
    def backward(ctx, g):
        (p,) = ctx.saved_tensors
        return ops.clamp_bwd(p, ctx.lo, ctx.hi, g), None, None


class BCELossFunction(Function):
    """BCELoss(reduction='mean'); loss and dL/dp are produced by one kernel pass.  class_weights = (w_neg, w_pos) turns it
    into the reference's weighted BCE (mean of loss_ws[T.long()] * bce, loss_fn_wrap dlrm_s_pytorch.py:150-156)."""

    @staticmethod
    def forward(ctx, p, target, weights, class_weights=(1.0, 1.0)):
        loss, dp = ops.bce_loss(p.contiguous(), target.contiguous(), weights, 1.0, want_grad=True, class_weights=class_weights)
        ctx.save_for_backward(dp)
        ctx.shape = p.shape
        return loss.reshape(())

    @staticmethod
    def backward(ctx, g):
        (dp,) = ctx.saved_tensors
        return ops.scale_by_scalar(dp, g.reshape(1).float()).view(ctx.shape), None, None, None


class BCEWithLogitsLossFunction(Function):
    """BCEWithLogitsLoss(reduction='mean') on raw logits (torchrec DLRMTrain); loss and dL/dlogits in one kernel pass."""

    @staticmethod
    def forward(ctx, z, target):
        loss, dz = ops.bce_logits_loss(z.contiguous(), target.contiguous(), 1.0, want_grad=True)
        ctx.save_for_backward(dz)
        ctx.shape = z.shape
        return loss.reshape(())

    @staticmethod
    def backward(ctx, g):
        (dz,) = ctx.saved_tensors
        return ops.scale_by_scalar(dz, g.reshape(1).float()).view(ctx.shape), None


class BCEElementwiseFunction(Function):
    """BCELoss(reduction='none'): the per-sample loss the reference's wbce path multiplies by class weights and averages
    in loss_fn_wrap (dlrm_s_pytorch.py:388-391, 150-156)."""

    @staticmethod
    def forward(ctx, p, target):
        p, target = p.contiguous(), target.contiguous()
        ctx.save_for_backward(p, target)
        return ops.bce_elementwise(p, target)

    @staticmethod
    def backward(ctx, g):
        p, target = ctx.saved_tensors
        return ops.bce_elementwise_bwd(p, target, g.to(torch.float32)), None


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
        return ops.scale_by_scalar(dp, g.reshape(1).float()).view(ctx.shape), None
