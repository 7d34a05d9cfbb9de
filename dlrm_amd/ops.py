"""Tensor-level wrappers around the C ABI (no autograd here; see functional.py).

Every function enqueues HIP kernels on torch's current stream for the tensors' device and returns
immediately.  Tensors must be fp32 CUDA(HIP) tensors whose last dimension is contiguous; leading
dimensions are passed as explicit strides so strided views (e.g. a slot of the [B, F, D] interaction
buffer) are used in place.  Anything else raises — there is no CPU/eager fallback.
"""
from __future__ import annotations

import os

import ctypes as C
from typing import List, Optional, Sequence, Union

import torch

from . import _lib
from ._lib import ACT_NONE, ACT_RELU, ACT_SIGMOID, UPD_ATOMIC, UPD_DETERMINISTIC, UPD_SORTED  # noqa: F401


def _stream(t: Optional[torch.Tensor] = None) -> C.c_void_p:
    """torch's current stream of the device the kernels will run on.  HIP launches go to the CURRENT device, so an operand
    that lives on another device (a model on cuda:1 in a process that never called torch.cuda.set_device(1)) is refused
    loudly instead of launching device-0 kernels on device-1 pointers."""
    if t is not None and t.is_cuda and t.device.index != torch.cuda.current_device():
        raise RuntimeError(f"dlrm_amd: operand on {t.device} but the current device is cuda:{torch.cuda.current_device()}; "
                           "call torch.cuda.set_device(...) (one process per GPU) before using the model")
    return C.c_void_p(torch.cuda.current_stream().cuda_stream)


class KernelTimers:
    """Optional HIP-event timing of the C-ABI launches, on the stream they are enqueued on (bench.py turns it on inside its
    timed region; None by default).

    One event is recorded where the launch CATEGORY changes (e.g. linear_fwd -> emb_fwd), not around every launch: a run of
    consecutive launches of one category on one stream is a single (start, end) pair, and the end of one run is the start
    of the next.  Event records are not free on this hardware — each is a barrier packet, ~5 us of idle queue: two per
    launch cost 0.33 ms of a 9.1 ms step (rocprofv3 trace, profiles/r03/ceilings.md) and inflated every per-kernel time
    by 10 us — so this keeps ~35 of them per step instead of 120.  A run's time covers whatever the stream executed
    between its first launch and the next category's first launch: `mark()` closes the open run early wherever something
    that is not a kernel of this library follows (a collective, a cross-stream wait)."""

    def __init__(self):
        self.records = {}          # name -> {"calls": launches, "pairs": [(start_event, end_event), ...]}
        self._enabled = True       # bench.py samples a subset of its timed steps
        self._open = None          # [name, start_event, torch stream, launches]

    @property
    def enabled(self):
        return self._enabled

    @enabled.setter
    def enabled(self, on):
        if not on:
            self.mark()
        self._enabled = bool(on)

    def _close(self, end_event):
        name, start, _, calls = self._open
        rec = self.records.setdefault(name, {"calls": 0, "pairs": []})
        rec["calls"] += calls
        rec["pairs"].append((start, end_event))
        self._open = None

    def mark(self):
        """close the open run now (its end event is recorded on the stream the run was launched on)"""
        if self._open is not None:
            e = torch.cuda.Event(enable_timing=True)
            e.record(self._open[2])
            self._close(e)

    def enter(self, name):
        if not self._enabled:
            return
        st = torch.cuda.current_stream()
        o = self._open
        if o is not None and o[2] == st:
            if o[0] == name:
                o[3] += 1
                return
            e = torch.cuda.Event(enable_timing=True)
            e.record(st)
            self._close(e)                       # shared: end of the previous run == start of this one
        else:
            self.mark()                          # a run on another stream ends with an event of its own
            e = torch.cuda.Event(enable_timing=True)
            e.record(st)
        self._open = [name, e, st, 1]

    def summary(self):
        self.mark()
        torch.cuda.synchronize()
        out = {}
        for name, rec in self.records.items():
            total = float(sum(a.elapsed_time(b) for a, b in rec["pairs"]))
            out[name] = {"calls": rec["calls"], "total_ms": total, "avg_ms": total / max(rec["calls"], 1)}
        return out


timers: Optional[KernelTimers] = None


def timer_mark() -> None:
    """called by code that enqueues non-library work (collectives, cross-stream waits): ends the open timing run"""
    if timers is not None:
        timers.mark()


CALL_COUNT = [0]      # categorised C-ABI calls since import (bench.py reports calls per step for the launch-bound workload)


class _timed:
    __slots__ = ("name",)

    def __init__(self, name):
        self.name = name

    def __enter__(self):
        CALL_COUNT[0] += 1
        if timers is not None:
            timers.enter(self.name)
        return self

    def __exit__(self, *exc):
        return False


def _req(t: torch.Tensor, name: str, dtype=torch.float32, ndim: Optional[int] = None) -> None:
    if not isinstance(t, torch.Tensor) or not t.is_cuda:
        raise RuntimeError(f"dlrm_amd: `{name}` must be a GPU tensor (the HIP path has no CPU fallback)")
    if t.dtype != dtype:
        raise RuntimeError(f"dlrm_amd: `{name}` must be {dtype}, got {t.dtype}")
    if ndim is not None and t.dim() != ndim:
        raise RuntimeError(f"dlrm_amd: `{name}` must be {ndim}-D, got shape {tuple(t.shape)}")
    if t.dim() >= 1 and t.numel() > 0 and t.stride(-1) != 1:
        raise RuntimeError(f"dlrm_amd: `{name}` must be contiguous in its last dimension")


def _ld(t: torch.Tensor) -> int:
    """leading dimension (elements) of a 2-D row-major view"""
    return t.stride(0) if t.size(0) > 1 else max(t.stride(0), t.size(1))


# ------------------------------------------------------------------------------------------------
# out-of-range embedding indices
# ------------------------------------------------------------------------------------------------
_err_blocks = {}   # device index -> pinned host int64[4] the embedding kernels report into (include/dlrm_hip.h, `err`)


def _err_block(device: torch.device) -> torch.Tensor:
    """Pinned host memory is device-visible under HIP's unified addressing: the kernels store {1, table, index, rows}
    straight into it when they meet an index outside [0, rows), and the host polls it WITHOUT synchronising."""
    key = device.index if device.index is not None else torch.cuda.current_device()
    blk = _err_blocks.get(key)
    if blk is None:
        blk = torch.zeros(4, dtype=torch.int64).pin_memory()
        _err_blocks[key] = blk
    return blk


def check_index_errors(device=None, sync: bool = False) -> None:
    """Raise IndexError if an embedding kernel that has FINISHED met an out-of-range index (the lookup was skipped on the
    device; the reference's EmbeddingBag raises at the call).  sync=True first waits for the device, which makes the
    check exact for everything enqueued so far; without it the check costs one host memory read (DLRM_Net.forward does
    that at every step, so bad data is reported at most one step late)."""
    if sync and torch.cuda.is_available():
        torch.cuda.synchronize(device)
    for key, blk in _err_blocks.items():
        if device is not None and torch.device(device).index not in (None, key):
            continue
        if int(blk[0]) != 0:
            _, t, idx, rows = blk.tolist()
            blk.zero_()
            if rows == -1:
                raise IndexError(f"dlrm_amd: the fused one-lookup-per-bag embedding + interaction path met a bag of table {t} that "
                                 f"does not start at its own position (offset {-idx - 1}); set model.fuse_emb_interact = False for "
                                 f"multi-hot / ragged batches (cuda:{key})")
            raise IndexError(f"dlrm_amd: embedding index out of range: table {t}, index {idx}, rows {rows} (cuda:{key})")


# ------------------------------------------------------------------------------------------------
# embedding bags
# ------------------------------------------------------------------------------------------------
class BagBatch:
    """Host-side descriptor of one batch of bags for T tables (device pointers + sizes).

    Mirrors what `apply_emb` receives (dlrm_s_pytorch.py:407): lS_o[k] = bag starts of table k,
    lS_i[k] = flat indices of table k; both either lists of 1-D tensors or stacked 2-D tensors."""

    def __init__(self, lS_o, lS_i, per_sample_weights: Optional[Sequence[Optional[torch.Tensor]]] = None):
        T = len(lS_i)
        if len(lS_o) != T:
            raise RuntimeError("dlrm_amd: offsets / indices table counts differ")
        self.T = T
        self.ignore_oob = False   # True: out-of-range ids are expected (row-wise shards: ids of other ranks' rows) — skipped, not reported
        self.keep = []  # keep tensors alive while kernels may still read them
        if isinstance(lS_i, torch.Tensor) and isinstance(lS_o, torch.Tensor) and lS_i.dim() == 2 and lS_o.dim() == 2:
            # stacked [T, n] inputs (what the reference's collate functions build, dlrm_data_pytorch.py:686,337): the
            # per-table pointers are plain address arithmetic — no 2*T view tensors per step on the host path
            if lS_i.dtype not in (torch.int64, torch.int32) or lS_o.dtype != lS_i.dtype:
                raise RuntimeError("dlrm_amd: indices/offsets must both be int64 or both int32")
            if not lS_i.is_cuda or not lS_o.is_cuda:
                raise RuntimeError("dlrm_amd: indices/offsets must be GPU tensors")
            if lS_i.stride(1) != 1 and lS_i.size(1) > 1:
                lS_i = lS_i.contiguous()
            if lS_o.stride(1) != 1 and lS_o.size(1) > 1:
                lS_o = lS_o.contiguous()
            self.keep += [lS_i, lS_o]
            isz = lS_i.element_size()
            B, n_i = lS_o.size(1), lS_i.size(1)
            dt = lS_i.dtype
            idx_ptrs = [lS_i.data_ptr() + k * lS_i.stride(0) * isz if n_i else 0 for k in range(T)]
            off_ptrs = [lS_o.data_ptr() + k * lS_o.stride(0) * isz for k in range(T)]
            nnz = [n_i] * T
            self._idx_src = lS_i
        else:
            idx_ptrs, off_ptrs, nnz = [], [], []
            dt = None
            B = None
            self._idx_src = []
            for k in range(T):
                i_k, o_k = lS_i[k], lS_o[k]
                if i_k.dtype not in (torch.int64, torch.int32) or o_k.dtype != i_k.dtype:
                    raise RuntimeError("dlrm_amd: indices/offsets must both be int64 or both int32")
                if dt is None:
                    dt = i_k.dtype
                elif dt != i_k.dtype:
                    raise RuntimeError("dlrm_amd: all tables must use the same index dtype")
                if not i_k.is_cuda or not o_k.is_cuda:
                    raise RuntimeError("dlrm_amd: indices/offsets must be GPU tensors")
                if not i_k.is_contiguous():
                    i_k = i_k.contiguous()
                if not o_k.is_contiguous():
                    o_k = o_k.contiguous()
                if B is None:
                    B = o_k.numel()
                elif B != o_k.numel():
                    raise RuntimeError("dlrm_amd: every table must have the same number of bags")
                self.keep += [i_k, o_k]
                self._idx_src.append(i_k)
                idx_ptrs.append(i_k.data_ptr() if i_k.numel() else 0)
                off_ptrs.append(o_k.data_ptr())
                nnz.append(i_k.numel())
        self.B = int(B)
        self.idx_bits = 64 if dt == torch.int64 else 32
        self.nnz = nnz
        self._idx = _lib.ptr_array(idx_ptrs)
        self._off = _lib.ptr_array(off_ptrs)
        self._nnz = _lib.i64_array(nnz)
        self._psw = None
        if per_sample_weights is not None and any(w is not None for w in per_sample_weights):
            ptrs = []
            for k, w in enumerate(per_sample_weights):
                if w is None:
                    ptrs.append(0)
                else:
                    _req(w, "per_sample_weights")
                    if w.numel() != nnz[k]:
                        raise RuntimeError("dlrm_amd: per_sample_weights size mismatch")
                    w = w.contiguous()
                    self.keep.append(w)
                    ptrs.append(w.data_ptr())
            self._psw = _lib.ptr_array(ptrs)


def pool_weights_gather(vws: Sequence[torch.Tensor], bags: "BagBatch") -> List[torch.Tensor]:
    """psw[t][i] = vws[t][idx_t[i]] (`v_W_l[k].gather(0, indices)`, dlrm_s_pytorch.py:425-426) for all tables in one launch;
    attaches the result to `bags` as its per-sample weights."""
    lib = _lib.load()
    if len(vws) != bags.T:
        raise RuntimeError("dlrm_amd: pool_weights_gather needs one weight vector per table")
    for v in vws:
        _req(v, "pooling weights", ndim=1)
    dev = vws[0].device
    psw = [torch.empty(n, dtype=torch.float32, device=dev) for n in bags.nnz]
    rc = lib.dlrm_pool_weights_gather(bags.T, _lib.i64_array([v.numel() for v in vws]), bags._idx, bags._nnz, bags.idx_bits,
                                      _lib.ptr_array([v.data_ptr() for v in vws]),
                                      _lib.ptr_array([p_.data_ptr() if p_.numel() else 0 for p_ in psw]),
                                      C.c_void_p(_err_block(dev).data_ptr()), _stream(vws[0]))
    _lib.check(rc, "dlrm_pool_weights_gather")
    bags.keep += psw
    bags._psw = _lib.ptr_array([p_.data_ptr() if p_.numel() else 0 for p_ in psw])
    return psw


def emb_psw_grad(weights: Sequence[torch.Tensor], bags: "BagBatch", dout: torch.Tensor, like: Sequence[torch.Tensor]) -> List[torch.Tensor]:
    """dense gradients of the learned pooling-weight vectors: dvW_t[r] = sum_{lookups of row r} <dout[bag], W_t[r]>"""
    lib = _lib.load()
    D, wp, rows = _weights_desc(weights)
    _req(dout, "dout", ndim=2)
    dvw = [torch.empty_like(v) for v in like]
    rc = lib.dlrm_emb_psw_grad(bags.T, bags.B, D, wp, rows, bags._idx, bags._off, bags._nnz, bags.idx_bits,
                               C.c_void_p(dout.data_ptr()), _ld(dout), _lib.ptr_array([g.data_ptr() for g in dvw]), _stream(dout))
    _lib.check(rc, "dlrm_emb_psw_grad")
    return dvw


def bag_index_tensor(bags: "BagBatch", k: int) -> torch.Tensor:
    """the 1-D index tensor of table k as it was passed in (the COO gradient's indices, verbatim)"""
    return bags._idx_src[k]


def _weights_desc(weights: Sequence[torch.Tensor]):
    D = None
    for w in weights:
        _req(w, "embedding weight", ndim=2)
        if not w.is_contiguous():
            raise RuntimeError("dlrm_amd: embedding tables must be contiguous [rows, D]")
        if D is None:
            D = w.size(1)
        elif D != w.size(1):
            raise RuntimeError("dlrm_amd: all embedding tables must share one embedding dimension")
    return int(D), _lib.ptr_array([w.data_ptr() for w in weights]), _lib.i64_array([w.size(0) for w in weights])


def _pred_args(pred):
    """pred = None | (flag, nonzero): flag a device int32 tensor; the launch runs iff (flag != 0) == bool(nonzero) (include/dlrm_hip.h, ABI 16)"""
    flag, nonzero = pred
    if flag.dtype != torch.int32 or not flag.is_cuda or flag.numel() < 1:
        raise RuntimeError("dlrm_amd: a launch predicate is a device int32 tensor")
    return C.c_void_p(flag.data_ptr()), int(bool(nonzero))


def emb_fwd(weights: Sequence[torch.Tensor], bags: BagBatch, out: torch.Tensor, pred=None) -> torch.Tensor:
    """out[b, t*D:(t+1)*D] = sum-pooled bag (t, b).  `out` is a [B, >= T*D] view (row stride free).  pred: see _pred_args."""
    lib = _lib.load()
    D, wp, rows = _weights_desc(weights)
    _req(out, "out", ndim=2)
    if out.size(0) != bags.B or out.size(1) < bags.T * D or len(weights) != bags.T:
        raise RuntimeError("dlrm_amd: emb_fwd shape mismatch")
    err = None if bags.ignore_oob else C.c_void_p(_err_block(out.device).data_ptr())
    # (a predicated launch belongs to the step's fused lookup + interaction: same timing category, so the three launches are ONE run)
    with _timed("emb_fwd" if pred is None else "emb_interact_fwd"):
        if pred is None:
            rc = lib.dlrm_emb_fwd(bags.T, bags.B, D, wp, rows, bags._idx, bags._off, bags._nnz, bags._psw,
                                  bags.idx_bits, C.c_void_p(out.data_ptr()), _ld(out), err, _stream(out))
        else:
            rc = lib.dlrm_emb_fwd_pred(bags.T, bags.B, D, wp, rows, bags._idx, bags._off, bags._nnz, bags._psw,
                                       bags.idx_bits, C.c_void_p(out.data_ptr()), _ld(out), err, *_pred_args(pred), _stream(out))
    _lib.check(rc, "dlrm_emb_fwd")
    return out


_emb_ws = {}   # (device, stream) -> cached workspace of the sort-based updates.  Per stream, like the split-k slabs: kernels of one stream
               # are ordered, so one workspace suffices — and a workspace allocated under one stream is never handed to kernels of another
               # (the caching allocator orders re-use of freed memory within the allocating stream only)


def _emb_workspace(need: int, device) -> torch.Tensor:
    key = (device, torch.cuda.current_stream(device).cuda_stream)
    ws = _emb_ws.get(key)
    if ws is None or ws.numel() < need:
        ws = torch.empty(max(int(need), 256), dtype=torch.uint8, device=device)
        _emb_ws[key] = ws
    return ws


def sort_is_graph_safe(weights: Sequence[torch.Tensor], bags: BagBatch) -> bool:
    """True when the sort-based embedding updates of these shapes run entirely on the library's own segmented sorter
    (csrc/seg_sort.h: plain kernels, replayable inside a HIP graph); False when a table segment needs the general sorter."""
    if len(weights) != bags.T:
        raise RuntimeError("dlrm_amd: sort_is_graph_safe needs one table per bag list")
    _, _, rows = _weights_desc(weights)
    return _lib.load().dlrm_emb_sort_kind(bags.T, bags._nnz, rows) == 1


def sort_lookups(rows: Sequence[int], bags: BagBatch):
    """The (table, row) sort of the sort-based updates on its own (dlrm_emb_sort_lookups; bags.T <= 32): returns
    (positions uint32-as-int64 [L], keys int64 [L] = table << row_bits | row, bag_of int64 [L] (-1: skipped lookup), row_bits)."""
    lib = _lib.load()
    dev = bags.keep[0].device
    rows_a = _lib.i64_array([int(r) for r in rows])
    need = lib.dlrm_emb_bwd_workspace_bytes(bags.T, bags._nnz, rows_a)
    if need < 0:
        raise RuntimeError("dlrm_amd: dlrm_emb_bwd_workspace_bytes failed")
    ws = _emb_workspace(need, dev)
    L = int(sum(bags.nnz))
    pos = torch.empty(L, dtype=torch.int32, device=dev)
    keys = torch.empty(L, dtype=torch.int64, device=dev)
    bag_of = torch.empty(L, dtype=torch.int32, device=dev)
    rb = C.c_int(0)
    rc = lib.dlrm_emb_sort_lookups(bags.T, bags.B, rows_a, bags._idx, bags._off, bags._nnz, bags.idx_bits, C.c_void_p(ws.data_ptr()),
                                   ws.numel(), C.c_void_p(pos.data_ptr()), C.c_void_p(keys.data_ptr()), C.c_void_p(bag_of.data_ptr()),
                                   C.byref(rb), C.c_void_p(_err_block(dev).data_ptr()), _stream(pos))
    _lib.check(rc, "dlrm_emb_sort_lookups")
    return pos.long() & 0xFFFFFFFF, keys, bag_of.long(), rb.value


# ---- learning rates: a Python float (by value in the launch) or a 1-element fp32 GPU tensor the kernel reads WHEN IT RUNS (include/dlrm_hip.h,
# "LEARNING RATES").  The second form is what a captured whole-step HIP graph takes, so that the reference's per-iteration LRPolicyScheduler
# (dlrm_s_pytorch.py:169-203, stepped at :1621) is followed without re-capture: GraphedTrainStep registers one device scalar per param group for
# the duration of the capture, the optimizer-side callers ask device_lr(group, value) for it.
# the bf16-shaped GEMM kernels (csrc/gemm_bf16.hip) are what the product library runs wherever their preconditions hold; the fp32-shaped
# ones can only be forced in a TUNING build of the library (DLRM_HIP_LIB=... DLRM_BF16_PHASED=0: tools/bf16_gemm_bench.py), and only then does
# the host plan follow the switch
BF16_PHASED = not (os.environ.get("DLRM_HIP_LIB") and os.environ.get("DLRM_BF16_PHASED", "1") == "0")

LrLike = Union[float, torch.Tensor]
_graph_lr = {}          # id(param_group dict) -> 1-element fp32 GPU tensor; populated only while a GraphedTrainStep captures


def device_lr(group, value):
    """the device scalar registered for this param group (inside a graph capture), else `value` unchanged"""
    t = _graph_lr.get(id(group))
    return t if t is not None else value


def _lr_args(lr: LrLike):
    if isinstance(lr, torch.Tensor):
        if lr.dtype != torch.float32 or not lr.is_cuda or lr.numel() != 1:
            raise RuntimeError("dlrm_amd: a device-side learning rate must be a 1-element float32 GPU tensor")
        return 0.0, C.c_void_p(lr.data_ptr())
    return float(lr), None


def emb_bwd_sgd(weights: Sequence[torch.Tensor], bags: BagBatch, dout: torch.Tensor, lr: LrLike,
                mode: int = UPD_SORTED) -> None:
    """Fused EmbeddingBag backward + sparse SGD: W_t[idx] -= lr * dout[bag, t*D:(t+1)*D] (in place)."""
    lib = _lib.load()
    D, wp, rows = _weights_desc(weights)
    _req(dout, "dout", ndim=2)
    if dout.size(0) != bags.B or dout.size(1) < bags.T * D or len(weights) != bags.T:
        raise RuntimeError("dlrm_amd: emb_bwd_sgd shape mismatch")
    ws_ptr, ws_bytes = None, 0
    if mode == UPD_SORTED:
        need = lib.dlrm_emb_bwd_workspace_bytes(bags.T, bags._nnz, rows)
        if need < 0:
            raise RuntimeError("dlrm_amd: dlrm_emb_bwd_workspace_bytes failed")
        ws = _emb_workspace(need, dout.device)
        ws_ptr, ws_bytes = C.c_void_p(ws.data_ptr()), ws.numel()
    lr_v, lr_p = _lr_args(lr)
    with _timed("emb_bwd_sgd"):
        rc = lib.dlrm_emb_bwd_sgd(bags.T, bags.B, D, wp, rows, bags._idx, bags._off, bags._nnz, bags._psw,
                                  bags.idx_bits, C.c_void_p(dout.data_ptr()), _ld(dout), lr_v, lr_p, int(mode),
                                  ws_ptr, ws_bytes, None if bags.ignore_oob else C.c_void_p(_err_block(dout.device).data_ptr()), _stream(dout))
    _lib.check(rc, "dlrm_emb_bwd_sgd")


class Presorted:
    """What dlrm_emb_presort left behind for one backward pass: the workspace holding the lookups sorted by (table, row), the per-sample
    mask of SINGLE lookups (bit t of mask[b]: lookup (t, b) is the only one of the batch naming its row), the step size both update calls
    use and the launch predicate the fused backward ran under (None: unconditional)."""

    __slots__ = ("ws", "mask", "lr", "pred")

    def __init__(self, ws, mask, lr, pred):
        self.ws, self.mask, self.lr, self.pred = ws, mask, lr, pred


def presort_ok(weights: Sequence[torch.Tensor], bags: BagBatch) -> bool:
    """shapes dlrm_emb_presort / dlrm_emb_bwd_sgd_presorted take: one launch group, one lookup per bag position, D = 128, no pooling weights"""
    return (0 < bags.T <= 32 and len(weights) == bags.T and all(n == bags.B for n in bags.nnz) and bags._psw is None
            and all(w.size(1) == 128 and w.is_contiguous() and w.data_ptr() % 16 == 0 for w in weights))


def emb_presort(weights: Sequence[torch.Tensor], bags: BagBatch, lr: LrLike, pred=None) -> Presorted:
    """The sort of the sparse update, IN FRONT of the fused backward (dlrm_emb_presort, ABI 17): a fresh workspace (it must survive until
    emb_bwd_sgd_presorted consumes it, whatever else sorts in between) + the mask of single lookups."""
    lib = _lib.load()
    D, wp, rows = _weights_desc(weights)
    dev = weights[0].device
    need = lib.dlrm_emb_bwd_workspace_bytes(bags.T, bags._nnz, rows)
    if need < 0:
        raise RuntimeError("dlrm_amd: dlrm_emb_bwd_workspace_bytes failed")
    ws = torch.empty(max(int(need), 256), dtype=torch.uint8, device=dev)
    mask = torch.empty(bags.B, dtype=torch.int32, device=dev)
    with _timed("emb_bwd_sgd"):
        rc = lib.dlrm_emb_presort(bags.T, bags.B, rows, bags._idx, bags._off, bags._nnz, bags.idx_bits, C.c_void_p(ws.data_ptr()), ws.numel(),
                                  C.c_void_p(mask.data_ptr()), None if bags.ignore_oob else C.c_void_p(_err_block(dev).data_ptr()), _stream(mask))
    _lib.check(rc, "dlrm_emb_presort")
    return Presorted(ws, mask, lr, pred)


def emb_bwd_sgd_presorted(weights: Sequence[torch.Tensor], bags: BagBatch, dout: torch.Tensor, pre: Presorted, skip_singles: bool = True) -> None:
    """The second half of emb_bwd_sgd(UPD_SORTED) from a Presorted workspace; skip_singles: the lookups the fused backward already applied
    (interact_bwd_gather(..., presorted=pre)) are left out — when the launch predicate it ran under holds."""
    lib = _lib.load()
    D, wp, rows = _weights_desc(weights)
    _req(dout, "dout", ndim=2)
    if dout.size(0) != bags.B or dout.size(1) < bags.T * D or len(weights) != bags.T:
        raise RuntimeError("dlrm_amd: emb_bwd_sgd_presorted shape mismatch")
    lr_v, lr_p = _lr_args(pre.lr)
    flag, nz = _pred_args(pre.pred) if pre.pred is not None else (None, 0)
    with _timed("emb_bwd_sgd"):
        rc = lib.dlrm_emb_bwd_sgd_presorted(bags.T, bags.B, D, wp, rows, bags._nnz, C.c_void_p(dout.data_ptr()), _ld(dout), lr_v, lr_p,
                                            C.c_void_p(pre.ws.data_ptr()), pre.ws.numel(), int(bool(skip_singles)), flag, nz, _stream(dout))
    _lib.check(rc, "dlrm_emb_bwd_sgd_presorted")


def emb_bwd_rowwise_adagrad(weights: Sequence[torch.Tensor], states: Sequence[torch.Tensor], bags: BagBatch,
                            dout: torch.Tensor, lr: LrLike, eps: float) -> None:
    """Fused EmbeddingBag backward + row-wise sparse Adagrad (optim/rwsadagrad.py:117-143), in place:
    per touched row r:  g_r = sum of its lookups' gradients;  states[t][r] += mean(g_r^2);
    W_t[r] -= lr * g_r / (sqrt(states[t][r]) + eps).  `lr` is the decayed clr of rwsadagrad.py:115."""
    lib = _lib.load()
    D, wp, rows = _weights_desc(weights)
    _req(dout, "dout", ndim=2)
    if dout.size(0) != bags.B or dout.size(1) < bags.T * D or len(weights) != bags.T or len(states) != bags.T:
        raise RuntimeError("dlrm_amd: emb_bwd_rowwise_adagrad shape mismatch")
    for s_, w in zip(states, weights):
        _req(s_, "adagrad state", ndim=1)
        if s_.numel() != w.size(0) or not s_.is_contiguous():
            raise RuntimeError("dlrm_amd: row-wise adagrad state must be a contiguous [rows] tensor")
    sp = _lib.ptr_array([s_.data_ptr() for s_ in states])
    need = lib.dlrm_emb_adagrad_workspace_bytes(bags.T, D, bags._nnz, rows)
    if need < 0:
        raise RuntimeError("dlrm_amd: dlrm_emb_adagrad_workspace_bytes failed")
    ws = _emb_workspace(need, dout.device)
    lr_v, lr_p = _lr_args(lr)
    with _timed("emb_bwd_adagrad"):
        rc = lib.dlrm_emb_bwd_rowwise_adagrad(bags.T, bags.B, D, wp, sp, rows, bags._idx, bags._off, bags._nnz,
                                              bags._psw, bags.idx_bits, C.c_void_p(dout.data_ptr()), _ld(dout),
                                              lr_v, lr_p, float(eps), C.c_void_p(ws.data_ptr()), ws.numel(),
                                              None if bags.ignore_oob else C.c_void_p(_err_block(dout.device).data_ptr()), _stream(dout))
    _lib.check(rc, "dlrm_emb_bwd_rowwise_adagrad")


def emb_bwd_coo(bags: BagBatch, dout: torch.Tensor, D: int) -> List[torch.Tensor]:
    """The reference's EmbeddingBag backward WITHOUT the fused update: per table the [nnz_t, D] value block of the sparse
    COO gradient (values[i] = psw[i] * dout[bag(i), t*D:(t+1)*D]; its indices are the lookup indices verbatim)."""
    lib = _lib.load()
    _req(dout, "dout", ndim=2)
    if dout.size(0) != bags.B or dout.size(1) < bags.T * D:
        raise RuntimeError("dlrm_amd: emb_bwd_coo shape mismatch")
    values = [torch.empty((n, D), dtype=torch.float32, device=dout.device) for n in bags.nnz]
    with _timed("emb_bwd_coo"):
        rc = lib.dlrm_emb_bwd_coo(bags.T, bags.B, D, bags._off, bags._nnz, bags._psw, bags.idx_bits,
                                  C.c_void_p(dout.data_ptr()), _ld(dout),
                                  _lib.ptr_array([v.data_ptr() if v.numel() else 0 for v in values]), _stream())
    _lib.check(rc, "dlrm_emb_bwd_coo")
    return values


# ------------------------------------------------------------------------------------------------
# interaction
# ------------------------------------------------------------------------------------------------
def _feature_table(tensors: Sequence[torch.Tensor], D: int):
    """Each [B, k*D] tensor contributes k features: (ptr + j*D*4, row stride)."""
    ptrs, lds = [], []
    B = tensors[0].size(0)
    for t in tensors:
        _req(t, "feature block", ndim=2)
        if t.size(0) != B or t.size(1) % D != 0:
            raise RuntimeError("dlrm_amd: feature blocks must be [B, k*D]")
        for j in range(t.size(1) // D):
            ptrs.append(t.data_ptr() + 4 * j * D)
            lds.append(_ld(t))
    return B, ptrs, lds


def interact_out_width(F: int, D: int, self_interaction) -> int:
    """self_interaction: False/0 = strictly lower triangle (the reference's order), True/1 = with the diagonal, 2 = torchrec's
    torch.triu_indices(F, F, 1) order (same pairs as 0, permuted columns)"""
    return D + (F * (F + 1) // 2 if (int(self_interaction) & 1) else F * (F - 1) // 2)


def _permute(order, *lists):
    return lists if order is None else tuple([l[i] for i in order] for l in lists)


def interact_fwd(blocks: Sequence[torch.Tensor], D: int, self_interaction: bool, R: torch.Tensor, order=None, pred=None) -> torch.Tensor:
    """order: optional permutation — canonical feature f is the order[f]-th feature of the block list (lets the interaction read
    features that live in several buffers, e.g. all-to-all blocks of table-wise shards + the reduce-scatter block of row-wise
    ones, in global table order without copying them together)"""
    lib = _lib.load()
    B, ptrs, lds = _feature_table(blocks, D)
    ptrs, lds = _permute(order, ptrs, lds)
    _req(R, "R", ndim=2)
    F = len(ptrs)
    if R.size(0) != B or R.size(1) < interact_out_width(F, D, self_interaction):
        raise RuntimeError("dlrm_amd: interact_fwd output shape mismatch")
    with _timed("interact_fwd" if pred is None else "emb_interact_fwd"):
        if pred is None:
            rc = lib.dlrm_interact_fwd(B, F, D, _lib.ptr_array(ptrs), _lib.i64_array(lds), int(self_interaction),
                                       C.c_void_p(R.data_ptr()), _ld(R), _stream(R))
        else:
            rc = lib.dlrm_interact_fwd_pred(B, F, D, _lib.ptr_array(ptrs), _lib.i64_array(lds), None, None, None, 64, int(self_interaction),
                                            C.c_void_p(R.data_ptr()), _ld(R), None, *_pred_args(pred), _stream(R))
    _lib.check(rc, "dlrm_interact_fwd")
    return R


def gather_ok(F: int, D: int) -> bool:
    return bool(_lib.load().dlrm_interact_gather_ok(F, D))


# --- "one lookup per bag" proof for the fused lookup + interaction path -------------------------------------------------------
# nnz == B does not prove offsets == arange(B): EmbeddingBag accepts an empty bag next to a two-lookup bag (dlrm_s_pytorch.py:453-457)
# and the reference computes that input correctly.  The proof is one pass over the offsets on the device + ONE stream
# synchronisation, paid once per distinct offsets tensor: the verdict is cached on the tensor OBJECT (weak reference, so a recycled
# address can never alias) together with its in-place version counter.
_iota_cache: dict = {}          # id(tensor) -> (weakref, _version, verdict)
IOTA_STATS = {"checked": 0, "cached": 0, "tagged": 0, "device_predicates": 0, "host_us": 0.0, "wait_us": 0.0}
_IOTA_TAG = "_dlrm_one_lookup_per_bag"      # attribute a PRODUCER sets on an offsets tensor it wrote as 0, 1, ..., B-1 (value: t._version)


def mark_one_lookup_per_bag(t: torch.Tensor) -> torch.Tensor:
    """Producer-side proof: whoever WROTE the bag starts as 0, 1, ..., B-1 (dlrm_amd.datagen with one fixed lookup per bag,
    CriteoBinBatches, Multihot over all-ones hot sizes — by construction of their kernels) tags the tensor object, and
    `offsets_are_iota` then needs neither a device pass nor a synchronisation for it.  The tag holds the tensor's in-place version
    counter, so a later versioned write voids it; views and copies are new objects and carry no tag (they take the device proof) — and
    because Python attributes DO travel with copy.deepcopy / pickle, the tag also names the object and the storage address it was given
    for: a deep copy or an unpickled tensor is a different object at a different address and is therefore untagged.  What no tag
    can see is a write that bypasses the version counter (`.data` writes, a foreign kernel): producers tag tensors their own kernel has just
    written and hand them over; the fused kernels still verify every bag start they use and report a violation through the error block."""
    setattr(t, _IOTA_TAG, (t._version, id(t), t.data_ptr()))
    return t


def _iota_tagged(t: torch.Tensor) -> bool:
    return getattr(t, _IOTA_TAG, None) == (t._version, id(t), t.data_ptr())


def _iota_cached(t: torch.Tensor):
    e = _iota_cache.get(id(t))
    if e is not None and e[0]() is t and e[1] == t._version:
        return e[2]
    return None


def _iota_remember(t: torch.Tensor, verdict: bool) -> None:
    import weakref
    if len(_iota_cache) > 256:
        for k in [k for k, e in _iota_cache.items() if e[0]() is None]:
            del _iota_cache[k]
        if len(_iota_cache) > 256:
            _iota_cache.clear()
    _iota_cache[id(t)] = (weakref.ref(t), t._version, verdict)


class _IotaProof:
    """a proof in flight: the check kernel runs on the proof stream behind `entry` (an event of the caller's stream recorded when
    the proof was requested); `done` fires when the verdict is in `flag` (pinned host memory the kernel counts violations into)"""
    __slots__ = ("srcs", "keep", "flag", "done", "t0")


_proof_streams: dict = {}        # device index -> the (high-priority) stream the check kernels run on
_iota_flag_pool: dict = {}       # device index -> list of free pinned int32[1] flags


def offsets_are_iota_start(lS_o):
    """First half of the proof.  Returns True / False when the verdict is already known (producer tag, cached per tensor object), None
    while a HIP graph is being captured (undecided: GraphedTrainStep proves every incoming batch before the replay), else a handle for
    `offsets_are_iota_finish`.  The check kernel is NOT put on the caller's stream: it runs on a proof stream that waits only for what
    the caller's stream holds at this moment (the tensor's producer, by torch's stream convention), so everything the caller enqueues
    AFTER this call — DLRM_Net runs the whole bottom tower here — is queued behind nothing of the proof and keeps the GPU busy while
    the host waits for the verdict."""
    srcs = [lS_o] if isinstance(lS_o, torch.Tensor) else list(lS_o)
    if all(_iota_tagged(t) for t in srcs):
        IOTA_STATS["tagged"] += 1
        return True
    verdicts = [True if _iota_tagged(t) else _iota_cached(t) for t in srcs]
    if all(v is not None for v in verdicts):
        IOTA_STATS["cached"] += 1
        return all(verdicts)
    dev = srcs[0].device
    if torch.cuda.is_current_stream_capturing():
        return None
    import time as _time
    h = _IotaProof()
    h.t0 = _time.perf_counter()
    ptrs, keep = [], []
    for t in srcs:
        if not t.is_cuda or t.dtype not in (torch.int64, torch.int32) or t.dtype != srcs[0].dtype:
            raise RuntimeError("dlrm_amd: offsets must be int32/int64 GPU tensors of one dtype")
        if t.dim() == 2:
            if t.stride(1) != 1 and t.size(1) > 1:
                t = t.contiguous()
            ptrs += [t.data_ptr() + k * t.stride(0) * t.element_size() for k in range(t.size(0))]
        else:
            t = t.contiguous()
            ptrs.append(t.data_ptr())
        keep.append(t)
    B = srcs[0].size(-1)
    pool = _iota_flag_pool.setdefault(dev.index, [])
    flag = pool.pop() if pool else torch.zeros(1, dtype=torch.int32).pin_memory()
    flag[0] = 0
    cur = torch.cuda.current_stream(dev)
    ps = _proof_streams.get(dev.index)
    if ps is None:
        ps = _proof_streams[dev.index] = torch.cuda.Stream(device=dev, priority=-1)
    ps.wait_event(cur.record_event())
    rc = _lib.load().dlrm_offsets_are_iota(len(ptrs), B, _lib.ptr_array(ptrs), 64 if srcs[0].dtype == torch.int64 else 32,
                                           C.c_void_p(flag.data_ptr()), C.c_void_p(ps.cuda_stream))
    _lib.check(rc, "dlrm_offsets_are_iota")
    h.srcs, h.keep, h.flag, h.done = srcs, keep, flag, ps.record_event()
    IOTA_STATS["host_us"] += (_time.perf_counter() - h.t0) * 1e6
    return h


# Host wait for an event that ends the host's run-ahead once per step: POLL it (hipEventQuery) instead of sleeping on it.
# hipEventSynchronize blocks the thread in the driver, and how long the wake-up takes after the GPU signalled is the BOX's business
# (interrupt routing, CPU idle states, a runtime that polls with sleeps): measured 30-110 us per step on most MI355X boxes of the pool and
# 1.25 ms on one (profiles/round6/proof_wait.md: the same step 9.14 ms instead of 7.89, the GPU idle while the host slept — the verdict of the
# proof was there, the bottom tower long finished).  The wait is at most one training step long; after SPIN_WAIT_S the thread sleeps after all.
SPIN_WAIT_S = 0.0 if os.environ.get("DLRM_SPIN_WAIT", "1") == "0" else 0.05        # DLRM_SPIN_WAIT=0: sleep on the event (A/B)


def _wait_event_spinning(ev, limit_s: Optional[float] = None) -> None:
    """ev: a torch.cuda.Event or a torch.cuda.Stream (both have query() / synchronize()); limit_s: how long to poll before sleeping after all"""
    import time as _time
    if ev.query():
        return
    end = _time.perf_counter() + (SPIN_WAIT_S if limit_s is None else (limit_s if SPIN_WAIT_S > 0 else 0.0))
    while not ev.query():
        if _time.perf_counter() > end:
            ev.synchronize()
            return


wait_spinning = _wait_event_spinning


# ---- the verdict left on the DEVICE (ABI 16): no host wait at all ---------------------------------------------------------------------------
_FLAG_POOL = 1024
_iota_flag_slots: dict = {}     # device index -> [device int32 pool, pinned int32 pool, next slot]: a slot is used ONCE (zeroed at allocation, never re-armed)
_iota_unresolved: list = []     # (event, pinned slot, [weak references to the tensor objects]): proofs whose host-visible verdict has not been looked at yet


def _iota_drain() -> None:
    """remember the verdicts of finished device proofs (event.query(): no wait) — a tensor object that comes back is then known"""
    keep = []
    for ev, host, refs in _iota_unresolved:
        if ev.query():
            ok = int(host[0]) == 0
            for r in refs:                     # (weak references: the proof keeps no offsets tensor alive; an object that is gone needs no verdict)
                t = r()
                if t is not None:
                    _iota_remember(t, ok)
        else:
            keep.append((ev, host, refs))
    _iota_unresolved[:] = keep[-64:]           # (bounded: a verdict nobody came back for within 64 steps is dropped — the next encounter proves again)


def offsets_iota_state(lS_o):
    """What the caller of the fused lookup + interaction path needs to know about `lS_o`, WITHOUT waiting for the device:
      True / False    the verdict is known (producer tag, or this tensor object was proven earlier);
      None            a HIP graph is being captured (undecided: GraphedTrainStep proves every incoming batch before the replay);
      a device int32  the proof was enqueued on the CURRENT stream (dlrm_offsets_iota_flags: number of bags whose start differs from their
                      number) — the caller enqueues both implementations with that launch predicate (GatherInteractFunction) and never
                      waits; the host-visible copy of the verdict is looked at whenever a later call finds its event complete."""
    srcs = [lS_o] if isinstance(lS_o, torch.Tensor) else list(lS_o)
    _iota_drain()
    if all(_iota_tagged(t) for t in srcs):
        IOTA_STATS["tagged"] += 1
        return True
    verdicts = [True if _iota_tagged(t) else _iota_cached(t) for t in srcs]
    if all(v is not None for v in verdicts):
        IOTA_STATS["cached"] += 1
        return all(verdicts)
    dev = srcs[0].device
    if torch.cuda.is_current_stream_capturing():
        return None
    import time as _time
    t0 = _time.perf_counter()
    ptrs, keep = [], []
    for t in srcs:
        if not t.is_cuda or t.dtype not in (torch.int64, torch.int32) or t.dtype != srcs[0].dtype:
            raise RuntimeError("dlrm_amd: offsets must be int32/int64 GPU tensors of one dtype")
        if t.dim() == 2:
            if t.stride(1) != 1 and t.size(1) > 1:
                t = t.contiguous()
            ptrs += [t.data_ptr() + k * t.stride(0) * t.element_size() for k in range(t.size(0))]
        else:
            t = t.contiguous()
            ptrs.append(t.data_ptr())
        keep.append(t)
    slots = _iota_flag_slots.get(dev.index)
    if slots is None or slots[2] >= _FLAG_POOL:
        slots = _iota_flag_slots[dev.index] = [torch.zeros(_FLAG_POOL, dtype=torch.int32, device=dev),
                                               torch.zeros(_FLAG_POOL, dtype=torch.int32).pin_memory(), 0]
    k = slots[2]
    slots[2] = k + 1
    flag, host = slots[0][k:k + 1], slots[1][k:k + 1]
    with _timed("iota_proof"):
        rc = _lib.load().dlrm_offsets_iota_flags(len(ptrs), srcs[0].size(-1), _lib.ptr_array(ptrs), 64 if srcs[0].dtype == torch.int64 else 32,
                                                 C.c_void_p(flag.data_ptr()), C.c_void_p(host.data_ptr()), _stream(srcs[0]))
    _lib.check(rc, "dlrm_offsets_iota_flags")
    import weakref
    _iota_unresolved.append((torch.cuda.current_stream(dev).record_event(), host, [weakref.ref(t) for t in srcs]))
    IOTA_STATS["device_predicates"] = IOTA_STATS.get("device_predicates", 0) + 1
    IOTA_STATS["host_us"] += (_time.perf_counter() - t0) * 1e6
    del keep
    return flag


def offsets_are_iota_finish(h) -> bool:
    """Second half: wait for the check kernel alone (an event of the proof stream, not a stream synchronisation of the caller's) and
    read the verdict.  The offsets tensors stayed alive in the handle until here."""
    if not isinstance(h, _IotaProof):
        return h
    import time as _time
    t0 = _time.perf_counter()
    _wait_event_spinning(h.done)
    ok = int(h.flag[0]) == 0
    _iota_flag_pool[h.srcs[0].device.index].append(h.flag)
    IOTA_STATS["checked"] += 1
    # one verdict for the whole set: each tensor of a list is remembered with it (a False verdict of the set is re-examined only if
    # the same objects come back, and then it is False again)
    for t in h.srcs:
        _iota_remember(t, ok)
    IOTA_STATS["wait_us"] += (_time.perf_counter() - t0) * 1e6
    h.keep = h.srcs = None
    return ok


def offsets_are_iota(lS_o):
    """True iff every table's bag starts are 0, 1, ..., B-1 (with nnz == B: exactly one lookup per bag).  `lS_o` is what the caller
    passed to the module (a stacked [T, B] tensor or a list of [B] tensors).  Free for tensors their PRODUCER tagged
    (`mark_one_lookup_per_bag`: dlrm_amd.datagen, CriteoBinBatches, Multihot know it by construction) and for tensor objects seen
    before (verdict cached per object + in-place version; a write that does not bump `_version` — `t.data.copy_`, a collective or a
    custom kernel writing into a reused buffer — is not noticed here: the fused kernels still verify every bag start themselves and
    report a violation through the index-error block, so reused offsets buffers must be updated through versioned in-place ops).
    Any other tensor takes one device pass + a host wait for it (`offsets_are_iota_start` / `_finish`).  While a HIP graph is being
    captured no wait is possible: returns None (undecided) — GraphedTrainStep proves the incoming batch before every replay."""
    return offsets_are_iota_finish(offsets_are_iota_start(lS_o))


def _gather_desc(x: torch.Tensor, weights: Sequence[torch.Tensor], bags: BagBatch, D: int):
    """feature 0 = the [B, D] block x, features 1..T = the tables addressed through the bags' indices"""
    if bags.T != len(weights) or any(n != bags.B for n in bags.nnz):
        raise RuntimeError("dlrm_amd: the fused embedding + interaction path needs exactly one lookup per bag")
    if bags._psw is not None:
        raise RuntimeError("dlrm_amd: the fused embedding + interaction path does not take per-sample weights")
    _req(x, "x", ndim=2)
    ptrs = [x.data_ptr()] + [w.data_ptr() for w in weights]
    lds = [_ld(x)] + [D] * len(weights)
    F = len(ptrs)
    gidx = (C.c_void_p * F)(None, *[C.c_void_p(bags._idx[k]) for k in range(bags.T)])
    goff = (C.c_void_p * F)(None, *[C.c_void_p(bags._off[k]) for k in range(bags.T)])
    rows = _lib.i64_array([0] + [w.size(0) for w in weights])
    return F, _lib.ptr_array(ptrs), _lib.i64_array(lds), gidx, goff, rows


def interact_fwd_gather(x: torch.Tensor, weights: Sequence[torch.Tensor], bags: BagBatch, D: int, self_interaction: bool,
                        R: torch.Tensor, pred=None) -> torch.Tensor:
    """R = interaction of [x | one-hot embedding rows], the rows fetched by the kernel itself (no pooled-embedding buffer)."""
    lib = _lib.load()
    F, p, ld, gidx, goff, rows = _gather_desc(x, weights, bags, D)
    _req(R, "R", ndim=2)
    if R.size(0) != bags.B or x.size(0) != bags.B or R.size(1) < interact_out_width(F, D, self_interaction):
        raise RuntimeError("dlrm_amd: interact_fwd_gather shape mismatch")
    with _timed("emb_interact_fwd"):
        if pred is None:
            rc = lib.dlrm_interact_fwd_gather(bags.B, F, D, p, ld, gidx, goff, rows, bags.idx_bits, int(self_interaction),
                                              C.c_void_p(R.data_ptr()), _ld(R), C.c_void_p(_err_block(R.device).data_ptr()), _stream(R))
        else:
            rc = lib.dlrm_interact_fwd_pred(bags.B, F, D, p, ld, gidx, goff, rows, bags.idx_bits, int(self_interaction),
                                            C.c_void_p(R.data_ptr()), _ld(R), C.c_void_p(_err_block(R.device).data_ptr()),
                                            *_pred_args(pred), _stream(R))
    _lib.check(rc, "dlrm_interact_fwd_gather")
    return R


INTERACT_RELU_X = 4         # DLRM_INTERACT_RELU_X of include/dlrm_hip.h: OR-ed into the backward kernels' interaction mode


def interact_bwd_gather(x: torch.Tensor, weights: Sequence[torch.Tensor], bags: BagBatch, D: int, self_interaction: bool,
                        dR: torch.Tensor, dx: torch.Tensor, dE: torch.Tensor, pred=None, presorted: Optional["Presorted"] = None) -> None:
    """dx [B, D] = gradient of x; dE [B, T*D] = gradients of the T gathered rows (the dout of the fused embedding update).
    `self_interaction | INTERACT_RELU_X`: x is a ReLU output and dx comes back multiplied by [x > 0] (see interact_bwd).
    presorted (emb_presort): the sparse SGD step of the single lookups is taken here — their tables rows are UPDATED, their dE rows not
    written; emb_bwd_sgd_presorted applies the rest (dlrm_interact_bwd_gather_sgd, ABI 17)."""
    lib = _lib.load()
    F, p, ld, gidx, goff, rows = _gather_desc(x, weights, bags, D)
    _req(dR, "dR", ndim=2); _req(dx, "dx", ndim=2); _req(dE, "dE", ndim=2)
    dptrs = [dx.data_ptr()] + [dE.data_ptr() + 4 * k * D for k in range(bags.T)]
    dlds = [_ld(dx)] + [_ld(dE)] * bags.T
    with _timed("emb_interact_bwd"):
        if presorted is not None:
            lr_v, lr_p = _lr_args(presorted.lr)
            flag, nz = _pred_args(pred) if pred is not None else (None, 0)
            rc = lib.dlrm_interact_bwd_gather_sgd(bags.B, F, D, p, ld, gidx, goff, rows, bags.idx_bits, int(self_interaction),
                                                  C.c_void_p(dR.data_ptr()), _ld(dR), _lib.ptr_array(dptrs), _lib.i64_array(dlds),
                                                  C.c_void_p(presorted.mask.data_ptr()), lr_v, lr_p,
                                                  C.c_void_p(_err_block(dR.device).data_ptr()), flag, nz, _stream(dR))
        elif pred is None:
            rc = lib.dlrm_interact_bwd_gather(bags.B, F, D, p, ld, gidx, goff, rows, bags.idx_bits, int(self_interaction),
                                              C.c_void_p(dR.data_ptr()), _ld(dR), _lib.ptr_array(dptrs), _lib.i64_array(dlds),
                                              C.c_void_p(_err_block(dR.device).data_ptr()), _stream(dR))
        else:
            rc = lib.dlrm_interact_bwd_pred(bags.B, F, D, p, ld, gidx, goff, rows, bags.idx_bits, int(self_interaction),
                                            C.c_void_p(dR.data_ptr()), _ld(dR), _lib.ptr_array(dptrs), _lib.i64_array(dlds),
                                            C.c_void_p(_err_block(dR.device).data_ptr()), *_pred_args(pred), _stream(dR))
    _lib.check(rc, "dlrm_interact_bwd_gather")


def interact_bwd(blocks: Sequence[torch.Tensor], D: int, self_interaction: bool, dR: torch.Tensor,
                 dblocks: Sequence[torch.Tensor], order=None, pred=None) -> None:
    """`self_interaction` is the forward's mode, optionally OR-ed with INTERACT_RELU_X: feature 0 (the first D columns of blocks[0],
    the bottom tower's output) is then taken as a ReLU output and its gradient is multiplied by the derivative [feature 0 > 0] inside
    the kernel, which has the feature staged anyway — the tower's backward then starts without its act_bwd pass over [B, D]."""
    lib = _lib.load()
    B, ptrs, lds = _feature_table(blocks, D)
    B2, dptrs, dlds = _feature_table(dblocks, D)
    ptrs, lds, dptrs, dlds = _permute(order, ptrs, lds, dptrs, dlds)
    _req(dR, "dR", ndim=2)
    if B != B2 or len(ptrs) != len(dptrs) or dR.size(0) != B:
        raise RuntimeError("dlrm_amd: interact_bwd shape mismatch")
    F = len(ptrs)
    with _timed("interact_bwd" if pred is None else "emb_interact_bwd"):
        if pred is None:
            rc = lib.dlrm_interact_bwd(B, F, D, _lib.ptr_array(ptrs), _lib.i64_array(lds), int(self_interaction),
                                       C.c_void_p(dR.data_ptr()), _ld(dR), _lib.ptr_array(dptrs), _lib.i64_array(dlds),
                                       _stream(dR))
        else:
            rc = lib.dlrm_interact_bwd_pred(B, F, D, _lib.ptr_array(ptrs), _lib.i64_array(lds), None, None, None, 64, int(self_interaction),
                                            C.c_void_p(dR.data_ptr()), _ld(dR), _lib.ptr_array(dptrs), _lib.i64_array(dlds), None,
                                            *_pred_args(pred), _stream(dR))
    _lib.check(rc, "dlrm_interact_bwd")


# ------------------------------------------------------------------------------------------------
# MLP layers
# ------------------------------------------------------------------------------------------------
ARITH_NAMES = {"f32": _lib.ARITH_F32, "bf16x6": _lib.ARITH_BF16X6, "bf16": _lib.ARITH_BF16}


def arith_code(arith) -> int:
    """MLP arithmetic as the C ABI's per-call argument.  "f32": native fp32 MFMA.  "bf16x6": fp32 operands split exactly into
    3 bf16 terms, 6 bf16 MFMA products, fp32 accumulation (fp32 round-off class, 2.7x the matrix rate).  "bf16": operands
    rounded to bf16, one bf16 MFMA per 16-k step, fp32 accumulation — the reduced-precision "bf16 MLP" of BASELINE.json
    configs[4] (NOT fp32-class).  There is no process-wide switch: the arithmetic belongs to the module (FusedMLP.arith)
    and travels with every call.  See include/dlrm_hip.h."""
    if isinstance(arith, int) and arith in ARITH_NAMES.values():
        return arith
    if arith not in ARITH_NAMES:
        raise RuntimeError(f"dlrm_amd: unknown MLP arithmetic {arith!r} (f32 | bf16x6 | bf16)")
    return ARITH_NAMES[arith]


def round_bf16_k(K: int) -> int:
    """reduction length a bf16 operand copy is padded to: a multiple of 64 from 64 up (the bf16-shaped kernel's k-tile), of 32 below"""
    return (K + 63) & ~63 if K >= 64 else (K + 31) & ~31


def relu_bits_alloc(M: int, N: int, device) -> torch.Tensor:
    """buffer for the ReLU sign bits of an [M, N] activation (one bit per element, include/dlrm_hip.h)"""
    return torch.empty(_lib.load().dlrm_relu_bits_bytes(M, N) // 8, dtype=torch.int64, device=device)


def linear_fwd(X: torch.Tensor, W: torch.Tensor, bias: Optional[torch.Tensor], act: int, Y: torch.Tensor,
               arith=_lib.ARITH_F32, relu_bits: Optional[torch.Tensor] = None) -> torch.Tensor:
    lib = _lib.load()
    _req(X, "X", ndim=2); _req(W, "W", ndim=2); _req(Y, "Y", ndim=2)
    M, K = X.shape
    N = W.size(0)
    if W.size(1) != K or Y.size(0) != M or Y.size(1) != N:
        raise RuntimeError(f"dlrm_amd: linear_fwd shape mismatch X{tuple(X.shape)} W{tuple(W.shape)} Y{tuple(Y.shape)}")
    if bias is not None:
        _req(bias, "bias", ndim=1)
    with _timed("linear_fwd"):
        rc = lib.dlrm_linear_fwd(M, N, K, C.c_void_p(X.data_ptr()), _ld(X), C.c_void_p(W.data_ptr()), _ld(W),
                                 C.c_void_p(bias.data_ptr()) if bias is not None else None, int(act),
                                 C.c_void_p(Y.data_ptr()), _ld(Y),
                                 C.c_void_p(relu_bits.data_ptr()) if relu_bits is not None else None, arith_code(arith), _stream(Y))
    _lib.check(rc, "dlrm_linear_fwd")
    return Y


def linear_bwd_data(dY: torch.Tensor, W: torch.Tensor, Xact: Optional[torch.Tensor], xact_kind: int,
                    dX: torch.Tensor, arith=_lib.ARITH_F32, relu_bits: Optional[torch.Tensor] = None) -> torch.Tensor:
    """dX = (dY @ W) * act'(Xact)   (Xact None -> no mask; relu_bits = the sign bits linear_fwd stored for Xact)"""
    lib = _lib.load()
    _req(dY, "dY", ndim=2); _req(W, "W", ndim=2); _req(dX, "dX", ndim=2)
    M, N = dY.shape
    K = W.size(1)
    if W.size(0) != N or dX.size(0) != M or dX.size(1) != K:
        raise RuntimeError("dlrm_amd: linear_bwd_data shape mismatch")
    if Xact is not None:
        _req(Xact, "Xact", ndim=2)
    elif xact_kind != ACT_NONE:
        # (the sign bits alone are not enough here: only the fast GEMM path reads them, every other path of dlrm_linear_bwd_data masks
        # with the fp32 activation — a caller that kept bits only must take dlrm_gemm_bf16's relu_bits_in, functional.MLPFunction)
        raise RuntimeError("dlrm_amd: linear_bwd_data with an activation derivative needs the fp32 activation Xact "
                           "(got xact_kind=%d, Xact=None%s)" % (xact_kind, ", relu_bits given" if relu_bits is not None else ""))
    with _timed("linear_bwd_data"):
        rc = lib.dlrm_linear_bwd_data(M, N, K, C.c_void_p(dY.data_ptr()), _ld(dY), C.c_void_p(W.data_ptr()), _ld(W),
                                      C.c_void_p(Xact.data_ptr()) if Xact is not None else None,
                                      _ld(Xact) if Xact is not None else 0, int(xact_kind if Xact is not None else ACT_NONE),
                                      C.c_void_p(relu_bits.data_ptr()) if (relu_bits is not None and Xact is not None) else None,
                                      C.c_void_p(dX.data_ptr()), _ld(dX), arith_code(arith), _stream(dX))
    _lib.check(rc, "dlrm_linear_bwd_data")
    return dX


# ---- bf16-storage tower (dlrm_cast_bf16 / dlrm_gemm_bf16, include/dlrm_hip.h) -------------------------------------------------
def round32(n: int) -> int:
    return (n + 31) & ~31


def cast_bf16(src: torch.Tensor, Npad: Optional[int] = None, category: str = "cast_bf16") -> torch.Tensor:
    """[M, N] fp32 -> contiguous [M, Npad] bf16 (round to nearest even; zero columns N..Npad-1)"""
    lib = _lib.load()
    _req(src, "src", ndim=2)
    M, N = src.shape
    Npad = N + (N & 1) if Npad is None else int(Npad)
    dst = torch.empty((M, Npad), dtype=torch.bfloat16, device=src.device)
    with _timed(category):
        rc = lib.dlrm_cast_bf16(M, N, Npad, C.c_void_p(src.data_ptr()), _ld(src), C.c_void_p(dst.data_ptr()), Npad, _stream(dst))
    _lib.check(rc, "dlrm_cast_bf16")
    return dst


def cast_bf16_transposed(src: torch.Tensor, Rpad: Optional[int] = None, category: str = "cast_bf16") -> torch.Tensor:
    """[R, C] fp32 -> contiguous [C, Rpad] bf16 = its transpose (zero columns R..Rpad-1): W^T for the data-gradient GEMM"""
    lib = _lib.load()
    _req(src, "src", ndim=2)
    R, Cc = src.shape
    Rpad = R if Rpad is None else int(Rpad)
    dst = torch.empty((Cc, Rpad), dtype=torch.bfloat16, device=src.device)
    with _timed(category):
        rc = lib.dlrm_cast_bf16_transposed(R, Cc, Rpad, C.c_void_p(src.data_ptr()), _ld(src), C.c_void_p(dst.data_ptr()), Rpad, _stream(dst))
    _lib.check(rc, "dlrm_cast_bf16_transposed")
    return dst


CAST_MULTI_MAX = 16


def cast_bf16_multi(items, category: str = "cast_bf16"):
    """bf16 copies of several fp32 matrices in ONE launch (dlrm_cast_bf16_multi).  items: (src [R, C], Cpad or None, Rpad or None) —
    Cpad: also return the row-major copy [R, Cpad] (zero columns C..); Rpad: also return the transposed copy [C, Rpad] (zero columns R..).
    Returns a list of (copy or None, transposed copy or None)."""
    lib = _lib.load()
    out = []
    for k0 in range(0, len(items), CAST_MULTI_MAX):
        chunk = items[k0:k0 + CAST_MULTI_MAX]
        n = len(chunk)
        srcs, lds, R, Cc, dst, ldd, cpad, dstT, lddT, rpad, res = [], [], [], [], [], [], [], [], [], [], []
        for src, Cpad, Rpad in chunk:
            _req(src, "src", ndim=2)
            r_, c_ = src.shape
            d = torch.empty((r_, int(Cpad)), dtype=torch.bfloat16, device=src.device) if Cpad else None
            dT = torch.empty((c_, int(Rpad)), dtype=torch.bfloat16, device=src.device) if Rpad else None
            srcs.append(src.data_ptr()); lds.append(_ld(src)); R.append(r_); Cc.append(c_)
            dst.append(d.data_ptr() if d is not None else 0); ldd.append(int(Cpad) if Cpad else 0); cpad.append(int(Cpad) if Cpad else 0)
            dstT.append(dT.data_ptr() if dT is not None else 0); lddT.append(int(Rpad) if Rpad else 0); rpad.append(int(Rpad) if Rpad else 0)
            res.append((d, dT))
        ia = lambda v: (C.c_int * n)(*v)            # noqa: E731
        with _timed(category):
            rc = lib.dlrm_cast_bf16_multi(n, _lib.ptr_array(srcs), _lib.i64_array(lds), ia(R), ia(Cc), _lib.ptr_array(dst), _lib.i64_array(ldd), ia(cpad),
                                          _lib.ptr_array(dstT), _lib.i64_array(lddT), ia(rpad), _stream(chunk[0][0]))
        _lib.check(rc, "dlrm_cast_bf16_multi")
        out += res
    return out


def gemm_bf16(A: torch.Tensor, B: torch.Tensor, bias: Optional[torch.Tensor], act: int, Cf: Optional[torch.Tensor],
              Cb: Optional[torch.Tensor], relu_bits_out: Optional[torch.Tensor] = None, relu_bits_in: Optional[torch.Tensor] = None,
              category: str = "linear_fwd", addend: Optional[torch.Tensor] = None, addend2: Optional[torch.Tensor] = None) -> None:
    """Cf (fp32) and/or Cb (bf16) [M, N] = epilogue(A[M, K] @ B[N, K]^T) with bf16 operands in memory (dlrm_gemm_bf16)."""
    lib = _lib.load()
    for t, name in ((A, "A"), (B, "B")):
        if t.dtype != torch.bfloat16 or not t.is_cuda or t.dim() != 2 or t.stride(1) != 1:
            raise RuntimeError(f"dlrm_amd: gemm_bf16 operand {name} must be a 2-D bf16 GPU tensor with contiguous rows")
    M, K = A.shape
    N = B.size(0)
    if B.size(1) != K:
        raise RuntimeError("dlrm_amd: gemm_bf16 reduction lengths differ")
    if Cf is not None:
        _req(Cf, "C", ndim=2)
    if Cb is not None and (Cb.dtype != torch.bfloat16 or tuple(Cb.shape) != (M, N) or Cb.stride(1) != 1):
        raise RuntimeError("dlrm_amd: gemm_bf16 bf16 output must be [M, N] bf16 with contiguous rows")
    out = Cf if Cf is not None else Cb
    with _timed(category):
        rc = lib.dlrm_gemm_bf16(M, N, K, C.c_void_p(A.data_ptr()), A.stride(0), C.c_void_p(B.data_ptr()), B.stride(0),
                                C.c_void_p(bias.data_ptr()) if bias is not None else None, int(act),
                                C.c_void_p(relu_bits_out.data_ptr()) if relu_bits_out is not None else None,
                                C.c_void_p(relu_bits_in.data_ptr()) if relu_bits_in is not None else None,
                                C.c_void_p(addend.data_ptr()) if addend is not None else None, _ld(addend) if addend is not None else 0,
                                C.c_void_p(addend2.data_ptr()) if addend2 is not None else None, _ld(addend2) if addend2 is not None else 0,
                                C.c_void_p(Cf.data_ptr()) if Cf is not None else None, _ld(Cf) if Cf is not None else 0,
                                C.c_void_p(Cb.data_ptr()) if Cb is not None else None, Cb.stride(0) if Cb is not None else 0, _stream(out))
    _lib.check(rc, "dlrm_gemm_bf16")


# ---- arith "bf16x6" from pre-split operands: an fp32 tensor travels as a contiguous [3, rows, cols] bf16 tensor = its planes h, m, l ----

def _planes(t: torch.Tensor, name: str) -> None:
    if t.dtype != torch.bfloat16 or not t.is_cuda or t.dim() != 3 or t.size(0) != 3 or not t.is_contiguous():
        raise RuntimeError(f"dlrm_amd: {name} must be a contiguous [3, rows, cols] bf16 GPU tensor (the planes of an fp32 matrix)")


def round_x6_k(K: int) -> int:
    """reduction length an operand's planes are padded to: the 16-k tile of the planes kernel"""
    return (K + 15) & ~15


def gemm_bf16x6_ok(M: int, N: int, K: int) -> bool:
    """preconditions of dlrm_gemm_bf16x6 for contiguous planes [3, M, K] x [3, N, K]"""
    return bool(_lib.load().dlrm_gemm_bf16x6_supported(M, N, K, K, K)) and 3 * max(M, N) * K < (1 << 31)


def split_bf16x3(src: torch.Tensor, Npad: Optional[int] = None, category: str = "cast_bf16") -> torch.Tensor:
    """[M, N] fp32 -> [3, M, Npad] bf16: the truncation planes h, m, l with src == h + m + l exactly (zero columns N..Npad-1)"""
    lib = _lib.load()
    _req(src, "src", ndim=2)
    M, N = src.shape
    Npad = (N + 7) & ~7 if Npad is None else int(Npad)
    dst = torch.empty((3, M, Npad), dtype=torch.bfloat16, device=src.device)
    with _timed(category):
        rc = lib.dlrm_split_bf16x3(M, N, Npad, C.c_void_p(src.data_ptr()), _ld(src), C.c_void_p(dst.data_ptr()), Npad, M * Npad, _stream(dst))
    _lib.check(rc, "dlrm_split_bf16x3")
    return dst


def split_bf16x3_transposed(src: torch.Tensor, Rpad: Optional[int] = None, category: str = "cast_bf16") -> torch.Tensor:
    """[R, C] fp32 -> [3, C, Rpad] bf16: the planes of its transpose (zero columns R..Rpad-1)"""
    lib = _lib.load()
    _req(src, "src", ndim=2)
    R, Cc = src.shape
    Rpad = R if Rpad is None else int(Rpad)
    dst = torch.empty((3, Cc, Rpad), dtype=torch.bfloat16, device=src.device)
    with _timed(category):
        rc = lib.dlrm_split_bf16x3_transposed(R, Cc, Rpad, C.c_void_p(src.data_ptr()), _ld(src), C.c_void_p(dst.data_ptr()), Rpad, Cc * Rpad, _stream(dst))
    _lib.check(rc, "dlrm_split_bf16x3_transposed")
    return dst


def gemm_bf16x6(A: torch.Tensor, B: torch.Tensor, bias: Optional[torch.Tensor], act: int, Cf: Optional[torch.Tensor],
                Cp: Optional[torch.Tensor], relu_bits_out: Optional[torch.Tensor] = None, relu_bits_in: Optional[torch.Tensor] = None,
                category: str = "linear_fwd") -> None:
    """Cf (fp32 [M, N]) and/or Cp (planes [3, M, N]) = epilogue(A @ B^T) for operands given as planes A [3, M, K], B [3, N, K] (dlrm_gemm_bf16x6)."""
    lib = _lib.load()
    _planes(A, "A"); _planes(B, "B")
    _, M, K = A.shape
    N = B.size(1)
    if B.size(2) != K:
        raise RuntimeError("dlrm_amd: gemm_bf16x6 reduction lengths differ")
    if Cf is not None:
        _req(Cf, "C", ndim=2)
    if Cp is not None:
        _planes(Cp, "Cp")
        if tuple(Cp.shape) != (3, M, N):
            raise RuntimeError("dlrm_amd: gemm_bf16x6 planes output must be [3, M, N]")
    out = Cf if Cf is not None else Cp
    with _timed(category):
        rc = lib.dlrm_gemm_bf16x6(M, N, K, C.c_void_p(A.data_ptr()), K, M * K, C.c_void_p(B.data_ptr()), K, N * K,
                                  C.c_void_p(bias.data_ptr()) if bias is not None else None, int(act),
                                  C.c_void_p(relu_bits_out.data_ptr()) if relu_bits_out is not None else None,
                                  C.c_void_p(relu_bits_in.data_ptr()) if relu_bits_in is not None else None,
                                  C.c_void_p(Cf.data_ptr()) if Cf is not None else None, _ld(Cf) if Cf is not None else 0,
                                  C.c_void_p(Cp.data_ptr()) if Cp is not None else None, N if Cp is not None else 0, M * N if Cp is not None else 0,
                                  _stream(out))
    _lib.check(rc, "dlrm_gemm_bf16x6")


def linear_bwd_weight_bf16x6(dZ3: torch.Tensor, X3: torch.Tensor, dW: torch.Tensor, dbias: Optional[torch.Tensor] = None,
                             accumulate: bool = False) -> torch.Tensor:
    """dW [N, K_store] fp32 = dZ^T @ X[:, :K_store], dbias = column sums of dZ, from the planes dZ3 [3, M, N], X3 [3, M, K] (read k-strided).
    X3's columns dW.size(1) .. round8(dW.size(1)) must be zero padding."""
    lib = _lib.load()
    _planes(dZ3, "dZ3"); _planes(X3, "X3")
    _req(dW, "dW", ndim=2)
    _, M, N = dZ3.shape
    Kx = X3.size(2)
    K_store = dW.size(1)
    K = (K_store + 7) & ~7
    if X3.size(1) != M or dW.size(0) != N or Kx < K:
        raise RuntimeError("dlrm_amd: linear_bwd_weight_bf16x6 shape mismatch")
    if dbias is not None:
        _req(dbias, "dbias", ndim=1)
    ws = _wgrad_workspace(lib.dlrm_linear_bwd_weight_bf16_workspace_bytes(M, N, K), dW.device)
    with _timed("linear_bwd_weight"):
        rc = lib.dlrm_linear_bwd_weight_bf16x6(M, N, K, K_store, C.c_void_p(dZ3.data_ptr()), N, M * N, C.c_void_p(X3.data_ptr()), Kx, M * Kx,
                                               C.c_void_p(dW.data_ptr()), _ld(dW), C.c_void_p(dbias.data_ptr()) if dbias is not None else None,
                                               int(bool(accumulate)), C.c_void_p(ws.data_ptr()), ws.numel(), _stream(dW))
    _lib.check(rc, "dlrm_linear_bwd_weight_bf16x6")
    return dW


_wgrad_ws = {}   # (device, stream) -> cached split-K workspace (kernels of one stream are ordered, so one slab set suffices)


def _wgrad_workspace(need: int, device) -> Optional[torch.Tensor]:
    if need <= 0:
        return None
    key = (device, torch.cuda.current_stream(device).cuda_stream)        # (the stream of the TENSORS' device: what the launch uses)
    ws = _wgrad_ws.get(key)
    if ws is None or ws.numel() < need:
        ws = torch.empty(int(need), dtype=torch.uint8, device=device)
        _wgrad_ws[key] = ws
    return ws


def linear_bwd_weight(dY: torch.Tensor, X: torch.Tensor, dW: torch.Tensor, dbias: Optional[torch.Tensor] = None,
                      accumulate: bool = False, use_workspace: bool = True, arith=_lib.ARITH_F32) -> torch.Tensor:
    """dW = dY^T @ X and (optionally) dbias = column sums of dY, one GEMM (+ the split-K slab reduction).
    dW may be NARROWER than X (dW [N, K_store], X [M, K >= K_store]): the extra columns of X are alignment padding
    (zeros) and their gradient is dropped.  use_workspace=False exercises the atomic-accumulation variant."""
    lib = _lib.load()
    _req(dY, "dY", ndim=2); _req(X, "X", ndim=2); _req(dW, "dW", ndim=2)
    M, N = dY.shape
    K = X.size(1)
    K_store = dW.size(1)
    if X.size(0) != M or dW.size(0) != N or K_store > K:
        raise RuntimeError("dlrm_amd: linear_bwd_weight shape mismatch")
    if dbias is not None:
        _req(dbias, "dbias", ndim=1)
        if dbias.numel() != N:
            raise RuntimeError("dlrm_amd: linear_bwd_weight dbias size mismatch")
    ws = _wgrad_workspace(lib.dlrm_linear_bwd_weight_workspace_bytes(M, N, K), dY.device) if use_workspace else None
    with _timed("linear_bwd_weight"):
        common = (C.c_void_p(dY.data_ptr()), _ld(dY), C.c_void_p(X.data_ptr()), _ld(X),
                  C.c_void_p(dW.data_ptr()), _ld(dW),
                  C.c_void_p(dbias.data_ptr()) if dbias is not None else None,
                  int(bool(accumulate)),
                  C.c_void_p(ws.data_ptr()) if ws is not None else None,
                  ws.numel() if ws is not None else 0, arith_code(arith), _stream(dW))
        if K_store == K:
            rc = lib.dlrm_linear_bwd_weight(M, N, K, *common)
        else:
            rc = lib.dlrm_linear_bwd_weight_padded(M, N, K, K_store, *common)
    _lib.check(rc, "dlrm_linear_bwd_weight")
    return dW


def linear_head_bwd(dY: torch.Tensor, Y: Optional[torch.Tensor], act: int, X: torch.Tensor, W: torch.Tensor, xact_kind: int,
                    dX: Optional[torch.Tensor], dW: torch.Tensor, dbias: Optional[torch.Tensor], accumulate: bool = False) -> bool:
    """The whole backward of an N == 1 layer in one pass over X (dlrm_linear_head_bwd): dz = dY * act'(Y), dW [1, K] = dz^T X, dbias [1] = sum dz,
    dX [M, K] = (dz W) * xact'(X) — the bits of act_bwd + linear_bwd_weight + linear_bwd_data.  False: the shape is outside the fast path and
    NOTHING was launched (the caller makes the three calls)."""
    lib = _lib.load()
    _req(dY, "dY", ndim=2); _req(X, "X", ndim=2); _req(W, "W", ndim=2); _req(dW, "dW", ndim=2)
    M, K = X.shape
    if dY.size(0) != M or dY.size(1) != 1 or W.shape != (1, K) or dW.shape != (1, K) or (dX is not None and dX.shape != (M, K)):
        raise RuntimeError("dlrm_amd: linear_head_bwd shape mismatch")
    if Y is not None:
        _req(Y, "Y", ndim=2)
    if dX is not None:
        _req(dX, "dX", ndim=2)
    ws = _wgrad_workspace(lib.dlrm_linear_bwd_weight_workspace_bytes(M, 1, K), dY.device)
    if ws is None:
        return False
    with _timed("linear_bwd_weight"):
        rc = lib.dlrm_linear_head_bwd(M, K, C.c_void_p(dY.data_ptr()), _ld(dY), C.c_void_p(Y.data_ptr()) if Y is not None else None,
                                      _ld(Y) if Y is not None else 1, int(act), C.c_void_p(X.data_ptr()), _ld(X), C.c_void_p(W.data_ptr()),
                                      int(xact_kind), C.c_void_p(dX.data_ptr()) if dX is not None else None, _ld(dX) if dX is not None else K,
                                      C.c_void_p(dW.data_ptr()), C.c_void_p(dbias.data_ptr()) if dbias is not None else None,
                                      int(bool(accumulate)), C.c_void_p(ws.data_ptr()), ws.numel(), _stream(dW))
    if rc == -4:                                     # DLRM_E_MODE: outside the fast path, nothing launched
        return False
    _lib.check(rc, "dlrm_linear_head_bwd")
    return True


def linear_bwd_weight_bf16_ok(M: int, N: int, K: int, dZ16: torch.Tensor, X16: torch.Tensor) -> bool:
    """preconditions of dlrm_linear_bwd_weight_bf16 (csrc/gemm_bf16.hip, weight-gradient form)"""
    K = (K + 7) & ~7
    return (M >= 256 and M % 64 == 0 and N % 8 == 0 and N >= 64 and K >= 64 and X16.size(1) >= K and dZ16.stride(0) % 8 == 0 and X16.stride(0) % 8 == 0
            and dZ16.data_ptr() % 16 == 0 and X16.data_ptr() % 16 == 0 and BF16_PHASED)


def linear_bwd_weight_bf16(dZ16: torch.Tensor, X16: torch.Tensor, dW: torch.Tensor, dbias: Optional[torch.Tensor] = None,
                           accumulate: bool = False) -> torch.Tensor:
    """dW [N, K] fp32 = dZ16[M, N]^T @ X16[M, :K] and dbias = column sums of dZ16, both operands bf16 AS STORED (read k-strided through
    ds_read_b64_tr_b16).  dW may be narrower than the product: X16's columns dW.size(1) .. round8(dW.size(1)) must then be ZERO padding."""
    lib = _lib.load()
    for t, name in ((dZ16, "dZ16"), (X16, "X16")):
        if t.dtype != torch.bfloat16 or not t.is_cuda or t.dim() != 2 or t.stride(1) != 1:
            raise RuntimeError(f"dlrm_amd: linear_bwd_weight_bf16 operand {name} must be a 2-D bf16 GPU tensor with contiguous rows")
    _req(dW, "dW", ndim=2)
    M, N = dZ16.shape
    K_store = dW.size(1)
    K = (K_store + 7) & ~7                       # the product's width; columns K_store.. of X16 are zero padding
    if X16.size(0) != M or dW.size(0) != N or X16.size(1) < K:
        raise RuntimeError("dlrm_amd: linear_bwd_weight_bf16 shape mismatch")
    if dbias is not None:
        _req(dbias, "dbias", ndim=1)
    ws = _wgrad_workspace(lib.dlrm_linear_bwd_weight_bf16_workspace_bytes(M, N, K), dW.device)
    with _timed("linear_bwd_weight"):
        rc = lib.dlrm_linear_bwd_weight_bf16(M, N, K, K_store, C.c_void_p(dZ16.data_ptr()), dZ16.stride(0), C.c_void_p(X16.data_ptr()), X16.stride(0),
                                             C.c_void_p(dW.data_ptr()), _ld(dW), C.c_void_p(dbias.data_ptr()) if dbias is not None else None,
                                             int(bool(accumulate)), C.c_void_p(ws.data_ptr()), ws.numel(), _stream(dW))
    _lib.check(rc, "dlrm_linear_bwd_weight_bf16")
    return dW


def pad_cols(src: torch.Tensor, Kp: int) -> torch.Tensor:
    """[M, K] -> new contiguous [M, Kp] with zero columns K..Kp-1 (one kernel; replaces torch.zeros + slice copy_)."""
    lib = _lib.load()
    _req(src, "src", ndim=2)
    M, K = src.shape
    if Kp < K:
        raise RuntimeError("dlrm_amd: pad_cols target width is smaller than the source")
    dst = torch.empty((M, Kp), dtype=torch.float32, device=src.device)
    if M == 0:
        return dst
    rc = lib.dlrm_pad_cols(M, K, Kp, C.c_void_p(src.data_ptr()), _ld(src), C.c_void_p(dst.data_ptr()), Kp, _stream())
    _lib.check(rc, "dlrm_pad_cols")
    return dst


# ------------------------------------------------------------------------------------------------
# small-batch towers (csrc/tower.hip): all layers of an MLP in one launch per direction
# ------------------------------------------------------------------------------------------------
TOWER_MAX_LAYERS, TOWER_MAX_WIDTH = 8, 512          # DLRM_TOWER_MAX_LAYERS / DLRM_TOWER_MAX_WIDTH of include/dlrm_hip.h
_tower_ws = {}     # (device, stream) -> slab workspace of tower_wgrad


def _int_array(vals):
    return (C.c_int * len(vals))(*[int(v) for v in vals])


def tower_ok(M: int, widths: Sequence[int]) -> bool:
    return 0 < M and 1 <= len(widths) - 1 <= TOWER_MAX_LAYERS and all(0 < w <= TOWER_MAX_WIDTH for w in widths)


def tower_fwd(x: torch.Tensor, weights: Sequence[torch.Tensor], biases: Sequence[Optional[torch.Tensor]], acts: Sequence[int],
              outs: Sequence[torch.Tensor]) -> None:
    """outs[l] = act_l(in_l @ weights[l]^T + biases[l]) for the whole tower, in_0 = x, in_l = outs[l-1]: ONE launch (dlrm_tower_fwd).
    weights[l] is [N_l, K_l] with K_0 == x.size(1) and K_l == N_{l-1}; every outs[l] [M, N_l] is written (backward reads them)."""
    lib = _lib.load()
    L = len(weights)
    _req(x, "x", ndim=2)
    M = x.size(0)
    widths = [x.size(1)] + [w.size(0) for w in weights]
    for l, (w, o) in enumerate(zip(weights, outs)):
        _req(w, "W", ndim=2); _req(o, "out", ndim=2)
        if w.size(1) != widths[l] or o.size(0) != M or o.size(1) != widths[l + 1] or (biases[l] is not None and biases[l].numel() != widths[l + 1]):
            raise RuntimeError("dlrm_amd: tower_fwd shape mismatch at layer %d" % l)
    if len(outs) != L or len(biases) != L or len(acts) != L or not tower_ok(M, widths):
        raise RuntimeError("dlrm_amd: tower_fwd: %d layers of widths %s are outside the small-batch tower kernels" % (L, widths))
    with _timed("linear_fwd"):
        rc = lib.dlrm_tower_fwd(M, L, _int_array(widths), _int_array(acts), C.c_void_p(x.data_ptr()), _ld(x),
                                _lib.ptr_array([w.data_ptr() for w in weights]), _lib.i64_array([_ld(w) for w in weights]),
                                _lib.ptr_array([b.data_ptr() if b is not None else 0 for b in biases]),
                                _lib.ptr_array([o.data_ptr() for o in outs]), _lib.i64_array([_ld(o) for o in outs]), _stream(x))
    _lib.check(rc, "dlrm_tower_fwd")


def tower_bwd(dY: torch.Tensor, weights: Sequence[torch.Tensor], acts: Sequence[int], outs: Sequence[torch.Tensor],
              dZs: Sequence[torch.Tensor], dX: Optional[torch.Tensor], last_act_applied: bool = False) -> None:
    """The data-gradient chain of the tower in ONE launch (dlrm_tower_bwd): dZs[l] [M, N_l] = dL/dz_l of every layer (the operands of
    tower_wgrad) and, when dX is given, the gradient of the tower's input [M, K_0]."""
    lib = _lib.load()
    L = len(weights)
    _req(dY, "dY", ndim=2)
    M = dY.size(0)
    widths = [weights[0].size(1)] + [w.size(0) for w in weights]
    if len(outs) != L or len(dZs) != L or len(acts) != L or dY.size(1) != widths[L] or not tower_ok(M, widths):
        raise RuntimeError("dlrm_amd: tower_bwd argument mismatch")
    for l in range(L):
        _req(outs[l], "out", ndim=2); _req(dZs[l], "dZ", ndim=2)
        if outs[l].shape != (M, widths[l + 1]) or dZs[l].shape != (M, widths[l + 1]) or weights[l].size(1) != widths[l]:
            raise RuntimeError("dlrm_amd: tower_bwd shape mismatch at layer %d" % l)
    if dX is not None:
        _req(dX, "dX", ndim=2)
        if dX.shape != (M, widths[0]):
            raise RuntimeError("dlrm_amd: tower_bwd dX shape mismatch")
    with _timed("linear_bwd_data"):
        rc = lib.dlrm_tower_bwd(M, L, _int_array(widths), _int_array(acts), C.c_void_p(dY.data_ptr()), _ld(dY), int(bool(last_act_applied)),
                                _lib.ptr_array([w.data_ptr() for w in weights]), _lib.i64_array([_ld(w) for w in weights]),
                                _lib.ptr_array([o.data_ptr() for o in outs]), _lib.i64_array([_ld(o) for o in outs]),
                                _lib.ptr_array([z.data_ptr() for z in dZs]), _lib.i64_array([_ld(z) for z in dZs]),
                                C.c_void_p(dX.data_ptr()) if dX is not None else None, _ld(dX) if dX is not None else 0, _stream(dY))
    _lib.check(rc, "dlrm_tower_bwd")


def tower_wgrad(dZs: Sequence[torch.Tensor], ins: Sequence[torch.Tensor], dWs: Sequence[torch.Tensor],
                dbs: Sequence[Optional[torch.Tensor]]) -> None:
    """dWs[l] [N_l, K_store_l] = dZs[l]^T @ ins[l] (K_store_l = dWs[l].size(1) <= ins[l].size(1): trailing padding columns of the input
    are dropped) and dbs[l] = column sums of dZs[l], for ALL layers in one launch, deterministic (dlrm_tower_wgrad)."""
    lib = _lib.load()
    L = len(dZs)
    M = dZs[0].size(0)
    widths = [ins[0].size(1)] + [z.size(1) for z in dZs]
    for l in range(L):
        _req(dZs[l], "dZ", ndim=2); _req(ins[l], "in", ndim=2); _req(dWs[l], "dW", ndim=2)
        if (ins[l].size(0) != M or dZs[l].size(0) != M or ins[l].size(1) != widths[l] or dWs[l].size(0) != widths[l + 1]
                or dWs[l].size(1) > widths[l] or (dbs[l] is not None and (dbs[l].numel() != widths[l + 1] or not dbs[l].is_contiguous()))):
            raise RuntimeError("dlrm_amd: tower_wgrad shape mismatch at layer %d" % l)
    if len(ins) != L or len(dWs) != L or len(dbs) != L or not tower_ok(M, widths):
        raise RuntimeError("dlrm_amd: tower_wgrad argument mismatch")
    wa = _int_array(widths)
    need = int(lib.dlrm_tower_wgrad_workspace_bytes(M, L, wa))
    dev = dZs[0].device
    key = (dev, torch.cuda.current_stream(dev).cuda_stream)              # (the stream of the tensors' device: what _stream(dZs[0]) launches on)
    ws = _tower_ws.get(key)
    if ws is None or ws.numel() < need:
        ws = _tower_ws[key] = torch.empty(need, dtype=torch.uint8, device=dev)
    with _timed("linear_bwd_weight"):
        rc = lib.dlrm_tower_wgrad(M, L, wa, _int_array([w.size(1) for w in dWs]),
                                  _lib.ptr_array([z.data_ptr() for z in dZs]), _lib.i64_array([_ld(z) for z in dZs]),
                                  _lib.ptr_array([i.data_ptr() for i in ins]), _lib.i64_array([_ld(i) for i in ins]),
                                  _lib.ptr_array([w.data_ptr() for w in dWs]), _lib.i64_array([_ld(w) for w in dWs]),
                                  _lib.ptr_array([b.data_ptr() if b is not None else 0 for b in dbs]),
                                  C.c_void_p(ws.data_ptr()), ws.numel(), _stream(dZs[0]))
    _lib.check(rc, "dlrm_tower_wgrad")


def act_bwd(dY: torch.Tensor, Y: torch.Tensor, act: int, dZ: torch.Tensor, dbias: Optional[torch.Tensor]) -> torch.Tensor:
    lib = _lib.load()
    _req(dY, "dY", ndim=2); _req(Y, "Y", ndim=2); _req(dZ, "dZ", ndim=2)
    M, N = dY.shape
    with _timed("act_bwd"):
        rc = lib.dlrm_act_bwd(M, N, C.c_void_p(dY.data_ptr()), _ld(dY), C.c_void_p(Y.data_ptr()), _ld(Y), int(act),
                              C.c_void_p(dZ.data_ptr()), _ld(dZ),
                              C.c_void_p(dbias.data_ptr()) if dbias is not None else None, _stream())
    _lib.check(rc, "dlrm_act_bwd")
    return dZ


# ------------------------------------------------------------------------------------------------
# loss, dense SGD
# ------------------------------------------------------------------------------------------------
def _loss_ws(B: int, device) -> torch.Tensor:
    n = _lib.load().dlrm_loss_workspace_bytes(B)
    return torch.empty(max(int(n) // 4, 1), dtype=torch.float32, device=device)


def bce_loss(p: torch.Tensor, target: torch.Tensor, weights: Optional[torch.Tensor], grad_scale: float,
             want_grad: bool, class_weights=(1.0, 1.0)):
    """returns (loss[1], dp or None); p/target are [B] or [B,1] contiguous.  class_weights = (w_neg, w_pos): the wbce
    weights `loss_ws[T.long()]` of the reference applied inside the kernel."""
    lib = _lib.load()
    _req(p, "p"); _req(target, "target")
    if not p.is_contiguous() or not target.is_contiguous() or p.numel() != target.numel():
        raise RuntimeError("dlrm_amd: bce_loss needs contiguous p/target of equal size")
    B = p.numel()
    loss = torch.empty(1, dtype=torch.float32, device=p.device)
    dp = torch.empty_like(p) if want_grad else None
    ws = _loss_ws(B, p.device)
    if weights is not None:
        _req(weights, "weights")
        weights = weights.contiguous()
    with _timed("bce_loss"):
        rc = lib.dlrm_bce_loss(B, C.c_void_p(p.data_ptr()), C.c_void_p(target.data_ptr()),
                               C.c_void_p(weights.data_ptr()) if weights is not None else None,
                               float(class_weights[0]), float(class_weights[1]), float(grad_scale),
                               C.c_void_p(loss.data_ptr()), C.c_void_p(dp.data_ptr()) if dp is not None else None,
                               C.c_void_p(ws.data_ptr()), _stream())
    _lib.check(rc, "dlrm_bce_loss")
    return loss, dp


def bce_logits_loss(z: torch.Tensor, target: torch.Tensor, grad_scale: float, want_grad: bool):
    """BCEWithLogitsLoss(mean) on raw logits (torchrec DLRMTrain): returns (loss[1], dz or None)."""
    lib = _lib.load()
    _req(z, "logits"); _req(target, "target")
    if not z.is_contiguous() or not target.is_contiguous() or z.numel() != target.numel():
        raise RuntimeError("dlrm_amd: bce_logits_loss needs contiguous logits/target of equal size")
    B = z.numel()
    loss = torch.empty(1, dtype=torch.float32, device=z.device)
    dz = torch.empty_like(z) if want_grad else None
    ws = _loss_ws(B, z.device)
    rc = lib.dlrm_bce_logits_loss(B, C.c_void_p(z.data_ptr()), C.c_void_p(target.data_ptr()), float(grad_scale),
                                  C.c_void_p(loss.data_ptr()), C.c_void_p(dz.data_ptr()) if dz is not None else None,
                                  C.c_void_p(ws.data_ptr()), _stream(z))
    _lib.check(rc, "dlrm_bce_logits_loss")
    return loss, dz


def mse_loss(p: torch.Tensor, target: torch.Tensor, grad_scale: float, want_grad: bool):
    lib = _lib.load()
    _req(p, "p"); _req(target, "target")
    if not p.is_contiguous() or not target.is_contiguous() or p.numel() != target.numel():
        raise RuntimeError("dlrm_amd: mse_loss needs contiguous p/target of equal size")
    B = p.numel()
    loss = torch.empty(1, dtype=torch.float32, device=p.device)
    dp = torch.empty_like(p) if want_grad else None
    ws = _loss_ws(B, p.device)
    rc = lib.dlrm_mse_loss(B, C.c_void_p(p.data_ptr()), C.c_void_p(target.data_ptr()), float(grad_scale),
                           C.c_void_p(loss.data_ptr()), C.c_void_p(dp.data_ptr()) if dp is not None else None,
                           C.c_void_p(ws.data_ptr()), _stream())
    _lib.check(rc, "dlrm_mse_loss")
    return loss, dp


def scale_by_scalar(x: torch.Tensor, scalar: torch.Tensor) -> torch.Tensor:
    """x * scalar with `scalar` a 1-element GPU tensor (read on the device)."""
    lib = _lib.load()
    _req(x, "x"); _req(scalar, "scalar")
    if not x.is_contiguous() or scalar.numel() != 1:
        raise RuntimeError("dlrm_amd: scale_by_scalar needs a contiguous tensor and a 1-element scalar")
    y = torch.empty_like(x)
    rc = lib.dlrm_scale_by_device_scalar(x.numel(), C.c_void_p(x.data_ptr()), C.c_void_p(scalar.data_ptr()),
                                         C.c_void_p(y.data_ptr()), _stream())
    _lib.check(rc, "dlrm_scale_by_device_scalar")
    return y


def sgd_dense(w: torch.Tensor, g: torch.Tensor, lr: LrLike) -> None:
    lib = _lib.load()
    _req(w, "w"); _req(g, "g")
    if not w.is_contiguous() or not g.is_contiguous() or w.numel() != g.numel():
        raise RuntimeError("dlrm_amd: sgd_dense needs contiguous tensors of equal size")
    with _timed("sgd_dense"):
        rc = lib.dlrm_sgd_dense(w.numel(), C.c_void_p(w.data_ptr()), C.c_void_p(g.data_ptr()), *_lr_args(lr), _stream())
    _lib.check(rc, "dlrm_sgd_dense")


def sgd_dense_multi(ws: Sequence[torch.Tensor], gs: Sequence[torch.Tensor], lr: LrLike) -> None:
    """w -= lr * g for a whole list of dense parameters, one kernel launch."""
    if not ws:
        return
    lib = _lib.load()
    for w, g in zip(ws, gs):
        _req(w, "w"); _req(g, "g")
        if not w.is_contiguous() or not g.is_contiguous() or w.numel() != g.numel():
            raise RuntimeError("dlrm_amd: sgd_dense_multi needs contiguous tensors of equal size")
    with _timed("sgd_dense"):
        rc = lib.dlrm_sgd_dense_multi(len(ws), _lib.ptr_array([w.data_ptr() for w in ws]),
                                      _lib.ptr_array([g.data_ptr() for g in gs]), _lib.i64_array([w.numel() for w in ws]),
                                      *_lr_args(lr), _stream())
    _lib.check(rc, "dlrm_sgd_dense_multi")


def adagrad_dense(w: torch.Tensor, state_sum: torch.Tensor, g: torch.Tensor, lr: LrLike, eps: float) -> None:
    """state_sum += g*g; w -= lr * g / (sqrt(state_sum) + eps)   (optim/rwsadagrad.py:145-148)"""
    lib = _lib.load()
    _req(w, "w"); _req(g, "g"); _req(state_sum, "state_sum")
    if not (w.is_contiguous() and g.is_contiguous() and state_sum.is_contiguous()) or \
            w.numel() != g.numel() or w.numel() != state_sum.numel():
        raise RuntimeError("dlrm_amd: adagrad_dense needs contiguous tensors of equal size")
    rc = lib.dlrm_adagrad_dense(w.numel(), C.c_void_p(w.data_ptr()), C.c_void_p(state_sum.data_ptr()),
                                C.c_void_p(g.data_ptr()), *_lr_args(lr), float(eps), _stream())
    _lib.check(rc, "dlrm_adagrad_dense")


def set_f32(dsts: Sequence[torch.Tensor], values: Sequence[float]) -> None:
    """dsts[i][0] = values[i] on the current stream, the values travelling in the kernarg (dlrm_set_f32: at most 16 per call)"""
    if len(dsts) != len(values):
        raise RuntimeError("dlrm_amd: set_f32 needs one value per destination")
    for i in range(0, len(dsts), 16):
        d, v = dsts[i:i + 16], values[i:i + 16]
        for t in d:
            _req(t, "scalar")
        rc = _lib.load().dlrm_set_f32(len(d), _lib.ptr_array([t.data_ptr() for t in d]), (C.c_float * len(d))(*[float(x) for x in v]), _stream(d[0]))
        _lib.check(rc, "dlrm_set_f32")


_METRIC_NAMES = ("n", "positives", "tp", "fp", "fn", "tn", "roc_auc", "ap", "round_matches")


def binary_metrics_raw(scores: torch.Tensor, targets: torch.Tensor) -> torch.Tensor:
    """Device float64[9] = (n, positives, TP, FP, FN, TN, roc_auc, average_precision, #{rint(score) == target});
    no host synchronisation.  Replaces the numpy / scikit-learn metrics of inference() (dlrm_s_pytorch.py:819-847)."""
    lib = _lib.load()
    _req(scores, "scores"); _req(targets, "targets")
    scores, targets = scores.reshape(-1), targets.reshape(-1)
    if not scores.is_contiguous() or not targets.is_contiguous() or scores.numel() != targets.numel() or scores.numel() == 0:
        raise RuntimeError("dlrm_amd: binary_metrics needs non-empty contiguous scores/targets of equal size")
    n = scores.numel()
    need = lib.dlrm_binary_metrics_workspace_bytes(n)
    if need < 0:
        raise RuntimeError("dlrm_amd: dlrm_binary_metrics_workspace_bytes failed")
    ws = torch.empty(max(int(need), 256), dtype=torch.uint8, device=scores.device)
    out = torch.empty(9, dtype=torch.float64, device=scores.device)
    rc = lib.dlrm_binary_metrics(n, C.c_void_p(scores.data_ptr()), C.c_void_p(targets.data_ptr()),
                                 C.c_void_p(out.data_ptr()), C.c_void_p(ws.data_ptr()), ws.numel(), _stream())
    _lib.check(rc, "dlrm_binary_metrics")
    return out


def binary_metrics(scores: torch.Tensor, targets: torch.Tensor) -> dict:
    """The validation_results dictionary of the reference's inference() (recall, precision, f1, ap, roc_auc,
    accuracy; dlrm_s_pytorch.py:828-847) plus `round_accuracy` = its non-mlperf accuracy (:819-821).  One D2H copy."""
    v = dict(zip(_METRIC_NAMES, binary_metrics_raw(scores, targets).tolist()))
    tp, fp, fn, tn, n = v["tp"], v["fp"], v["fn"], v["tn"], v["n"]
    # sklearn's zero_division="warn" convention: an undefined ratio counts as 0
    recall = tp / (tp + fn) if tp + fn > 0 else 0.0
    precision = tp / (tp + fp) if tp + fp > 0 else 0.0
    f1 = 2 * tp / (2 * tp + fp + fn) if 2 * tp + fp + fn > 0 else 0.0
    v.update(recall=recall, precision=precision, f1=f1, accuracy=(tp + tn) / n, round_accuracy=v["round_matches"] / n)
    return v


def copy_blocks(srcs: Sequence[torch.Tensor], dsts: Sequence[torch.Tensor]) -> None:
    """dsts[k][:, :] = srcs[k][:, :] for [M, w_k] row-major views with arbitrary row strides, one launch: torch.cat / split
    along dim 1 without ATen (the "cat" interaction and its backward)."""
    lib = _lib.load()
    if len(srcs) != len(dsts) or not srcs:
        raise RuntimeError("dlrm_amd: copy_blocks needs equally many sources and destinations")
    M = srcs[0].size(0)
    for a_, b_ in zip(srcs, dsts):
        _req(a_, "src", ndim=2); _req(b_, "dst", ndim=2)
        if a_.shape != b_.shape or a_.size(0) != M:
            raise RuntimeError("dlrm_amd: copy_blocks shape mismatch")
    if M == 0:
        return
    widths = (C.c_int * len(srcs))(*[int(a_.size(1)) for a_ in srcs])
    rc = lib.dlrm_copy_blocks(M, len(srcs), _lib.ptr_array([a_.data_ptr() for a_ in srcs]), _lib.i64_array([_ld(a_) for a_ in srcs]),
                              _lib.ptr_array([b_.data_ptr() for b_ in dsts]), _lib.i64_array([_ld(b_) for b_ in dsts]),
                              widths, _stream())
    _lib.check(rc, "dlrm_copy_blocks")


def copy_id_blocks(srcs: Sequence[torch.Tensor], dsts: Sequence[torch.Tensor]) -> None:
    """copy_blocks for integer id tensors (int32 / int64, 2-D [M, w_k] views whose rows are contiguous): the id re-layouts of the
    sharded input distribution (ext_dist.kjt_input_dist: destination-major send buffer, source-major -> global-batch-order unpack)
    as ONE strided block-copy launch instead of torch.cat / reshape copies.  Ids travel as 32-bit words."""
    def words(t):
        if t.dtype not in (torch.int32, torch.int64) or t.dim() != 2 or (t.size(1) > 1 and t.stride(1) != 1):
            raise RuntimeError("dlrm_amd: copy_id_blocks needs 2-D int32 / int64 tensors with contiguous rows")
        return (t.view(torch.int32) if t.dtype == torch.int64 else t).view(torch.float32)
    pairs = [(words(a_), words(b_)) for a_, b_ in zip(srcs, dsts) if a_.numel()]
    if pairs:
        copy_blocks([a_ for a_, _ in pairs], [b_ for _, b_ in pairs])


def bce_elementwise(p: torch.Tensor, target: torch.Tensor) -> torch.Tensor:
    lib = _lib.load()
    _req(p, "p"); _req(target, "target")
    if not p.is_contiguous() or not target.is_contiguous() or p.numel() != target.numel():
        raise RuntimeError("dlrm_amd: bce_elementwise needs contiguous p/target of equal size")
    loss = torch.empty_like(p)
    rc = lib.dlrm_bce_elementwise(p.numel(), C.c_void_p(p.data_ptr()), C.c_void_p(target.data_ptr()),
                                  C.c_void_p(loss.data_ptr()), _stream())
    _lib.check(rc, "dlrm_bce_elementwise")
    return loss


def bce_elementwise_bwd(p: torch.Tensor, target: torch.Tensor, dloss: torch.Tensor) -> torch.Tensor:
    lib = _lib.load()
    _req(dloss, "dloss")
    dloss = dloss.contiguous()
    dp = torch.empty_like(p)
    rc = lib.dlrm_bce_elementwise_bwd(p.numel(), C.c_void_p(p.data_ptr()), C.c_void_p(target.data_ptr()),
                                      C.c_void_p(dloss.data_ptr()), C.c_void_p(dp.data_ptr()), _stream())
    _lib.check(rc, "dlrm_bce_elementwise_bwd")
    return dp


def _flat3(*ts):
    n = ts[0].numel()
    for t in ts:
        _req(t, "operand")
        if not t.is_contiguous() or t.numel() != n:
            raise RuntimeError("dlrm_amd: elementwise operands must be contiguous and of equal size")
    return n


def cross_fwd(x0: torch.Tensor, u: torch.Tensor, xl: torch.Tensor, want16: bool = False):
    """x0 * u + xl (DCN-v2 cross layer, elementwise half); want16: also its bf16 rounding (returns (out, out16))"""
    n = _flat3(x0, u, xl)
    out = torch.empty_like(x0)
    out16 = torch.empty(x0.shape, dtype=torch.bfloat16, device=x0.device) if want16 else None
    with _timed("cross_ew"):
        rc = _lib.load().dlrm_cross_fwd(n, C.c_void_p(x0.data_ptr()), C.c_void_p(u.data_ptr()), C.c_void_p(xl.data_ptr()),
                                        C.c_void_p(out.data_ptr()), C.c_void_p(out16.data_ptr()) if want16 else None, _stream(out))
    _lib.check(rc, "dlrm_cross_fwd")
    return (out, out16) if want16 else out


def gemm_bf16_cross(v16: torch.Tensor, W16: torch.Tensor, bias: Optional[torch.Tensor], x0: torch.Tensor, xl: torch.Tensor,
                    want16: bool, category: str = "linear_fwd"):
    """DCN-v2 cross layer, second product with the elementwise half in its epilogue (dlrm_gemm_bf16_cross):
    u = v16 . W16^T + bias;  returns (x_next = x0 * u + xl fp32, bf16(x_next) or None, bf16(u)), or None when the bf16-shaped kernel does not take
    the shape (the caller keeps gemm_bf16 + cross_fwd)."""
    lib = _lib.load()
    M, K = v16.shape
    N = W16.size(0)
    if v16.dtype != torch.bfloat16 or W16.dtype != torch.bfloat16 or W16.size(1) != K or x0.shape != (M, N) or xl.shape != (M, N):
        raise RuntimeError("dlrm_amd: gemm_bf16_cross shape / dtype mismatch")
    _req(x0, "x0", ndim=2); _req(xl, "xl", ndim=2)
    out = torch.empty((M, N), dtype=torch.float32, device=x0.device)
    out16 = torch.empty((M, N), dtype=torch.bfloat16, device=x0.device) if want16 else None
    u16 = torch.empty((M, N), dtype=torch.bfloat16, device=x0.device)
    with _timed(category):
        rc = lib.dlrm_gemm_bf16_cross(M, N, K, C.c_void_p(v16.data_ptr()), v16.stride(0), C.c_void_p(W16.data_ptr()), W16.stride(0),
                                      C.c_void_p(bias.data_ptr()) if bias is not None else None,
                                      C.c_void_p(x0.data_ptr()), _ld(x0), C.c_void_p(xl.data_ptr()), _ld(xl),
                                      C.c_void_p(u16.data_ptr()), u16.stride(0), C.c_void_p(out.data_ptr()), _ld(out),
                                      C.c_void_p(out16.data_ptr()) if out16 is not None else None, out16.stride(0) if out16 is not None else 0,
                                      _stream(out))
    if rc == -4:                       # DLRM_E_MODE: outside the bf16-shaped kernel's preconditions
        return None
    _lib.check(rc, "dlrm_gemm_bf16_cross")
    return out, out16, u16


def cross_bwd(g: torch.Tensor, x0: torch.Tensor, u: torch.Tensor, dx0: torch.Tensor, accumulate: bool, out: str = "f32"):
    """du = g * x0 (out "f32": fp32 tensor; "bf16": only its bf16 rounding — what a bf16-storage weight / data gradient reads); dx0 (+)= g * u
    (u: the fp32 tensor, or the bf16 copy gemm_bf16_cross stored)"""
    n = _flat3(g, x0, dx0)
    if u.dtype == torch.bfloat16:
        if not u.is_contiguous() or u.numel() != n:
            raise RuntimeError("dlrm_amd: elementwise operands must be contiguous and of equal size")
    else:
        _flat3(g, u)
    du = torch.empty_like(g) if out == "f32" else None
    du16 = torch.empty(g.shape, dtype=torch.bfloat16, device=g.device) if out == "bf16" else None
    with _timed("cross_ew"):
        rc = _lib.load().dlrm_cross_bwd(n, C.c_void_p(g.data_ptr()), C.c_void_p(x0.data_ptr()),
                                        C.c_void_p(u.data_ptr()) if u.dtype != torch.bfloat16 else None,
                                        C.c_void_p(u.data_ptr()) if u.dtype == torch.bfloat16 else None,
                                        C.c_void_p(du.data_ptr()) if du is not None else None,
                                        C.c_void_p(du16.data_ptr()) if du16 is not None else None,
                                        C.c_void_p(dx0.data_ptr()), int(bool(accumulate)), _stream(g))
    _lib.check(rc, "dlrm_cross_bwd")
    return du if out == "f32" else du16


def add(a: torch.Tensor, b: torch.Tensor) -> torch.Tensor:
    n = _flat3(a, b)
    out = torch.empty_like(a)
    with _timed("cross_ew"):
        rc = _lib.load().dlrm_add(n, C.c_void_p(a.data_ptr()), C.c_void_p(b.data_ptr()), C.c_void_p(out.data_ptr()), _stream(a))
    _lib.check(rc, "dlrm_add")
    return out


def clamp(x: torch.Tensor, lo: float, hi: float) -> torch.Tensor:
    lib = _lib.load()
    _req(x, "x")
    if not x.is_contiguous():
        raise RuntimeError("dlrm_amd: clamp needs a contiguous tensor")
    y = torch.empty_like(x)
    _lib.check(lib.dlrm_clamp(x.numel(), C.c_void_p(x.data_ptr()), float(lo), float(hi), C.c_void_p(y.data_ptr()), _stream()),
               "dlrm_clamp")
    return y


def clamp_bwd(x: torch.Tensor, lo: float, hi: float, dy: torch.Tensor) -> torch.Tensor:
    lib = _lib.load()
    _req(dy, "dy")
    dy = dy.contiguous()
    dx = torch.empty_like(x)
    _lib.check(lib.dlrm_clamp_bwd(x.numel(), C.c_void_p(x.data_ptr()), float(lo), float(hi), C.c_void_p(dy.data_ptr()),
                                  C.c_void_p(dx.data_ptr()), _stream()), "dlrm_clamp_bwd")
    return dx


def device_info(device: int = 0) -> dict:
    lib = _lib.load()
    cu, lds, hbm = C.c_int(), C.c_int(), C.c_int64()
    name = C.create_string_buffer(256)
    rc = lib.dlrm_hip_device_info(device, C.byref(cu), C.byref(lds), C.byref(hbm), name, 256)
    _lib.check(rc, "dlrm_hip_device_info")
    return {"name": name.value.decode(), "cu_count": cu.value, "lds_bytes": lds.value, "hbm_bytes": hbm.value,
            "build": lib.dlrm_hip_build_info().decode(), "abi": lib.dlrm_hip_abi_version()}
