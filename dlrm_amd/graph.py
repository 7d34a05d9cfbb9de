"""Whole-step HIP graph: forward + loss + backward + fused embedding update + dense optimizer step captured once
and replayed with one `hipGraphLaunch` per training step.

Why: at Criteo-Kaggle shapes (BASELINE.json configs[1]: B = 2048, D = 16) one step is ~70 kernel launches of a few
microseconds each and the Python/ctypes/autograd host path (~1 ms per step, measured) is the whole step time; at
Criteo-Terabyte shapes the GPU work (9 ms) hides the host path and replay only removes the inter-kernel gaps.
The reference loop body (dlrm_s_pytorch.py:1574-1621) is what gets captured, unchanged in order:
    Z = dlrm(X, lS_o, lS_i); E = loss_fn(Z, T); optimizer.zero_grad(); E.backward(); optimizer.step()

Everything the C ABI enqueues is capture-safe by construction: kernels take table / tensor pointers by value in
the kernarg segment, nothing allocates or synchronises, scratch buffers come from torch's caching allocator (graph
private pool during capture) and the segmented sort of the embedding updates is made of plain kernels on the capture stream.

Constraints (checked, never silently worked around):
  * single process (ext_dist.my_size == 1): RCCL collectives and DDP hooks are not captured;
  * fixed shapes: every call must pass tensors of the shapes/dtypes of the first call (multi-hot batches whose number
    of lookups varies cannot be replayed) — inputs are copied into static device buffers before each replay;
  * no autograd graph of an earlier eager step may still be alive at capture time (e.g. a loss tensor the caller kept):
    the autograd engine would synchronise the capture stream with the stream those old nodes were created on, which
    invalidates the capture.  Reduce losses to Python floats (`float(loss)`) or `del` them before the first graphed call;
  * the GPU memory fault at B = 65536 of round 1 (profiles/r02/graph_probe.md d) came from replaying rocPRIM's radix sort, not from
    the batch size (tools/probes/graph_sorted_probe.py).  The sort-based updates now run on the library's own segmented sorter
    (csrc/seg_sort.h) whenever every table segment holds <= 262144 lookups — then the graph keeps the sorted update and the fused
    row-wise Adagrad; otherwise it switches SGD to the atomic update and refuses the Adagrad update (GraphedTrainStep.
    _settle_sort_mode).  tests/test_gpu_model.py replays 36 full-batch steps with host syncs in between, bit-identical to eager;
  * learning rates (round 6): with this package's optimizers (FusedSGD, FusedRWSAdagrad with lr_decay == 0) every captured update kernel
    reads its step size from a DEVICE scalar (one per param group; include/dlrm_hip.h "LEARNING RATES"), and a replay whose param groups
    carry new lr values writes them in front of the launch (dlrm_graph_replay / dlrm_set_f32: values in the kernarg, no host buffer, no
    synchronisation) — the reference's LRPolicyScheduler, which moves lr EVERY iteration during warm-up and decay
    (dlrm_s_pytorch.py:169-203, :1621), is followed with ONE capture.  Any other optimizer (torch.optim.SGD: its foreach step takes lr as a
    host number) keeps the old rule: lr is baked in at capture time and a change re-captures the step.
"""
from __future__ import annotations

from typing import List, Optional, Sequence, Union

import os as _os

import torch

from . import ext_dist, ops

TensorOrList = Union[torch.Tensor, Sequence[torch.Tensor]]
_TRACE = _os.environ.get("DLRM_GTS_TRACE", "0") == "1"


def _clone_struct(x: TensorOrList):
    """Static copies; tensors that appear several times in a list (e.g. one shared offsets tensor for all tables) are
    cloned once and aliased the same way."""
    if isinstance(x, torch.Tensor):
        return x.clone()
    seen = {}
    out = []
    for t in x:
        if id(t) not in seen:
            seen[id(t)] = t.clone()
        out.append(seen[id(t)])
    return out


def _copy_now(d: torch.Tensor, s_: torch.Tensor) -> None:
    d.copy_(s_, non_blocking=True)


def _copy_struct(dst: TensorOrList, src: TensorOrList, put=_copy_now) -> None:
    """static buffers <- the caller's tensors: `put(dst_tensor, src_tensor)` for every distinct tensor whose storage differs, after the
    shape / dtype checks (default: an ATen copy_ each; the raw replay path collects the pairs for dlrm_graph_replay instead).  (All copies
    in ONE launch of the library's block copy was measured too: building its views costs the host more than the four copy_ calls it
    replaces, and the host is what the GPU waits for between replays — Criteo-Kaggle graph 0.378 -> 0.406 ms, profiles/round5/kaggle_towers.md.)"""
    if isinstance(dst, torch.Tensor):
        if not isinstance(src, torch.Tensor) or src.shape != dst.shape or src.dtype != dst.dtype:
            raise RuntimeError("dlrm_amd.graph: input shape/dtype differs from the captured step")
        if src.data_ptr() != dst.data_ptr():
            put(dst, src)
        return
    if isinstance(src, torch.Tensor) or len(src) != len(dst):
        raise RuntimeError("dlrm_amd.graph: input structure differs from the captured step")
    done = set()
    for d, s_ in zip(dst, src):
        if s_.shape != d.shape or s_.dtype != d.dtype:
            raise RuntimeError("dlrm_amd.graph: input shape/dtype differs from the captured step "
                               "(variable-length multi-hot batches cannot be replayed)")
        if id(d) in done or s_.data_ptr() == d.data_ptr():
            continue
        done.add(id(d))
        # one copy per distinct tensor.  (torch._foreach_copy_ on int64 lists raised GPU memory faults on
        # torch 2.10 + ROCm 7 at B = 65536 — profiles/r02/graph_probe.md; pass stacked [T, B] index / offset tensors to
        # make this a single copy.)
        put(d, s_)


class GraphedTrainStep:
    """step = GraphedTrainStep(model, optimizer);  loss = step(X, lS_o, lS_i, T)   (loss: static 0-dim device tensor)

    The first `warmup` calls are ordinary eager steps (kernel attributes, workspaces and allocator pools settle); the
    next call captures the step and every call from then on copies its inputs into the static buffers and replays.
    `self.out` holds the predictions of the last step (static buffer)."""

    def __init__(self, model, optimizer, warmup: int = 2):
        if ext_dist.is_distributed():       # (a forced one-rank group routes through distributed_forward as well: its RCCL calls must not be captured)
            raise RuntimeError("dlrm_amd.graph: the whole-step HIP graph is single-process only "
                               "(RCCL all-to-all / DDP all-reduce are not captured)")
        self.model, self.optimizer = model, optimizer
        for g in optimizer.param_groups:
            # a decaying step size (RWSAdagrad: clr = lr / (1 + (step - 1) * lr_decay)) is computed on the host at capture time
            # and baked into kernel arguments: a replay would silently reuse a stale clr
            if float(g.get("lr_decay", 0.0) or 0.0) != 0.0:
                raise RuntimeError("dlrm_amd.graph: lr_decay != 0 changes the step size every step; a captured graph cannot "
                                   "follow it (use the eager step)")
        # Sort-based updates (DLRM_UPD_SORTED, the fused row-wise Adagrad): rocPRIM's onesweep radix sort ends in a memory-aperture
        # violation when REPLAYED from a HIP graph on ROCm 7.2 (tools/probes/graph_sorted_probe.py; the "TB-shape graph fault" of
        # profiles/r02/graph_probe.md).  Since round 3 of the build the library sorts table segments of up to 262144 lookups with its
        # own kernels (csrc/seg_sort.h: no memsets, no vendor code, replayable); whether THIS step's shapes are covered is only known
        # when the first batch arrives — decided in _settle_sort_mode().
        self._sort_checked = False
        self._one = None
        # the captured step is single-stream: a replayed graph has no launch gaps to hide, and the side-stream overlap
        # of the eager path (DLRM_Net.overlap_streams) would put cross-stream joins into the capture
        model.overlap_streams = False
        # ... and takes the whole sparse update at its optimizer step (the update-in-backward schedule of ABI 17 allocates its sorted
        # workspace per backward pass and binds to the optimizer's first EAGER step: not part of a captured step)
        model.update_in_backward = False
        self.warmup = max(int(warmup), 1)
        self.graph: Optional[torch.cuda.CUDAGraph] = None
        self.stream: Optional[torch.cuda.Stream] = None
        self.static = None
        self.loss: Optional[torch.Tensor] = None
        self.out: Optional[torch.Tensor] = None
        self._lrs: List[float] = []
        # step sizes on the device: one fp32 scalar per param group, read by the captured update kernels (module docstring)
        self._lr_on_device = type(optimizer).__name__ in ("FusedSGD", "FusedRWSAdagrad") and _os.environ.get("DLRM_GTS_DEVICE_LR", "1") == "1"
        self._lr_dev: Optional[torch.Tensor] = None
        self.lr_writes = 0          # replays that carried new step sizes (observability: tests, bench)
        self.captures = 0
        self._eager_calls = 0
        self._replayed = False
        self.serialize = _os.environ.get("DLRM_GTS_SERIALIZE", "1") == "1"
        # raw replay: wait + input copies + hipGraphLaunch in ONE C call (dlrm_graph_replay) instead of stream.synchronize(), a copy_ per
        # input and CUDAGraph.replay() from Python — the host path is what a launch-bound step waits for.  DLRM_GTS_RAW=0: the torch calls.
        self.raw = _os.environ.get("DLRM_GTS_RAW", "1") == "1"
        self._exec = None          # hipGraphExec_t of the captured step (None: not available -> torch's replay)
        self._raw_arrays = None    # cached ctypes arrays of the raw call

    def _settle_sort_mode(self, lS_o, lS_i) -> None:
        """First batch: keep the sort-based embedding update only if every table segment goes through the library's own segmented
        sorter (ops.sort_is_graph_safe); otherwise SGD falls back to the atomic update and the row-wise Adagrad is refused."""
        if self._sort_checked:
            return
        self._sort_checked = True
        model, optimizer = self.model, self.optimizer
        adagrad = any(type(optimizer).__name__ == n for n in ("RWSAdagrad", "FusedRWSAdagrad"))
        if not adagrad and getattr(model, "emb_update_mode", None) != ops.UPD_SORTED:
            return
        safe, lookups, why = False, 0, None
        try:
            bags = ops.BagBatch(lS_o, lS_i, None)
            safe = ops.sort_is_graph_safe([e.weight for e in model.emb_l], bags)
            lookups = int(sum(bags.nnz))
            # (decided once: every later batch has these segment sizes — _copy_struct refuses any other input shape)
        except Exception as e:                               # noqa: BLE001 - models without plain emb_l tables: be conservative
            safe, why = False, "%s: %s" % (type(e).__name__, e)
        # DLRM_GRAPH_SORTED: "1" keep the sorted update whenever it is replayable, "0" never, default "auto": keep it for batches of
        # >= 2^19 lookups (Criteo-Terabyte: 1.7 M) and take the atomic update for launch-bound batches, where the sort's nine small
        # launches cost more than its rows save (Criteo-Kaggle shapes, 53 k lookups: 0.83 ms sorted vs atomic, profiles/round3)
        pref = _os.environ.get("DLRM_GRAPH_SORTED", "auto")
        if pref == "0" or (pref == "auto" and not adagrad and lookups < (1 << 19)):
            safe = False
        if safe:
            return
        if adagrad:
            raise RuntimeError("dlrm_amd.graph: the fused row-wise Adagrad update of this batch cannot be replayed from a HIP graph: " +
                               ("the sorter check itself failed (%s)" % why if why else
                                "a table segment needs the general (rocPRIM) sorter, which cannot be replayed on this ROCm "
                                "(tools/probes/graph_sorted_probe.py)") + "; use the eager step")
        model.emb_update_mode = ops.UPD_ATOMIC               # LDS pre-reduction for tiny tables + hardware fp32 atomics

    def _prove_one_lookup_per_bag(self, lS_o, lS_i) -> None:
        """The fused lookup + interaction kernels are only valid for offsets == arange(B).  Eager steps prove that per offsets tensor
        (ops.offsets_are_iota); inside a capture nothing can synchronise and the static offsets buffer is rewritten before every
        replay, so the proof is made HERE, on the caller's tensors, before they are copied.  A batch that fails it turns the fused
        path off for this model and drops the captured graph (the next call re-captures with the two kernels)."""
        model = self.model
        if not getattr(model, "fuse_emb_interact", False):
            return
        # (a model that cannot take the fused path anyway — Criteo-Kaggle's D = 16, a cat interaction, pooling weights — needs no proof: a
        # per-replay host wait doubled that launch-bound step when every replay brought a new offsets tensor)
        try:
            D = model.emb_l[0].weight.size(1)
            if (getattr(model, "arch_interaction_op", "dot") != "dot" or not ops.gather_ok(1 + len(model.emb_l), D)
                    or any(w is not None for w in (getattr(model, "v_W_l", None) or []))):
                return
        except (AttributeError, IndexError, TypeError):
            pass
        n_i = lS_i.size(-1) if isinstance(lS_i, torch.Tensor) else None
        n_o = lS_o.size(-1) if isinstance(lS_o, torch.Tensor) else None
        if isinstance(lS_i, torch.Tensor) != isinstance(lS_o, torch.Tensor) or (n_i is not None and n_i != n_o):
            return                              # not the one-lookup shape: sequential_forward will not take the fused path
        if n_i is None and any(i.numel() != o.numel() for i, o in zip(lS_i, lS_o)):
            return
        if ops.offsets_are_iota(lS_o) is False:
            model.fuse_emb_interact = False
            self._drop_graph(lS_o)

    def _drop_graph(self, like) -> None:
        """forget the captured step (the next call re-captures): the replay in flight is waited for first, and the raw handle and its
        cached argument arrays go with the graph — a dangling hipGraphExec_t must never reach dlrm_graph_replay"""
        if self.graph is None:
            return
        dev = like.device if isinstance(like, torch.Tensor) else like[0].device
        torch.cuda.current_stream(dev).synchronize()
        self._replayed = False
        self.graph.reset()
        self.graph, self._exec, self._raw_arrays = None, None, None

    # one eager training step on the static buffers (the reference loop body)
    def _eager(self):
        X, lS_o, lS_i, T = self.static
        Z = self.model(X, lS_o, lS_i)
        E = self.model.loss_fn(Z, T)
        # E.backward() would run the parameters' AccumulateGrad nodes; those are created once and live as long as ANY
        # autograd graph references them (a loss tensor the caller kept from an earlier eager step is enough), on the
        # stream of that time — which breaks stream capture.  torch.autograd.grad returns the gradients without
        # AccumulateGrad; the embedding tables are listed so that their Function's backward (the fused-update stash)
        # runs, and come back as None exactly like after backward().
        params = [p for g in self.optimizer.param_groups for p in g["params"] if p.requires_grad]
        if self._one is None or self._one.device != E.device:
            self._one = torch.ones((), dtype=E.dtype, device=E.device)      # the seed of backward: without it autograd launches an ATen fill per step
        grads = torch.autograd.grad(E, params, grad_outputs=self._one, allow_unused=True)
        for p, g in zip(params, grads):
            p.grad = g
        self.optimizer.step()
        return Z, E

    def _current_lrs(self) -> List[float]:
        return [float(g["lr"]) for g in self.optimizer.param_groups]

    def _capture(self) -> None:
        if ops.timers is not None:
            raise RuntimeError("dlrm_amd.graph: per-kernel event timers cannot be recorded inside a graph capture")
        dev = self.static[0].device
        if self.stream is None:
            self.stream = torch.cuda.Stream(device=dev)
        self.stream.wait_stream(torch.cuda.current_stream(dev))
        if self.graph is not None:
            self.graph.reset()
        self.graph = torch.cuda.CUDAGraph()
        groups = self.optimizer.param_groups
        if self._lr_on_device:
            # one device scalar per param group, holding the CURRENT lr; registered for the duration of the capture only, so that an eager
            # step of the same optimizer outside the graph keeps passing its host value
            self._lr_dev = torch.tensor(self._current_lrs(), dtype=torch.float32, device=dev)
            self.stream.wait_stream(torch.cuda.current_stream(dev))
            for i, g in enumerate(groups):
                ops._graph_lr[id(g)] = self._lr_dev[i:i + 1]
        try:
            # backward runs on the autograd engine's thread: "relaxed" lets that thread enqueue into the capturing stream
            with torch.cuda.graph(self.graph, stream=self.stream, capture_error_mode="relaxed"):
                Z, E = self._eager()
        finally:
            for g in groups:
                ops._graph_lr.pop(id(g), None)
        self.out, self.loss = Z.detach(), E.detach()
        self._lrs = self._current_lrs()
        # the graph holds raw pointers into the cached scratch workspaces: pin those tensors so that a later, larger
        # request elsewhere (which replaces the cache entry) cannot free memory the replays still use
        self._pinned_ws = [w for (d_, _), w in list(ops._emb_ws.items()) + list(ops._wgrad_ws.items()) if d_ == dev]
        self._pinned_ws += [w for (d_, _), w in ops._tower_ws.items() if d_ == dev]
        self.captures += 1
        self._exec, self._raw_arrays = None, None
        if self.raw:
            try:
                self._exec = int(self.graph.raw_cuda_graph_exec())
            except Exception:                               # noqa: BLE001 - a torch without the raw handle: keep its own replay
                self._exec = None
        torch.cuda.current_stream(dev).wait_stream(self.stream)

    def _changed_lrs(self):
        """(count, device pointers, values) of the step sizes this replay must write first: every group's value whenever any changed (a
        schedule moves all groups together; at most 16 go with dlrm_graph_replay, more are written by dlrm_set_f32 right here)"""
        import ctypes as C
        if not self._lr_on_device or self._lr_dev is None:
            return 0, None, None
        lrs = self._current_lrs()
        if lrs == self._lrs:
            return 0, None, None
        self._lrs = lrs
        self.lr_writes += 1
        n = len(lrs)
        if n > 16:
            ops.set_f32([self._lr_dev[i:i + 1] for i in range(n)], lrs)
            return 0, None, None
        base = self._lr_dev.data_ptr()
        return n, (C.c_void_p * n)(*[base + 4 * i for i in range(n)]), (C.c_float * n)(*lrs)

    def _lr_current_or_on_device(self) -> bool:
        """True when the captured graph is valid for the optimizer's present learning rates"""
        return self._lr_on_device or self._lrs == self._current_lrs()

    def _raw_replay(self, X, lS_o, lS_i, T) -> bool:
        """the steady state: inputs checked and collected, then ONE call that waits for the previous replay, copies and launches"""
        import ctypes as C
        from . import _lib
        pairs = []
        xs, os_, is_, ts = self.static
        put = lambda d, s_: pairs.append((d, s_))           # noqa: E731
        _copy_struct(xs, X, put); _copy_struct(os_, lS_o, put); _copy_struct(is_, lS_i, put); _copy_struct(ts, T, put)
        n = len(pairs)
        for d, s_ in pairs:
            if not (d.is_cuda and s_.is_cuda and s_.device == d.device and d.is_contiguous() and s_.is_contiguous()):
                return False                                # (host tensors, strided views: torch's copy_ handles them)
        arr = self._raw_arrays
        if arr is None or arr[0] != n:
            arr = self._raw_arrays = (n, (C.c_void_p * max(n, 1))(), (C.c_void_p * max(n, 1))(), (C.c_int64 * max(n, 1))())
        _, dst, src, nbytes = arr
        for i, (d, s_) in enumerate(pairs):
            dst[i], src[i], nbytes[i] = d.data_ptr(), s_.data_ptr(), d.numel() * d.element_size()
        ns, sdst, sval = self._changed_lrs()
        rc = _lib.load().dlrm_graph_replay(n, dst, src, nbytes, ns, sdst, sval, C.c_void_p(self._exec), int(self._replayed),
                                           C.c_void_p(torch.cuda.current_stream(X.device).cuda_stream))
        _lib.check(rc, "dlrm_graph_replay")
        self._keep = pairs                                  # the sources stay alive until the next call (their copies are asynchronous)
        self._replayed = self.serialize
        return True

    def __call__(self, X, lS_o, lS_i, T):
        if (self._exec is not None and self.static is not None and self.graph is not None and self._eager_calls >= self.warmup
                and self._lr_current_or_on_device() and not _TRACE):
            self._prove_one_lookup_per_bag(lS_o, lS_i)
            if self.graph is not None and self._raw_replay(X, lS_o, lS_i, T):
                return self.loss
        if _TRACE:
            print("[gts] call eager=%d captures=%d replay_in_flight=%s" % (self._eager_calls, self.captures, self._replayed),
                  flush=True)
        if self._replayed:
            # at most ONE replay in flight, and a full stream synchronisation between replays.  Two launches of the same
            # executable graph queued behind each other overlapped on ROCm 7.2 (they share every intermediate buffer), and
            # event waits alone were not enough once the application synchronised the stream somewhere in between
            # (profiles/r02/graph_probe.md, b and d): the pattern that was clean in every probe is "synchronise the stream
            # after every replay", so that is what happens here before the static inputs are touched again.  At
            # Criteo-Terabyte sizes the GPU is the bottleneck (the wait costs one launch latency per step); at launch-bound
            # sizes the replay has finished long before the host gets here.
            ops.wait_spinning(torch.cuda.current_stream(X.device))      # (polled, not slept on: ops._wait_event_spinning)
            self._replayed = False
        self._prove_one_lookup_per_bag(lS_o, lS_i)
        if self.static is None:
            self._settle_sort_mode(lS_o, lS_i)
            self.static = (X.clone(), _clone_struct(lS_o), _clone_struct(lS_i), T.clone())
        else:
            xs, os_, is_, ts = self.static
            _copy_struct(xs, X)
            _copy_struct(os_, lS_o)
            _copy_struct(is_, lS_i)
            _copy_struct(ts, T)
        dev = X.device
        if self._eager_calls < self.warmup:
            # the first calls are ordinary eager steps, issued on the stream the capture will use so that kernel
            # attributes, scratch workspaces and allocator pools are those the graph will see
            if self.stream is None:
                self.stream = torch.cuda.Stream(device=dev)
            self.stream.wait_stream(torch.cuda.current_stream(dev))
            with torch.cuda.stream(self.stream):
                Z, E = self._eager()
            torch.cuda.current_stream(dev).wait_stream(self.stream)
            self._eager_calls += 1
            self.out, self.loss = Z.detach(), E.detach()
            return self.loss
        if self.graph is None or not self._lr_current_or_on_device():
            self._capture()          # capture only records; the replay below executes this step
        ns, sdst, sval = self._changed_lrs()
        if ns:                       # torch's replay path: the new step sizes go in front of the launch on the same stream
            from . import _lib
            _lib.check(_lib.load().dlrm_set_f32(ns, sdst, sval, torch.cuda.current_stream(dev).cuda_stream), "dlrm_set_f32")
        self.graph.replay()
        self._replayed = self.serialize
        return self.loss
