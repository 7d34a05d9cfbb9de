"""Synthetic input batches generated on the device.

Host-side mirror of the reference's random data path — `RandomDataset` / `generate_dist_input_batch(uniform)` /
`generate_random_output_batch` (dlrm_data_pytorch.py:614-700, 899-960, 835-846) — with the same knobs
(`m_den`, `ln_emb`, `num_indices_per_lookup`, `num_indices_per_lookup_fixed`, `round_targets`, seed) and the same
output structure (X [B, m_den] f32, lS_o: T offset tensors [B], lS_i: T index tensors [nnz_t], T [B, 1] f32), but
every value is produced in HBM by `dlrm_gen_uniform_bags` / `dlrm_gen_uniform_dense` (Philox4x32-10 keyed by
(seed, batch number)): no Python loop over bags, no H2D copy.
"""
from __future__ import annotations

import ctypes as C
from typing import List, Sequence, Tuple

import torch

from . import _lib, ops


class UniformBatchGenerator:
    def __init__(self, m_den: int, ln_emb: Sequence[int], num_indices_per_lookup: int = 10,
                 num_indices_per_lookup_fixed: bool = False, round_targets: bool = True, seed: int = 123,
                 device=None, index_dtype: torch.dtype = torch.int64):
        if index_dtype not in (torch.int64, torch.int32):
            raise RuntimeError("dlrm_amd.datagen: index dtype must be int64 or int32")
        self.m_den, self.rows = int(m_den), [int(n) for n in ln_emb]
        self.P, self.fixed = int(num_indices_per_lookup), bool(num_indices_per_lookup_fixed)
        self.round_targets, self.seed = bool(round_targets), int(seed)
        self.device = torch.device(device if device is not None else "cuda:0")
        if self.device.type != "cuda":
            raise RuntimeError("dlrm_amd.datagen: the generator runs on the GPU only (no CPU fallback)")
        self.index_dtype = index_dtype
        self._ws = None

    def _seed(self, batch_no: int, stream_id: int) -> int:
        # distinct Philox keys per (run seed, batch, purpose): splitmix-style mix kept inside 64 bits
        z = (self.seed * 0x9E3779B97F4A7C15 + batch_no * 0xBF58476D1CE4E5B9 + stream_id * 0x94D049BB133111EB) & (2 ** 64 - 1)
        z ^= z >> 31
        return z & (2 ** 64 - 1)

    def batch(self, B: int, batch_no: int = 0, stacked: bool = False):
        """-> (X, lS_o, lS_i, T): lS_o / lS_i lists of per-table tensors, or (stacked=True, fixed bag lengths only) the reference
        collate layout, one [T, B] tensor each (dlrm_data_pytorch.py:686,337)."""
        if stacked and not self.fixed:
            raise RuntimeError("dlrm_amd.datagen: stacked=True needs num_indices_per_lookup_fixed (equal nnz per table)")
        lib = _lib.load()
        dev, T = self.device, len(self.rows)
        st = C.c_void_p(torch.cuda.current_stream(dev).cuda_stream)
        X = torch.empty((B, self.m_den), dtype=torch.float32, device=dev)
        tgt = torch.empty((B, 1), dtype=torch.float32, device=dev)
        _lib.check(lib.dlrm_gen_uniform_dense(X.numel(), C.c_void_p(X.data_ptr()), 0, self._seed(batch_no, 1), st),
                   "dlrm_gen_uniform_dense")
        _lib.check(lib.dlrm_gen_uniform_dense(tgt.numel(), C.c_void_p(tgt.data_ptr()), int(self.round_targets),
                                              self._seed(batch_no, 2), st), "dlrm_gen_uniform_dense")
        lS_o: List[torch.Tensor] = []
        lS_i: List[torch.Tensor] = []
        bits = 64 if self.index_dtype == torch.int64 else 32
        onehot = self.P == 1 and self.fixed
        for t0 in range(0, T, 32):                                   # DLRM_MAX_TABLES_PER_LAUNCH tables per launch
            rows = self.rows[t0:t0 + 32]
            n = len(rows)
            off = torch.empty((n, B), dtype=self.index_dtype, device=dev)
            idx = torch.empty((n, B * self.P), dtype=self.index_dtype, device=dev)
            nnz = torch.empty(n, dtype=torch.int64, device=dev)
            ws_ptr, ws_bytes = None, 0
            if not onehot:
                need = lib.dlrm_gen_workspace_bytes(n, B)
                if need < 0:
                    raise RuntimeError("dlrm_amd.datagen: dlrm_gen_workspace_bytes failed")
                if self._ws is None or self._ws.numel() < need:
                    self._ws = torch.empty(int(need), dtype=torch.uint8, device=dev)
                ws_ptr, ws_bytes = C.c_void_p(self._ws.data_ptr()), self._ws.numel()
            rc = lib.dlrm_gen_uniform_bags(n, B, _lib.i64_array(rows), self.P, int(self.fixed), self._seed(batch_no, 16 + t0),
                                           bits, _lib.ptr_array([off[k].data_ptr() for k in range(n)]),
                                           _lib.ptr_array([idx[k].data_ptr() for k in range(n)]),
                                           C.c_void_p(nnz.data_ptr()), ws_ptr, ws_bytes, st)
            _lib.check(rc, "dlrm_gen_uniform_bags")
            counts = [B] * n if onehot else nnz.tolist()             # variable bag lengths: one small D2H copy per batch
            for k in range(n):
                o_k = off[k]
                if onehot:
                    # gen_onehot_kernel writes bag start b for bag b: the producer's own proof of "one lookup per bag"
                    # (ops.offsets_are_iota then needs no device pass and no synchronisation for this tensor object)
                    ops.mark_one_lookup_per_bag(o_k)
                lS_o.append(o_k)
                lS_i.append(idx[k, :counts[k]])
            if onehot and T <= 32 and stacked:
                return X, ops.mark_one_lookup_per_bag(off), idx, tgt
        if stacked:
            so = torch.stack(lS_o)
            if onehot:
                ops.mark_one_lookup_per_bag(so)          # a copy of rows this generator wrote as 0..B-1
            return X, so, torch.stack(lS_i), tgt
        return X, lS_o, lS_i, tgt
