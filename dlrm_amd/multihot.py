"""MLPerf-v2 multi-hot synthetic inputs on the device (BASELINE.json configs[4]).

`Multihot` mirrors the reference class of the same name (torchrec_dlrm/multi_hot.py:27-175): same constructor arguments,
same products — per table a seed-determined lookup table [n_t, h_t] whose column 0 is the id itself, the expansion of a
batch of 1-hot ids into h_t ids per sample (table-major KJT `values`) and the cumulative `offsets` — but the tables live in
HBM and a batch is expanded by one kernel launch (`dlrm_multihot_expand`) instead of T `F.embedding` calls on the host.

The expansion is exact integer work and is tested bit-for-bit against the reference class (tables uploaded with
`from_host_tables`).  Tables GENERATED here come from Philox4x32-10 (`dlrm_multihot_gen_table`): the reference's
distributions ("uniform": randint(0, n); "pareto": int32(pareto(0.25)) % n), not numpy's seed-0 MT19937 sequence — the
MLPerf sizes need 24 GB of lookup tables, which the reference draws through a 64-bit host temporary per table.
"""
from __future__ import annotations

import ctypes as C
from typing import List, Optional, Sequence

import numpy as np
import torch

from . import _lib, ops

# torchrec_dlrm/README.MD:159 (multi_hot_sizes) and :45 (num_embeddings_per_feature) of the MLPerf-v2 benchmark: 214 lookups/sample
MLPERF_V2_MULTI_HOT_SIZES = [3, 2, 1, 2, 6, 1, 1, 1, 1, 7, 3, 8, 1, 6, 9, 5, 1, 1, 1, 12, 100, 27, 10, 3, 1, 1]
MLPERF_V2_NUM_EMBEDDINGS = [40000000, 39060, 17295, 7424, 20265, 3, 7122, 1543, 63, 40000000, 3067956, 405282, 10, 2209, 11938,
                            155, 4, 976, 14, 40000000, 40000000, 40000000, 590152, 12973, 108, 36]


class Multihot:
    def __init__(self, multi_hot_sizes: Sequence[int], num_embeddings_per_feature: Sequence[int], batch_size: int,
                 collect_freqs_stats: bool = False, dist_type: str = "uniform", device=None, seed: int = 0,
                 host_tables: Optional[Sequence[np.ndarray]] = None):
        if dist_type not in {"uniform", "pareto"}:
            raise ValueError("Multi-hot distribution type {} is not supported."
                             'Only "uniform" and "pareto" are supported.'.format(dist_type))
        if collect_freqs_stats:
            raise RuntimeError("dlrm_amd.Multihot: access-frequency statistics are a host-side plotting aid of the reference "
                               "(multi_hot.py:57-78); not provided on the device path")
        if len(multi_hot_sizes) != len(num_embeddings_per_feature):
            raise ValueError("multi_hot_sizes and num_embeddings_per_feature differ in length")
        self.dist_type = dist_type
        self.multi_hot_sizes = [int(h) for h in multi_hot_sizes]
        self.num_embeddings_per_feature = [int(n) for n in num_embeddings_per_feature]
        self.batch_size = int(batch_size)
        self.device = torch.device("cuda", torch.cuda.current_device()) if device is None else torch.device(device)
        if self.device.type != "cuda":
            raise RuntimeError("dlrm_amd.Multihot: a GPU device is required (the HIP path has no CPU fallback)")
        lib = _lib.load()
        self.multi_hot_tables_l: List[torch.Tensor] = []
        for t, (n, h) in enumerate(zip(self.num_embeddings_per_feature, self.multi_hot_sizes)):
            if host_tables is not None:
                tab = torch.from_numpy(np.ascontiguousarray(host_tables[t], dtype=np.int32)).to(self.device)
                if tuple(tab.shape) != (n, h):
                    raise ValueError("host table %d has shape %s, expected (%d, %d)" % (t, tuple(tab.shape), n, h))
            else:
                tab = torch.empty((n, h), dtype=torch.int32, device=self.device)
                _lib.check(lib.dlrm_multihot_gen_table(t, n, h, 0 if dist_type == "uniform" else 1, C.c_uint64(seed),
                                                       C.c_void_p(tab.data_ptr()), ops._stream(tab)), "dlrm_multihot_gen_table")
            self.multi_hot_tables_l.append(tab)
        self._tables = _lib.ptr_array([t.data_ptr() for t in self.multi_hot_tables_l])
        self._rows = _lib.i64_array(self.num_embeddings_per_feature)
        self._hot = (C.c_int * len(self.multi_hot_sizes))(*self.multi_hot_sizes)
        self.lookups_per_sample = sum(self.multi_hot_sizes)

    @classmethod
    def from_host_tables(cls, tables: Sequence[np.ndarray], batch_size: int, device=None) -> "Multihot":
        """lookup tables produced elsewhere (e.g. by the reference's own class, `multi_hot_tables_l`) uploaded as they are"""
        return cls([t.shape[1] for t in tables], [t.shape[0] for t in tables], batch_size, device=device, host_tables=tables)

    def expand(self, ids: torch.Tensor, want_global_offsets: bool = True):
        """ids: [T, B] (or flat [T*B], key-major like the KJT `_values` the reference reshapes) int32/int64 1-hot ids.
        Returns (values int32 [B * sum(h)], offsets int64 [T*B + 1] or None, local_offsets int32 [T, B])."""
        T = len(self.multi_hot_sizes)
        if not ids.is_cuda or ids.dtype not in (torch.int32, torch.int64):
            raise RuntimeError("dlrm_amd.Multihot: ids must be an int32/int64 GPU tensor")
        ids = ids.reshape(T, -1).contiguous()
        B = ids.size(1)
        values = torch.empty(B * self.lookups_per_sample, dtype=torch.int32, device=ids.device)
        off_g = torch.empty(T * B + 1, dtype=torch.int64, device=ids.device) if want_global_offsets else None
        off_l = torch.empty((T, B), dtype=torch.int32, device=ids.device)
        rc = _lib.load().dlrm_multihot_expand(T, B, C.c_void_p(ids.data_ptr()), 64 if ids.dtype == torch.int64 else 32,
                                              self._tables, self._rows, self._hot, C.c_void_p(values.data_ptr()),
                                              C.c_void_p(off_g.data_ptr()) if off_g is not None else None,
                                              C.c_void_p(off_l.data_ptr()), C.c_void_p(ops._err_block(ids.device).data_ptr()),
                                              ops._stream(ids))
        _lib.check(rc, "dlrm_multihot_expand")
        if self.lookups_per_sample == T:                 # every hot size is 1: multihot_expand_kernel wrote off_l[t, b] = b * 1
            ops.mark_one_lookup_per_bag(off_l)
        return values, off_g, off_l

    def to_model_inputs(self, ids: torch.Tensor):
        """(lS_o, lS_i) for DLRM_Net.forward: lS_i = per-table views of the ONE values buffer (no copies), lS_o = per-table
        local bag starts — the form `apply_emb` takes (dlrm_s_pytorch.py:407-462) with int32 indices as torchrec KJTs carry."""
        values, _, off_l = self.expand(ids, want_global_offsets=False)
        B = off_l.size(1)
        lS_i, o = [], 0
        for h in self.multi_hot_sizes:
            lS_i.append(values[o:o + B * h])
            o += B * h
        return list(off_l.unbind(0)), lS_i
