"""Criteo-Terabyte binary data reader with the conversion done on the device.

Host-side mirror of `CriteoBinDataset` (data_loader_terabyte.py:197-251): the file is a sequence of records of 40
little-endian int32 (label, 13 dense counts, 26 categorical ids); batch `i` is the byte range
`[i * 160 * batch_size, (i + 1) * 160 * batch_size)` (the last batch may be short), `len()` = ceil(size / bytes per
batch).  The reference converts every batch on the host (`_transform_features`, :74-93) and the training loop then copies
four tensors to the GPU.  Here the RAW block is read into pinned host memory, copied once (10.5 MB at B = 65536 — 42 %
of the converted batch), and `dlrm_criteo_bin_transform` writes X / offsets / indices / targets in HBM; batch i+1 is read
and copied on a side stream while batch i trains (double buffering).

Output per batch, exactly the reference's: X [B, 13] f32 = log(dense + 1), lS_o [26, B] (arange rows), lS_i [26, B] =
ids % max_ind_range, T [B, 1] f32 — as stacked device tensors.
"""
from __future__ import annotations

import ctypes as C
import math
import os
from typing import Iterator, Tuple

import numpy as np
import torch

from . import _lib, ops

TOT_FEA, DEN_FEA, SPA_FEA = 40, 13, 26          # data_loader_terabyte.py:209-213


def batch_byte_range(file_bytes: int, batch_size: int, i: int, bytes_per_feature: int = 4) -> Tuple[int, int]:
    """[start, end) of batch i in the file, the reference's seek/read arithmetic (:233-236)."""
    per = bytes_per_feature * TOT_FEA * batch_size
    start = i * per
    return start, min(start + per, file_bytes)


def num_batches(file_bytes: int, batch_size: int, bytes_per_feature: int = 4) -> int:
    return math.ceil(file_bytes / (bytes_per_feature * TOT_FEA * batch_size))          # :222


class CriteoBinBatches:
    def __init__(self, data_file: str, batch_size: int, max_ind_range: int = -1, device=None,
                 index_dtype: torch.dtype = torch.int64, prefetch: bool = True):
        if index_dtype not in (torch.int64, torch.int32):
            raise RuntimeError("dlrm_amd.criteo_bin: index dtype must be int64 or int32")
        self.path, self.batch_size, self.max_ind_range = data_file, int(batch_size), int(max_ind_range)
        self.file_bytes = os.path.getsize(data_file)
        if self.file_bytes % (4 * TOT_FEA) != 0:
            raise RuntimeError("dlrm_amd.criteo_bin: %s is not a whole number of 160-byte records" % data_file)
        self.device = torch.device(device if device is not None else "cuda:0")
        if self.device.type != "cuda":
            raise RuntimeError("dlrm_amd.criteo_bin: the conversion kernel runs on the GPU only (no CPU fallback)")
        self.index_dtype, self.prefetch = index_dtype, prefetch
        self._mm = np.memmap(data_file, dtype=np.int32, mode="r")
        self._pinned = [torch.empty((self.batch_size, TOT_FEA), dtype=torch.int32).pin_memory() for _ in range(2)]
        self._raw = [torch.empty((self.batch_size, TOT_FEA), dtype=torch.int32, device=self.device) for _ in range(2)]
        self._copy_stream = torch.cuda.Stream(device=self.device)
        self._ready = [None, None]          # events: raw block of slot s has landed in HBM
        self._free = [None, None]           # events: the transform that read slot s has been enqueued and finished

    def __len__(self) -> int:
        return num_batches(self.file_bytes, self.batch_size)

    def _rows(self, i: int) -> int:
        s, e = batch_byte_range(self.file_bytes, self.batch_size, i)
        return (e - s) // (4 * TOT_FEA)

    def _stage(self, i: int, slot: int) -> int:
        """file -> pinned -> device raw buffer of `slot`, on the copy stream."""
        n = self._rows(i)
        s, _ = batch_byte_range(self.file_bytes, self.batch_size, i)
        if self._free[slot] is not None:
            self._free[slot].synchronize()                         # the pinned + raw buffers of this slot are reusable
        # page cache -> pinned staging buffer, one memcpy (the numpy view aliases the pinned tensor)
        self._pinned[slot].numpy()[:n] = self._mm[s // 4:s // 4 + n * TOT_FEA].reshape(n, TOT_FEA)
        with torch.cuda.stream(self._copy_stream):
            self._raw[slot][:n].copy_(self._pinned[slot][:n], non_blocking=True)
            ev = torch.cuda.Event()
            ev.record()
        self._ready[slot] = ev
        return n

    def _transform(self, slot: int, n: int):
        lib = _lib.load()
        dev = self.device
        cur = torch.cuda.current_stream(dev)
        cur.wait_event(self._ready[slot])
        X = torch.empty((n, DEN_FEA), dtype=torch.float32, device=dev)
        idx = torch.empty((SPA_FEA, n), dtype=self.index_dtype, device=dev)
        off = torch.empty((SPA_FEA, n), dtype=self.index_dtype, device=dev)
        tgt = torch.empty((n, 1), dtype=torch.float32, device=dev)
        rc = lib.dlrm_criteo_bin_transform(n, C.c_void_p(self._raw[slot].data_ptr()), self.max_ind_range,
                                           64 if self.index_dtype == torch.int64 else 32, C.c_void_p(X.data_ptr()), DEN_FEA,
                                           C.c_void_p(idx.data_ptr()), C.c_void_p(off.data_ptr()), n,
                                           C.c_void_p(tgt.data_ptr()), C.c_void_p(cur.cuda_stream))
        _lib.check(rc, "dlrm_criteo_bin_transform")
        ev = torch.cuda.Event()
        ev.record(cur)
        self._free[slot] = ev
        # criteo_bin_transform_kernel writes off[t, b] = b: the producer's proof of "one lookup per bag" (no device pass / sync later)
        ops.mark_one_lookup_per_bag(off)
        return X, off, idx, tgt

    def batch(self, i: int):
        """Batch i alone (random access, like CriteoBinDataset.__getitem__)."""
        if not 0 <= i < len(self):
            raise IndexError(i)
        n = self._stage(i, 0)
        return self._transform(0, n)

    def __iter__(self) -> Iterator:
        nb = len(self)
        if nb == 0:
            return
        n_next = self._stage(0, 0)
        for i in range(nb):
            slot, n = i & 1, n_next
            if self.prefetch and i + 1 < nb:
                n_next = self._stage(i + 1, slot ^ 1)              # overlaps the training step of batch i
            out = self._transform(slot, n)
            if not self.prefetch and i + 1 < nb:
                n_next = self._stage(i + 1, slot ^ 1)
            yield out
