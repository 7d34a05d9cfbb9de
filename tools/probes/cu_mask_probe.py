#!/usr/bin/env python3
"""CU-partitioned composition: the sort-based embedding update (HBM-bound, Criteo-Terabyte shapes) on a stream restricted to k
CUs beside a 1024 x 1024 fp32 MFMA GEMM chain on a stream restricted to the other 256 - k CUs (hipExtStreamCreateWithCUMask), against the two
run back to back on the whole chip.  Also each alone on k CUs: how the HBM-bound kernel scales with its CU count."""
import ctypes as C
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from dlrm_amd import _lib, ops  # noqa: E402

ROWS = [39884406, 39043, 17289, 7420, 20263, 3, 7120, 1543, 63, 38532951, 2953546, 403346, 10, 2208, 11938, 155,
        4, 976, 14, 39979771, 25641295, 39664984, 585935, 12972, 108, 36]
dev = torch.device("cuda:0")
lib = _lib.load()
B, D = 65536, 128
cap = 4_000_000
rows = [min(r, cap) for r in ROWS]
g = torch.Generator(device=dev).manual_seed(1)
Ws = [torch.randn(n, D, device=dev) * 0.01 for n in rows]
idx = torch.stack([torch.randint(0, n, (B,), device=dev, generator=g) for n in rows])
off = torch.arange(B, device=dev).repeat(len(rows), 1)
bags = ops.BagBatch(off, idx)
dout = torch.randn(B, len(rows) * D, device=dev)
X = torch.randn(B, 1024, device=dev)
W = torch.randn(1024, 1024, device=dev) * 0.03
bias = torch.randn(1024, device=dev)
Y = torch.empty(B, 1024, device=dev)
NG = 3                        # GEMMs per "MLP phase": ~3 ms of MFMA work beside ~0.5 ms of update


def mk(first, count):
    p = C.c_void_p()
    _lib.check(lib.dlrm_stream_create_cu_range(first, count, C.byref(p)), "dlrm_stream_create_cu_range")
    return torch.cuda.ExternalStream(p.value, device=dev), p


def upd():
    ops.emb_bwd_sgd(Ws, bags, dout, 0.01, ops.UPD_SORTED)


def mlp():
    for _ in range(NG):
        ops.linear_fwd(X, W, bias, 1, Y, "f32")


def timed(fn, n=10):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n * 1e3


for f in (upd, mlp):
    for _ in range(3):
        f()
print("whole chip, one stream: update %.3f ms, %d GEMMs %.3f ms, back to back %.3f ms" % (timed(upd), NG, timed(mlp), timed(lambda: (upd(), mlp()))))
main = torch.cuda.current_stream()
for k in (32, 64, 96, 128):
    sa, pa = mk(0, k)
    sb, pb = mk(k, 256 - k)

    def on(stream, fn):
        def run():
            stream.wait_stream(main)
            with torch.cuda.stream(stream):
                fn()
            main.wait_stream(stream)
        return run

    def both():
        sa.wait_stream(main); sb.wait_stream(main)
        with torch.cuda.stream(sa):
            upd()
        with torch.cuda.stream(sb):
            mlp()
        main.wait_stream(sa); main.wait_stream(sb)

    def both_unmasked_main():                      # update on k CUs, GEMMs on torch's ordinary (unmasked) stream
        sa.wait_stream(main)
        with torch.cuda.stream(sa):
            upd()
        mlp()
        main.wait_stream(sa)
    for f in (on(sa, upd), on(sb, mlp), both, both_unmasked_main):
        f()
    print("k = %3d: update alone on k CUs %.3f ms | GEMMs alone on %d CUs %.3f ms | partitioned together %.3f ms | update on k CUs beside unmasked GEMMs %.3f ms"
          % (k, timed(on(sa, upd)), 256 - k, timed(on(sb, mlp)), timed(both), timed(both_unmasked_main)), flush=True)
    del sa, sb
    lib.dlrm_stream_destroy(pa); lib.dlrm_stream_destroy(pb)
