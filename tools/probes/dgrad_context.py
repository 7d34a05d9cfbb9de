#!/usr/bin/env python3
"""dgrad_context.py — inside the step the data-gradient GEMMs are 4-38 % slower than launched alone (rocprofv3 trace vs
tools/microbench.py), the weight-gradient GEMMs are not.  What in front of a data-gradient launch costs it time?  One layer's dgrad is
timed (events around every launch) alone, after its own wgrad (the step's order), after an unrelated 512 MB copy, and after a GEMM of
another template instance.  Tuning aid; not part of the product."""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from dlrm_amd import ops  # noqa: E402

DEV = torch.device("cuda:0")
M = 65536


def run(N, K):
    # layer K -> N (forward X[M,K] -> Y[M,N]); dgrad: dY[M,N] @ W[N,K] -> dX[M,K] masked by the sign bits of X
    X = torch.empty(M, K, device=DEV)
    bits = ops.relu_bits_alloc(M, K, DEV)
    ops.linear_fwd(torch.randn(M, 64, device=DEV), torch.randn(K, 64, device=DEV), None, 1, X, "f32", relu_bits=bits)
    W = torch.randn(N, K, device=DEV) * 0.03
    dY, dX = torch.randn(M, N, device=DEV), torch.empty(M, K, device=DEV)
    dW, db = torch.empty(N, K, device=DEV), torch.empty(N, device=DEV)
    ja, jb = torch.empty(128 << 20, device=DEV), torch.empty(128 << 20, device=DEV)
    Xo, Wo, Yo = torch.randn(M, 256, device=DEV), torch.randn(128, 256, device=DEV), torch.empty(M, 128, device=DEV)
    dgrad = lambda: ops.linear_bwd_data(dY, W, X, 1, dX, "f32", relu_bits=bits)   # noqa: E731
    ctx = {"alone": None, "after its wgrad": lambda: ops.linear_bwd_weight(dY, X, dW, db, arith="f32"),
           "after a 512 MB copy": lambda: jb.copy_(ja), "after another GEMM instance": lambda: ops.linear_fwd(Xo, Wo, None, 1, Yo, "f32")}
    out = []
    for name, before in ctx.items():
        t_end = time.perf_counter() + 0.06
        while time.perf_counter() < t_end:
            if before:
                before()
            dgrad()
            torch.cuda.synchronize()
        evs = []
        for _ in range(20):
            if before:
                before()
            a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            a.record(); dgrad(); b.record()
            evs.append((a, b))
        torch.cuda.synchronize()
        out.append("%s %.1f" % (name, sum(a.elapsed_time(b) for a, b in evs) / len(evs) * 1e3))
    print("dgrad of %4d -> %-4d | " % (K, N) + " | ".join(out), flush=True)


for N, K in ((1024, 1024), (512, 1024), (256, 512), (128, 256)):
    run(N, K)
