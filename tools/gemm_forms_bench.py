#!/usr/bin/env python3
"""The fp32 GEMM of this library alone: the six MFMA layer shapes of the Criteo-Terabyte step x {forward, data gradient (sign bits),
weight gradient}, warm, microseconds per call + a bit checksum of every result (two builds / schedules that claim to be bit-identical
must print equal checksums) + the max error against torch.mm.  One line per (shape, form); the last line sums the 18 launches.

    [DLRM_HIP_LIB=.../libdlrm_hip_tuning.so DLRM_GEMM_SCHED=222] python tools/gemm_forms_bench.py [--check]

Tuning aid; not part of the product."""
import argparse
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from dlrm_amd import ops  # noqa: E402
from tools.microbench import timeit  # noqa: E402

DEV = torch.device("cuda:0")
B = 65536
LAYERS = [("bot512->256", B, 256, 512), ("bot256->128", B, 128, 256), ("top480->1024", B, 1024, 480), ("top1024->1024", B, 1024, 1024),
          ("top1024->512", B, 512, 1024), ("top512->256", B, 256, 512)]


def checksum(t):
    return int(t.contiguous().view(torch.int32).to(torch.int64).sum().item()) & 0xffffffffffff


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--check", action="store_true", help="also compare against torch.mm (slower)")
    ap.add_argument("--iters", type=int, default=20)
    a = ap.parse_args()
    torch.manual_seed(5)
    tot = {"fwd": 0.0, "dgrad": 0.0, "wgrad": 0.0}
    for name, M, N, K in LAYERS:
        X = torch.randn(M, K, device=DEV)
        W = torch.randn(N, K, device=DEV) * 0.03
        b = torch.randn(N, device=DEV)
        Y = torch.empty(M, N, device=DEV)
        dY = torch.randn(M, N, device=DEV)
        dX = torch.empty(M, K, device=DEV)
        dW = torch.empty(N, K, device=DEV)
        db = torch.zeros(N, device=DEV)
        bits_x = ops.relu_bits_alloc(M, K, DEV)
        Xf = torch.empty(M, K, device=DEV)
        ops.linear_fwd(torch.randn(M, 64, device=DEV), torch.randn(K, 64, device=DEV), None, 1, Xf, "f32", relu_bits=bits_x)
        bits_y = ops.relu_bits_alloc(M, N, DEV)
        fl = 2.0 * M * N * K
        forms = {"fwd": (lambda: ops.linear_fwd(X, W, b, 1, Y, "f32", relu_bits=bits_y), Y),
                 "dgrad": (lambda: ops.linear_bwd_data(dY, W, Xf, 1, dX, "f32", relu_bits=bits_x), dX),
                 "wgrad": (lambda: ops.linear_bwd_weight(dY, X, dW, db, arith="f32"), dW)}
        for form, (fn, out) in forms.items():
            t = timeit(fn, iters=a.iters) * 1e3
            tot[form] += t
            err = ""
            if a.check:
                torch.backends.cuda.matmul.allow_tf32 = False
                ref = {"fwd": lambda: torch.relu(X @ W.t() + b), "dgrad": lambda: (dY @ W) * (Xf > 0), "wgrad": lambda: dY.t() @ X}[form]()
                err = " maxerr %.2e" % float((out - ref).abs().max() / ref.abs().max())
                if form == "wgrad":
                    err += " db %.2e" % float((db - dY.sum(0)).abs().max() / dY.sum(0).abs().max())
            print("%-14s %-6s %8.1f us %6.1f TF  sum %012x%s" % (name, form, t, fl / t / 1e6, checksum(out), err), flush=True)
        del X, W, Y, dY, dX, Xf
    print("TOTAL fwd %.1f dgrad %.1f wgrad %.1f all %.1f us  [SCHED=%s]" % (tot["fwd"], tot["dgrad"], tot["wgrad"], sum(tot.values()),
                                                                         os.environ.get("DLRM_GEMM_SCHED", "-")), flush=True)


if __name__ == "__main__":
    main()
