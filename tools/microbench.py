#!/usr/bin/env python3
"""Per-kernel micro-benchmarks on the GPU box (tuning aid; not part of the product or of bench.py)."""
import argparse
import json
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from dlrm_amd import ops  # noqa: E402

DEV = torch.device("cuda:0")


def timeit(fn, iters=10, warm=3):
    # the clocks fall within milliseconds of an idle queue (allocation, host work) and take ~30 launches to come back: 1134 us for
    # the first dozen 1024x1024 GEMMs of a process, 989 us from then on (tools/probes/alloc_modes.py) — warm up by TIME, not by count
    import time
    t_end = time.perf_counter() + 0.06
    n = 0
    while n < warm or time.perf_counter() < t_end:
        fn()
        n += 1
        if n % 8 == 0:
            torch.cuda.synchronize()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(iters):
        fn()
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) / iters


def gemm(shapes):
    out = []
    for (M, N, K) in shapes:
        ldk = (K + 3) & ~3
        X = torch.randn(M, ldk, device=DEV)[:, :K]
        W = torch.randn(N, K, device=DEV) * 0.03
        b = torch.randn(N, device=DEV)
        Y = torch.empty(M, N, device=DEV)
        dY = torch.randn(M, N, device=DEV)
        dX = torch.empty(M, ldk, device=DEV)[:, :K]
        dW = torch.empty(N, K, device=DEV)
        db = torch.zeros(N, device=DEV)
        fl = 2.0 * M * N * K
        t_lib = float("nan")
        if os.environ.get("MICROBENCH_TORCH", "0") == "1" and K % 4 == 0:
            # the vendor library's fp32 GEMM on the same operands (torch.mm -> hipBLASLt / rocBLAS; no bias, no activation, no TF32)
            torch.backends.cuda.matmul.allow_tf32 = False
            Xc, Wt = X.contiguous(), W.t().contiguous()
            t_lib = timeit(lambda: torch.mm(Xc, Wt, out=Y))
        t_f = timeit(lambda: ops.linear_fwd(X, W, b, 1, Y, ARITH))
        bits = ops.relu_bits_alloc(M, K, DEV) if K % 4 == 0 else None      # sign bits of X as the forward of the previous layer stores them
        if bits is not None:
            Xf = torch.empty(M, K, device=DEV)
            ops.linear_fwd(torch.randn(M, 64, device=DEV), torch.randn(K, 64, device=DEV), None, 1, Xf, ARITH, relu_bits=bits)
        t_db = timeit(lambda: ops.linear_bwd_data(dY, W, Xf, 1, dX, ARITH, relu_bits=bits)) if bits is not None else float("nan")
        t_d = timeit(lambda: ops.linear_bwd_data(dY, W, X, 1, dX, ARITH))
        t_d0 = timeit(lambda: ops.linear_bwd_data(dY, W, None, 0, dX, ARITH))
        t_w = timeit(lambda: ops.linear_bwd_weight(dY, X, dW, db, arith=ARITH))
        r = dict(M=M, N=N, K=K, fwd_us=t_f * 1e3, fwd_tf=fl / t_f / 1e9, dgrad_us=t_d * 1e3, dgrad_tf=fl / t_d / 1e9,
                 dgrad_plain_us=t_d0 * 1e3, dgrad_plain_tf=fl / t_d0 / 1e9, wgrad_us=t_w * 1e3, wgrad_tf=fl / t_w / 1e9)
        out.append(r)
        if t_lib == t_lib:
            print("      vendor library fp32 GEMM (torch.mm, Y = X W^T without bias / activation): %.1f us %.1f TF" % (t_lib * 1e3, fl / t_lib / 1e9), flush=True)
        print("gemm M=%d N=%d K=%d | fwd %.1f us %.1f TF | dgrad bits %.1f us, fp32 mask %.1f us %.1f TF (plain %.1f us %.1f TF) | wgrad %.1f us %.1f TF"
              % (M, N, K, r["fwd_us"], r["fwd_tf"], t_db * 1e3, r["dgrad_us"], r["dgrad_tf"], r["dgrad_plain_us"], r["dgrad_plain_tf"],
                 r["wgrad_us"], r["wgrad_tf"]), flush=True)
    return out


def wgrad(shapes):
    """weight gradient only (tuning sweeps: DLRM_WGRAD_WGS / DLRM_WGRAD_MINROWS / DLRM_WGRAD_TM are read once per process)"""
    tot = 0.0
    for (M, N, K) in shapes:
        ldk = (K + 3) & ~3
        X = torch.randn(M, ldk, device=DEV)[:, :K]
        dY = torch.randn(M, N, device=DEV)
        dW = torch.empty(N, K, device=DEV)
        db = torch.zeros(N, device=DEV)
        t = timeit(lambda: ops.linear_bwd_weight(dY, X, dW, db, arith=ARITH))
        tot += t
        print("wgrad M=%d N=%d K=%d  %.1f us  %.1f TF" % (M, N, K, t * 1e3, 2.0 * M * N * K / t / 1e9), flush=True)
    print("wgrad sum %.1f us  [WGS=%s MINROWS=%s TM=%s]" % (tot * 1e3, os.environ.get("DLRM_WGRAD_WGS", "-"), os.environ.get("DLRM_WGRAD_MINROWS", "-"),
                                                        os.environ.get("DLRM_WGRAD_TM", "-")), flush=True)


def emb(B=65536, D=128, cap=0):
    rows = [39884406, 39043, 17289, 7420, 20263, 3, 7120, 1543, 63, 38532951, 2953546, 403346, 10, 2208, 11938, 155, 4, 976, 14,
            39979771, 25641295, 39664984, 585935, 12972, 108, 36]
    if cap:
        rows = [min(r, cap) for r in rows]
    T = len(rows)
    Ws = [torch.empty(n, D, device=DEV).uniform_(-0.01, 0.01) for n in rows]
    idx = [torch.randint(0, n, (B,), device=DEV) for n in rows]
    off = [torch.arange(B, device=DEV)] * T
    bags = ops.BagBatch(off, idx)
    feat = torch.empty(B, (T + 1) * D, device=DEV)
    dout = torch.randn(B, T * D, device=DEV) * 1e-3
    res = {}
    t = timeit(lambda: ops.emb_fwd(Ws, bags, feat[:, D:]))
    res["emb_fwd_us"] = t * 1e3
    res["emb_fwd_gbs"] = T * B * (4 * D * 2 + 16) / t / 1e6
    for name, mode in (("sorted", ops.UPD_SORTED), ("atomic", ops.UPD_ATOMIC)):
        t = timeit(lambda: ops.emb_bwd_sgd(Ws, bags, dout, 0.01, mode), iters=5, warm=2)
        res[f"emb_bwd_{name}_us"] = t * 1e3
        res[f"emb_bwd_{name}_gbs"] = T * B * (4 * D * 3 + 8) / t / 1e6
    x = torch.randn(B, D, device=DEV)
    feat[:, :D] = x
    R = torch.empty(B, 480, device=DEV)
    t = timeit(lambda: ops.interact_fwd([feat[:, :D], feat[:, D:]], D, False, R))
    res["interact_fwd_us"] = t * 1e3
    dR = torch.randn(B, 480, device=DEV)
    dfeat = torch.empty(B, (T + 1) * D, device=DEV)
    t = timeit(lambda: ops.interact_bwd([feat[:, :D], feat[:, D:]], D, False, dR, [dfeat[:, :D], dfeat[:, D:]]))
    res["interact_bwd_us"] = t * 1e3
    print(json.dumps(res), flush=True)
    return res


def emb_classes(B=65536, D=128):
    """sorted / atomic fused update per table class (where does the time go?)"""
    rows_all = [39884406, 39043, 17289, 7420, 20263, 3, 7120, 1543, 63, 38532951, 2953546, 403346, 10, 2208, 11938, 155, 4, 976, 14,
                39979771, 25641295, 39664984, 585935, 12972, 108, 36]
    classes = {"huge(>=400k rows)": [r for r in rows_all if r >= 400000],
               "mid(1k..40k rows)": [r for r in rows_all if 1000 <= r < 400000],
               "tiny(<1k rows)": [r for r in rows_all if r < 1000],
               "all": rows_all}
    for name, rows in classes.items():
        T = len(rows)
        Ws = [torch.empty(n, D, device=DEV).uniform_(-0.01, 0.01) for n in rows]
        idx = [torch.randint(0, n, (B,), device=DEV) for n in rows]
        off = [torch.arange(B, device=DEV)] * T
        bags = ops.BagBatch(off, idx)
        dout = torch.randn(B, T * D, device=DEV) * 1e-3
        res = {"class": name, "tables": T}
        for mname, mode in (("sorted", ops.UPD_SORTED), ("atomic", ops.UPD_ATOMIC)):
            t = timeit(lambda: ops.emb_bwd_sgd(Ws, bags, dout, 0.01, mode), iters=5, warm=2)
            res[mname + "_us"] = round(t * 1e3, 1)
            res[mname + "_gbs"] = round(T * B * (4 * D * 3 + 8) / t / 1e6, 1)
        print(json.dumps(res), flush=True)
        del Ws, idx, dout


ARITH = "f32"


if __name__ == "__main__":
    ap = argparse.ArgumentParser()
    ap.add_argument("what", choices=["gemm", "gemm_big", "wgrad", "emb", "emb_classes", "all"])
    ap.add_argument("--arith", default="f32")
    a = ap.parse_args()
    ARITH = a.arith
    B = 65536
    layer_shapes = [(B, 512, 16), (B, 256, 512), (B, 128, 256), (B, 1024, 480), (B, 1024, 1024), (B, 512, 1024), (B, 256, 512), (B, 1, 256)]
    if a.what in ("gemm", "all"):
        gemm(layer_shapes)
    if a.what == "wgrad":
        wgrad(layer_shapes[1:7])
    if a.what == "gemm_big":
        gemm([(B, 1024, 1024), (B, 512, 1024)])
    if a.what in ("emb", "all"):
        emb()
    if a.what == "emb_classes":
        emb_classes()
