#!/usr/bin/env python3
"""The vendor library's GEMM (torch.mm -> rocBLAS / hipBLASLt Tensile kernels) beside this library's, per layer shape of the headline
step and per form (forward Y = X W^T, data gradient dX = dY W, weight gradient dW = dY^T X), on the operand layouts each form really has.

    python tools/vendor_vs_own.py [--dtype f32|bf16] [--md out.md] [--names]

--names runs every vendor call ONCE with a marker print so that `rocprofv3 --kernel-trace --stats -- python tools/vendor_vs_own.py --names`
lists the Tensile kernel picked per shape (macro tile, depth-U, MI shape, workgroup are in the kernel name).
Tuning / evidence aid: not part of the product, not imported by bench.py."""
import argparse
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from dlrm_amd import ops  # noqa: E402
from tools.microbench import timeit  # noqa: E402

DEV = torch.device("cuda:0")
B = 65536
# (M, N, K) of nn.Linear(K -> N) on a batch of M rows: the seven MFMA layers of the Criteo-Terabyte step (13 -> 512 and 256 -> 1 are not GEMMs here)
LAYERS = [("bot 512->256", B, 256, 512), ("bot 256->128", B, 128, 256), ("top 480->1024", B, 1024, 480), ("top 1024->1024", B, 1024, 1024),
          ("top 1024->512", B, 512, 1024), ("top 512->256", B, 256, 512)]


def backends():
    out = []
    for lib in ("cublas", "cublaslt"):                     # torch's names: rocBLAS, hipBLASLt
        try:
            torch.backends.cuda.preferred_blas_library(lib)
            out.append(lib)
        except Exception as e:                             # noqa: BLE001
            print("backend %s unavailable: %s" % (lib, e), flush=True)
    return out


def fp32(md, names):
    torch.backends.cuda.matmul.allow_tf32 = False
    libs = backends()
    rows = []
    for name, M, N, K in LAYERS:
        X = torch.randn(M, K, device=DEV)
        W = torch.randn(N, K, device=DEV) * 0.03
        b = torch.randn(N, device=DEV)
        Y = torch.empty(M, N, device=DEV)
        dY = torch.randn(M, N, device=DEV)
        dX = torch.empty(M, K, device=DEV)
        dW = torch.empty(N, K, device=DEV)
        db = torch.zeros(N, device=DEV)
        bits = ops.relu_bits_alloc(M, K, DEV)
        Xf = torch.empty(M, K, device=DEV)
        ops.linear_fwd(torch.randn(M, 64, device=DEV), torch.randn(K, 64, device=DEV), None, 1, Xf, "f32", relu_bits=bits)
        fl = 2.0 * M * N * K
        vend = {}
        for lib in libs:
            torch.backends.cuda.preferred_blas_library(lib)
            calls = {"fwd": lambda: torch.mm(X, W.t(), out=Y),           # W^T as a view: the library reads W [N, K] k-contiguous, as ours does
                     "dgrad": lambda: torch.mm(dY, W, out=dX),
                     "wgrad": lambda: torch.mm(dY.t(), X, out=dW)}
            for form, fn in calls.items():
                if names:
                    fn(); torch.cuda.synchronize()
                    continue
                vend[(lib, form)] = timeit(fn, iters=20) * 1e3
        if names:
            continue
        bits_y = ops.relu_bits_alloc(M, N, DEV)
        own = {"fwd": timeit(lambda: ops.linear_fwd(X, W, b, 1, Y, "f32", relu_bits=bits_y), iters=20) * 1e3,
               "dgrad": timeit(lambda: ops.linear_bwd_data(dY, W, Xf, 1, dX, "f32", relu_bits=bits), iters=20) * 1e3,
               "wgrad": timeit(lambda: ops.linear_bwd_weight(dY, X, dW, db, arith="f32"), iters=20) * 1e3}
        for form in ("fwd", "dgrad", "wgrad"):
            best = min(vend[(lib, form)] for lib in libs)
            rows.append((name, "%dx%dx%d" % (M, N, K), form, own[form], fl / own[form] / 1e6,
                         [vend[(lib, form)] for lib in libs], fl / best / 1e6, own[form] / best))
            print("%-15s %-6s own %7.1f us %6.1f TF | " % (name, form, own[form], fl / own[form] / 1e6) +
                  " ".join("%s %7.1f us" % (lib, vend[(lib, form)]) for lib in libs) + " | own/vendor %.3f" % (own[form] / best), flush=True)
        del X, W, Y, dY, dX, Xf
    if md and rows:
        with open(md, "w") as f:
            f.write("| layer | M x N x K | form | own µs (epilogue fused) | own TF | " + " | ".join("vendor %s µs (bare GEMM)" % l for l in libs) + " | vendor TF | own / vendor |\n")
            f.write("|---|---|---|---:|---:|" + "---:|" * len(libs) + "---:|---:|\n")
            for r in rows:
                f.write("| %s | %s | %s | %.1f | %.1f | " % r[:5] + " | ".join("%.1f" % v for v in r[5]) + " | %.1f | %.3f |\n" % (r[6], r[7]))
            so, sv = sum(r[3] for r in rows), sum(min(r[5]) for r in rows)
            f.write("\nsum over the 18 launches: own %.1f µs, vendor (best backend per launch) %.1f µs, ratio %.3f\n" % (so, sv, so / sv))


def bf16(md, names):
    libs = backends()
    shapes = [("1024x1024", B, 1024, 1024), ("1024->512", B, 512, 1024), ("512->1024", B, 1024, 512), ("512->256", B, 256, 512),
              ("256->512", B, 512, 256), ("3456->512 (dcn)", B, 512, 3456), ("512->3456 (dcn)", B, 3456, 512)]
    rows = []
    for name, M, N, K in shapes:
        A = torch.randn(M, K, device=DEV).to(torch.bfloat16)
        Wt = (torch.randn(N, K, device=DEV) * 0.03).to(torch.bfloat16)
        bias = torch.randn(N, device=DEV)
        Cb = torch.empty(M, N, dtype=torch.bfloat16, device=DEV)
        dZ = torch.randn(M, N, device=DEV).to(torch.bfloat16)
        dWb = torch.empty(N, K, dtype=torch.bfloat16, device=DEV)
        fl = 2.0 * M * N * K
        vend = {}
        for lib in libs:
            torch.backends.cuda.preferred_blas_library(lib)
            calls = {"fwd": lambda: torch.mm(A, Wt.t(), out=Cb), "wgrad": lambda: torch.mm(dZ.t(), A, out=dWb)}
            for form, fn in calls.items():
                if names:
                    fn(); torch.cuda.synchronize()
                    continue
                vend[(lib, form)] = timeit(fn, iters=20) * 1e3
        if names:
            continue
        bits = ops.relu_bits_alloc(M, N, DEV)
        own = {"fwd": timeit(lambda: ops.gemm_bf16(A, Wt, bias, 1, None, Cb, relu_bits_out=bits), iters=20) * 1e3}
        dW, db = torch.empty(N, K, device=DEV), torch.empty(N, device=DEV)
        if ops.linear_bwd_weight_bf16_ok(M, N, K, dZ, A):
            own["wgrad"] = timeit(lambda: ops.linear_bwd_weight_bf16(dZ, A, dW, db), iters=20) * 1e3
        for form in own:
            best = min(vend[(lib, form)] for lib in libs)
            rows.append((name, "%dx%dx%d" % (M, N, K), form, own[form], fl / own[form] / 1e6, [vend[(lib, form)] for lib in libs], fl / best / 1e6, own[form] / best))
            print("%-16s %-6s own %7.1f us %7.1f TF | " % (name, form, own[form], fl / own[form] / 1e6) +
                  " ".join("%s %7.1f us %7.1f TF" % (lib, vend[(lib, form)], fl / vend[(lib, form)] / 1e6) for lib in libs) + " | own/vendor %.3f" % (own[form] / best), flush=True)
        del A, Wt, Cb, dZ
    if md and rows:
        with open(md, "w") as f:
            f.write("| layer | M x N x K | form | own µs (bias + ReLU + sign bits fused; wgrad: fp32 dW + db) | own TF | " + " | ".join("vendor %s µs (bare, bf16 out)" % l for l in libs) + " | vendor TF | own / vendor |\n")
            f.write("|---|---|---|---:|---:|" + "---:|" * len(libs) + "---:|---:|\n")
            for r in rows:
                f.write("| %s | %s | %s | %.1f | %.1f | " % r[:5] + " | ".join("%.1f" % v for v in r[5]) + " | %.1f | %.3f |\n" % (r[6], r[7]))


if __name__ == "__main__":
    ap = argparse.ArgumentParser()
    ap.add_argument("--dtype", default="f32", choices=["f32", "bf16"])
    ap.add_argument("--md", default=None)
    ap.add_argument("--names", action="store_true")
    a = ap.parse_args()
    (fp32 if a.dtype == "f32" else bf16)(a.md, a.names)
