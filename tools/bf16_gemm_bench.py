#!/usr/bin/env python3
"""dlrm_gemm_bf16 per shape: the bf16-shaped phased kernel (csrc/gemm_bf16.hip) vs the fp32-shaped one (DLRM_BF16_PHASED=0, a second
process), with fp32 + bf16 outputs / bf16 only / none-but-bits.  Tuning aid: python tools/bf16_gemm_bench.py [--json out]"""
import json
import os
import subprocess
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

SHAPES = [(65536, 1024, 1024), (65536, 1024, 512), (65536, 512, 1024), (65536, 512, 512), (65536, 512, 256), (65536, 256, 512), (65536, 512, 3456),
          (65536, 3456, 512), (65536, 1024, 3456), (8192, 1024, 1024)]


def main():
    global SHAPES
    if os.environ.get("BF16_BENCH_SHAPES"):         # ablation runs: the two layer shapes that matter
        SHAPES = [(65536, 1024, 1024), (65536, 512, 3456)] if os.environ["BF16_BENCH_SHAPES"] != "3" else [(65536, 1024, 1024), (65536, 3456, 512)]
    from dlrm_amd import ops
    from tools.microbench import timeit
    dev = torch.device("cuda:0")
    out = {}
    for M, N, K in SHAPES:
        A = (torch.randn(M, K, device=dev)).to(torch.bfloat16)
        B = (torch.randn(N, K, device=dev) * 0.03).to(torch.bfloat16)
        bias = torch.randn(N, device=dev)
        Cf = torch.empty(M, N, device=dev)
        Cb = torch.empty(M, N, dtype=torch.bfloat16, device=dev)
        bits = ops.relu_bits_alloc(M, N, dev)
        fl = 2.0 * M * N * K
        row = {}
        for tag, cf, cb in (("f32+bf16", Cf, Cb), ("bf16", None, Cb), ("f32", Cf, None)):
            t = timeit(lambda: ops.gemm_bf16(A, B, bias, 1, cf, cb, relu_bits_out=bits), iters=20)
            row[tag] = [round(t * 1e3, 1), round(fl / t / 1e9, 1)]       # us, TFLOP/s
        dW, db = torch.empty(N, K, device=dev), torch.empty(N, device=dev)
        dZ = torch.randn(M, N, device=dev).to(torch.bfloat16)
        if ops.linear_bwd_weight_bf16_ok(M, N, K, dZ, A):
            t = timeit(lambda: ops.linear_bwd_weight_bf16(dZ, A, dW, db), iters=20)
            row["wgrad_bf16"] = [round(t * 1e3, 1), round(fl / t / 1e9, 1)]
        out["%dx%dx%d" % (M, N, K)] = row
        print("%-22s" % ("%dx%dx%d" % (M, N, K)), "  ".join("%s %7.1f us %7.1f TF" % (k, v[0], v[1]) for k, v in row.items()), flush=True)
        del A, B, Cf, Cb
    return out


if __name__ == "__main__":
    if os.environ.get("_BF16_BENCH_CHILD") == "1":
        main()
        sys.exit(0)
    for mode in ("1", "0"):
        print("== DLRM_BF16_PHASED=%s" % mode, flush=True)
        subprocess.run([sys.executable, os.path.abspath(__file__)], env=dict(os.environ, DLRM_BF16_PHASED=mode, _BF16_BENCH_CHILD="1"), check=False)
