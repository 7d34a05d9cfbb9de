#!/usr/bin/env python3
"""arith "bf16x6" per layer shape: the planes kernel (dlrm_gemm_bf16x6 / dlrm_linear_bwd_weight_bf16x6, csrc/gemm_bf16.hip PL = 3) against the kernels
that split fp32 operands in their k-loops (dlrm_linear_fwd / _bwd_weight with DLRM_ARITH_BF16X6) and against native fp32 MFMA.  TFLOP/s are
fp32-EQUIVALENT (2 M N K per product; the matrix pipe executes six bf16 MFMAs for it).  Tuning aid: python tools/bf16x6_gemm_bench.py"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

SHAPES = [(65536, 1024, 1024), (65536, 1024, 512), (65536, 512, 1024), (65536, 1024, 480), (65536, 480, 1024), (65536, 512, 256), (65536, 256, 512)]


def main():
    from dlrm_amd import ops
    from tools.microbench import timeit
    dev = torch.device("cuda:0")
    for M, N, K in SHAPES:
        X = torch.randn(M, K, device=dev)
        W = torch.randn(N, K, device=dev) * 0.03
        bias = torch.randn(N, device=dev)
        Y = torch.empty(M, N, device=dev)
        bits = ops.relu_bits_alloc(M, N, dev)
        X3, W3 = ops.split_bf16x3(X, K), ops.split_bf16x3(W, K)
        Y3 = torch.empty(3, M, N, dtype=torch.bfloat16, device=dev)
        fl = 2.0 * M * N * K
        row = {}
        for tag, fn in (("planes f32+planes", lambda: ops.gemm_bf16x6(X3, W3, bias, 1, Y, Y3, relu_bits_out=bits)),
                        ("planes only", lambda: ops.gemm_bf16x6(X3, W3, bias, 1, None, Y3, relu_bits_out=bits)),
                        ("planes f32 only", lambda: ops.gemm_bf16x6(X3, W3, bias, 1, Y, None, relu_bits_out=bits)),
                        ("in-loop split", lambda: ops.linear_fwd(X, W, bias, 1, Y, "bf16x6", relu_bits=bits)),
                        ("fp32 mfma", lambda: ops.linear_fwd(X, W, bias, 1, Y, "f32", relu_bits=bits)),
                        ("split X", lambda: ops.split_bf16x3(X, K))):
            t = timeit(fn, iters=20)
            row[tag] = (t * 1e3, fl / t / 1e9)
        dZ = torch.randn(M, N, device=dev)
        dZ3 = ops.split_bf16x3(dZ, N)
        dW, db = torch.empty(N, K, device=dev), torch.empty(N, device=dev)
        for tag, fn in (("wgrad planes", lambda: ops.linear_bwd_weight_bf16x6(dZ3, X3, dW, db)),
                        ("wgrad in-loop", lambda: ops.linear_bwd_weight(dZ, X, dW, db, arith="bf16x6")),
                        ("wgrad fp32", lambda: ops.linear_bwd_weight(dZ, X, dW, db, arith="f32"))):
            t = timeit(fn, iters=20)
            row[tag] = (t * 1e3, fl / t / 1e9)
        print("%-20s" % ("%dx%dx%d" % (M, N, K)), " | ".join("%s %6.1f us %6.1f TF" % (k, v[0], v[1]) for k, v in row.items()), flush=True)
        del X, W, Y, X3, W3, Y3, dZ, dZ3


if __name__ == "__main__":
    main()
