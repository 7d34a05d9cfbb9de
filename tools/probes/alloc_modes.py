#!/usr/bin/env python3
"""alloc_modes.py — the same GEMM (M = 65536, 1024 -> 1024 forward) runs at 995 us or at 1120 us from one process start to the next.
Is that the PLACEMENT of its operands?  X, W, Y are carved out of one pool at controlled offsets; several pools (fresh hipMallocs
while the earlier ones stay alive) and several offset patterns are timed in one process.  Tuning aid; not part of the product."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from dlrm_amd import ops  # noqa: E402

DEV = torch.device("cuda:0")
M, N, K = 65536, 1024, 1024
MB = 1 << 20


def view(pool, off_bytes, rows, cols):
    n = rows * cols
    return pool[off_bytes // 4: off_bytes // 4 + n].view(rows, cols)


def timeit(fn, iters=12):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(iters):
        fn()
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) / iters * 1e3


def main():
    pools = []
    Wsrc = torch.randn(N, K, device=DEV) * 0.03
    bias = torch.randn(N, device=DEV)
    for p in range(4):
        pool = torch.empty(1100 * MB // 4, dtype=torch.float32, device=DEV)       # 1.1 GB: X (256 MB) + Y (256 MB) + slack
        pool.normal_()
        pools.append(pool)
        line = []
        for name, xo, yo in (("X@0 Y@256M", 0, 256 * MB), ("X@0 Y@512M", 0, 512 * MB), ("X@0 Y@256M+2M", 0, 258 * MB),
                             ("X@0 Y@256M+64K", 0, 256 * MB + 65536), ("X@4K Y@300M", 4096, 300 * MB), ("X@64M Y@700M", 64 * MB, 700 * MB)):
            X, Y = view(pool, xo, M, K), view(pool, yo, M, N)
            t = timeit(lambda: ops.linear_fwd(X, Wsrc, bias, 1, Y, "f32"))
            line.append("%s %.0f" % (name, t))
        print("pool %d @%#x | " % (p, pool.data_ptr()) + " | ".join(line), flush=True)
    # separate torch allocations, several times
    for r in range(6):
        X = torch.randn(M, K, device=DEV)
        Y = torch.empty(M, N, device=DEV)
        t = timeit(lambda: ops.linear_fwd(X, Wsrc, bias, 1, Y, "f32"))
        print("fresh tensors %d: X@%#x Y@%#x  %.0f us" % (r, X.data_ptr(), Y.data_ptr(), t), flush=True)
        pools.append((X, Y))       # keep them: the next pair lands elsewhere


if __name__ == "__main__":
    main()
