#!/usr/bin/env python3
"""sustained_gemm.py — does the fp32 GEMM slow down under sustained load (power / clock management)?

The isolated per-layer numbers of tools/microbench.py come from 10 launches after an idle gap; inside the training step the same
kernels run back to back for seconds and are ~8 % slower.  This probe runs ONE layer (M = 65536, 1024 -> 1024, forward) in
chunks of 25 launches for ~3 s after a 3 s idle gap and prints the per-chunk average next to `rocm-smi` clock / power samples
taken by a background thread, then the same for the HBM-bound embedding lookup.  Tuning aid; not part of the product."""
import os
import subprocess
import sys
import threading
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from dlrm_amd import ops  # noqa: E402

DEV = torch.device("cuda:0")
samples = []
stop = False


def sampler():
    while not stop:
        t = time.time()
        try:
            out = subprocess.run(["rocm-smi", "-c", "-P", "--csv"], capture_output=True, text=True, timeout=5).stdout
            samples.append((t, " | ".join(l for l in out.strip().splitlines() if l and not l.startswith("WARNING"))))
        except Exception as e:  # noqa: BLE001
            samples.append((t, "rocm-smi failed: %r" % (e,)))
        time.sleep(0.25)


def chunks(fn, n_chunks, per_chunk, flop):
    ev = [torch.cuda.Event(enable_timing=True) for _ in range(n_chunks + 1)]
    t0 = time.time()
    ev[0].record()
    for c in range(n_chunks):
        for _ in range(per_chunk):
            fn()
        ev[c + 1].record()
    torch.cuda.synchronize()
    t1 = time.time()
    ms = [ev[c].elapsed_time(ev[c + 1]) / per_chunk for c in range(n_chunks)]
    return t0, t1, ms, [flop / m / 1e9 for m in ms]


def main():
    M, N, K = 65536, 1024, 1024
    X = torch.randn(M, K, device=DEV)
    W = torch.randn(N, K, device=DEV) * 0.03
    b = torch.randn(N, device=DEV)
    Y = torch.empty(M, N, device=DEV)
    fn = lambda: ops.linear_fwd(X, W, b, 1, Y, "f32")  # noqa: E731
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    th = threading.Thread(target=sampler, daemon=True)
    th.start()
    time.sleep(3.0)
    t0, t1, ms, tf = chunks(fn, 120, 25, 2.0 * M * N * K)
    print("gemm 1024x1024 fwd, 120 chunks of 25 launches after 3 s idle: first %.1f us (%.1f TF)  chunk 5: %.1f  20: %.1f  60: %.1f  "
          "last %.1f us (%.1f TF)" % (ms[0] * 1e3, tf[0], ms[5] * 1e3, ms[20] * 1e3, ms[60] * 1e3, ms[-1] * 1e3, tf[-1]))
    print("per-chunk us:", " ".join("%.0f" % (m * 1e3) for m in ms))
    time.sleep(1.0)
    global stop
    stop = True
    th.join()
    print("rocm-smi samples (t relative to the start of the load; load ran %.2f s):" % (t1 - t0))
    for t, s in samples:
        print("  %+6.2f  %s" % (t - t0, s[:400]))


if __name__ == "__main__":
    main()
