#!/usr/bin/env python3
"""Probe: do the pooled embeddings stay in the Infinity Cache (256 MB, memory side) between dlrm_emb_fwd and dlrm_interact_fwd when the
batch is processed in chunks?  Whole batch: emb_fwd writes 872 MB, interact_fwd reads it back from HBM.  Chunks of B/C samples: the
chunk's 872/C MB may still be cached when the interaction reads them.  Prints total time of (emb_fwd + interact_fwd) for C = 1, 2, 4, 8, 16."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from dlrm_amd import ops  # noqa: E402

ROWS = [39884406, 39043, 17289, 7420, 20263, 3, 7120, 1543, 63, 38532951, 2953546, 403346, 10, 2208, 11938, 155,
        4, 976, 14, 39979771, 25641295, 39664984, 585935, 12972, 108, 36]
dev = torch.device("cuda:0")
B, D, T = 65536, 128, len(ROWS)
Ws = [torch.empty(n, D, device=dev).uniform_(-0.01, 0.01) for n in ROWS]
feat = torch.empty(B, (T + 1) * D, device=dev)
feat[:, :D] = torch.randn(B, D, device=dev)
R = torch.empty(B, 480, device=dev)
for C in (1, 2, 4, 8, 16, 1):
    Bc = B // C
    bags = [ops.BagBatch(torch.arange(Bc, device=dev).repeat(T, 1), torch.stack([torch.randint(0, n, (Bc,), device=dev) for n in ROWS]))
            for _ in range(C)]

    def run():
        for c in range(C):
            f = feat[c * Bc:(c + 1) * Bc]
            ops.emb_fwd(Ws, bags[c], f[:, D:])
            ops.interact_fwd([f[:, :D], f[:, D:]], D, False, R[c * Bc:(c + 1) * Bc])
    for _ in range(3):
        run()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(10):
        run()
    b.record()
    torch.cuda.synchronize()
    print("chunks %2d (%5d samples, %4.0f MB pooled per chunk): emb_fwd + interact_fwd = %.1f us per batch" % (C, Bc, Bc * T * D * 4 / 1e6, a.elapsed_time(b) / 10 * 1e3), flush=True)
