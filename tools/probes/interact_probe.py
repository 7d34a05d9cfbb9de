#!/usr/bin/env python3
"""Probe: what bounds the D = 128 interaction forward kernels at Criteo-Terabyte shapes — the multiplication or the row fetch?
Run once per DLRM_INTERACT_DEBUG value (0 = real kernel, 1 = no DMA / waits, 2 = DMA only); prints us per call of dlrm_emb_fwd,
dlrm_interact_fwd, dlrm_interact_fwd_gather and the backward pair."""
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
x = torch.randn(B, D, device=dev)
feat[:, :D] = x
R = torch.empty(B, 480, device=dev)
dR = torch.randn(B, 480, device=dev)
dfeat = torch.empty(B, (T + 1) * D, device=dev)
dx, dE = torch.empty(B, D, device=dev), torch.empty(B, T * D, device=dev)
bags = ops.BagBatch(torch.arange(B, device=dev).repeat(T, 1), torch.stack([torch.randint(0, n, (B,), device=dev) for n in ROWS]))


def timed(name, fn, n=20):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(n):
        fn()
    b.record()
    torch.cuda.synchronize()
    print("  %-28s %7.1f us" % (name, a.elapsed_time(b) / n * 1e3), flush=True)


print("DLRM_INTERACT_DEBUG=%s" % os.environ.get("DLRM_INTERACT_DEBUG", "0"))
timed("emb_fwd", lambda: ops.emb_fwd(Ws, bags, feat[:, D:]))
timed("interact_fwd", lambda: ops.interact_fwd([feat[:, :D], feat[:, D:]], D, False, R))
timed("interact_fwd_gather", lambda: ops.interact_fwd_gather(x, Ws, bags, D, False, R))
timed("interact_bwd", lambda: ops.interact_bwd([feat[:, :D], feat[:, D:]], D, False, dR, [dfeat[:, :D], dfeat[:, D:]]))
timed("interact_bwd_gather", lambda: ops.interact_bwd_gather(x, Ws, bags, D, False, dR, dx, dE))
ops.check_index_errors(sync=True)
