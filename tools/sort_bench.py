#!/usr/bin/env python3
"""Time the (table, row) sort of the sort-based embedding updates alone (dlrm_emb_sort_lookups) at Criteo-Terabyte shapes:
26 tables, B = 65536 one-hot int64 lookups.  DLRM_SORT=rocprim selects the general sorter (A/B).  Prints one line."""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from dlrm_amd import ops  # noqa: E402

ROWS = [39884406, 39043, 17289, 7420, 20263, 3, 7120, 1543, 63, 38532951, 2953546, 403346, 10, 2208, 11938, 155,
        4, 976, 14, 39979771, 25641295, 39664984, 585935, 12972, 108, 36]
B = int(sys.argv[1]) if len(sys.argv) > 1 and sys.argv[1].isdigit() else 65536
dev = torch.device("cuda:0")
g = torch.Generator(device=dev).manual_seed(1)
if "v2" in sys.argv:
    # MLPerf-v2 multi-hot batch (torchrec_dlrm/README.MD:45,159): 214 lookups per sample, int32 ids, the 100-hot table = 6.55 M lookups
    ROWS = [40000000, 39060, 17295, 7424, 20265, 3, 7122, 1543, 63, 40000000, 3067956, 405282, 10, 2209, 11938, 155, 4,
            976, 14, 40000000, 40000000, 40000000, 590152, 12973, 108, 36]
    HOT = [3, 2, 1, 2, 6, 1, 1, 1, 1, 7, 3, 8, 1, 6, 9, 5, 1, 1, 1, 12, 100, 27, 10, 3, 1, 1]
    idx = [torch.randint(0, n, (B * h,), device=dev, generator=g).to(torch.int32) for n, h in zip(ROWS, HOT)]
    off = [(torch.arange(B, device=dev) * h).to(torch.int32) for h in HOT]
else:
    idx = torch.stack([torch.randint(0, n, (B,), device=dev, generator=g) for n in ROWS])
    off = torch.arange(B, device=dev).repeat(len(ROWS), 1)
bags = ops.BagBatch(off, idx)
for _ in range(5):
    ops.sort_lookups(ROWS, bags)
torch.cuda.synchronize()
a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
a.record()
N = 10 if "v2" in sys.argv else 50
for _ in range(N):
    ops.sort_lookups(ROWS, bags)
b.record()
torch.cuda.synchronize()
print("sort_lookups B=%d sorter=%s: %.1f us per call (incl. key expansion, 3 small output copies and host launch path)"
      % (B, os.environ.get("DLRM_SORT", "own"), a.elapsed_time(b) / N * 1e3))
