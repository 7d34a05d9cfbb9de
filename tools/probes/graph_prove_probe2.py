#!/usr/bin/env python3
"""Locate the first divergence of the graphed step from the eager step when a host synchronisation precedes some calls (D = 16 fixture)."""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
from conftest import golden_batches, load_golden, params_with_prefix  # noqa: E402
from test_gpu_model import build_model  # noqa: E402

import dlrm_amd  # noqa: E402
from dlrm_amd import ops  # noqa: E402
from dlrm_amd.graph import GraphedTrainStep  # noqa: E402
from dlrm_amd.optim import FusedSGD  # noqa: E402

os.environ["DLRM_GRAPH_SORTED"] = "1"
d, meta = load_golden("config1_b128")
device = torch.device("cuda:0")
batches = [(torch.from_numpy(X).to(device), [torch.from_numpy(o).to(device) for o in lS_o],
            [torch.from_numpy(i).to(device) for i in lS_i], torch.from_numpy(T).to(device)) for X, lS_o, lS_i, T in golden_batches(d, meta)]
B = batches[0][0].size(0)
fixed = []
for X, lS_o, lS_i, T in batches:
    idx = [i[o.clamp(max=max(i.numel() - 1, 0))] if i.numel() else i for o, i in zip(lS_o, lS_i)]
    fixed.append((X, [torch.arange(B, device=device)] * len(idx), idx, T))
seq = [fixed[i % len(fixed)] for i in range(7)]


def run(sync_calls):
    ops.offsets_are_iota = lambda o: True
    me = build_model(meta, params_with_prefix(d, "init"), device)
    mg = build_model(meta, params_with_prefix(d, "init"), device)
    oe, og = FusedSGD(me.parameters(), lr=meta["lr"]), FusedSGD(mg.parameters(), lr=meta["lr"])
    step = None
    for n, (X, off, idx, T) in enumerate(seq):
        E = me.loss_fn(me(X, off, idx), T); oe.zero_grad(); E.backward(); oe.step(); le = float(E.detach()); del E
        if n == 0:
            E = mg.loss_fn(mg(X, off, idx), T); og.zero_grad(); E.backward(); og.step(); lg = float(E.detach()); del E
            step = GraphedTrainStep(mg, og, warmup=2)
        else:
            if n in sync_calls:
                torch.cuda.current_stream().synchronize()
            lg = float(step(X, off, idx, T))
        torch.cuda.synchronize()
        worst = max((float((a - b).abs().max()), k) for (k, a), (_, b) in zip(me.state_dict().items(), mg.state_dict().items()))
        print("  call %d  loss eager %.7f graph %.7f  max param diff %.3e (%s) captures %d" % (n, le, lg, worst[0], worst[1], step.captures), flush=True)


for sc in ([], [1], [2], [3], [1, 2, 3], [4, 5, 6]):
    print("sync before calls", sc, flush=True)
    run(set(sc))
