#!/usr/bin/env python3
"""chain_vs_isolated.py — why are the GEMMs ~8 % slower inside the training step than launched alone?

Runs the top tower's forward layers (M = 65536: 480 -> 1024 -> 1024 -> 512 -> 256) three ways and prints per-layer times:
  (a) isolated: one layer launched 20 times back to back (what tools/microbench.py measures);
  (b) chain:    the four layers one after the other, 20 rounds, events around every launch (the order inside the step);
  (c) chain with 2.7 GB of unrelated HBM traffic between rounds (stands for the embedding / interaction kernels);
  (d) chain, every round on FRESH buffers from torch's caching allocator (the step allocates activations every iteration).
Tuning aid; not part of the product."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from dlrm_amd import ops  # noqa: E402

DEV = torch.device("cuda:0")
M = 65536
DIMS = [480, 1024, 1024, 512, 256]
ROUNDS = 20


def alloc():
    acts = [torch.randn(M, DIMS[0], device=DEV)] + [torch.empty(M, n, device=DEV) for n in DIMS[1:]]
    bits = [ops.relu_bits_alloc(M, n, DEV) for n in DIMS[1:]]
    return acts, bits


def main():
    Ws = [torch.randn(DIMS[i + 1], DIMS[i], device=DEV) * (2.0 / (DIMS[i] + DIMS[i + 1])) ** 0.5 for i in range(4)]
    bs = [torch.randn(DIMS[i + 1], device=DEV) * 0.05 for i in range(4)]
    acts, bits = alloc()
    junk_a, junk_b = torch.empty(1350 << 18, device=DEV), torch.empty(1350 << 18, device=DEV)   # 1.35 GB each

    def layer(i, a=acts, b=bits):
        ops.linear_fwd(a[i], Ws[i], bs[i], 1, a[i + 1], "f32", b[i])

    for i in range(4):
        layer(i)
    torch.cuda.synchronize()

    def ev():
        return torch.cuda.Event(enable_timing=True)

    # (a)
    iso = []
    for i in range(4):
        e0, e1 = ev(), ev()
        for _ in range(3):
            layer(i)
        e0.record()
        for _ in range(ROUNDS):
            layer(i)
        e1.record()
        torch.cuda.synchronize()
        iso.append(e0.elapsed_time(e1) / ROUNDS * 1e3)

    def chain(between=None, fresh=False, sets=None):
        tot = [0.0] * 4
        evs = []
        for r in range(ROUNDS + 2):
            a, b = alloc() if fresh else (sets[r % len(sets)] if sets else (acts, bits))
            if between is not None:
                between()
            row = []
            for i in range(4):
                e0, e1 = ev(), ev()
                e0.record()
                layer(i, a, b)
                e1.record()
                row.append((e0, e1))
            if r >= 2:
                evs.append(row)
        torch.cuda.synchronize()
        for row in evs:
            for i, (e0, e1) in enumerate(row):
                tot[i] += e0.elapsed_time(e1)
        return [t / ROUNDS * 1e3 for t in tot]

    ch = chain()
    chj = chain(between=lambda: junk_b.copy_(junk_a))
    chf = chain(fresh=True)
    setB, setC, setD = alloc(), alloc(), alloc()
    for nm, st in (("set B only", [setB]), ("A/B alternating", [(acts, bits), setB]), ("A/B/C/D", [(acts, bits), setB, setC, setD]),
                   ("set A again", [(acts, bits)])):
        t = chain(sets=st)
        print("%-16s " % nm + "  ".join("%7.1f" % x for x in t) + "   sum %.1f us" % sum(t))
    print("addresses A:", [hex(x.data_ptr()) for x in acts], "B:", [hex(x.data_ptr()) for x in setB[0]])
    print("%-14s %10s %10s %12s %12s" % ("layer", "isolated", "chain", "chain+junk", "chain fresh"))
    for i in range(4):
        print("%4d -> %-6d %8.1f us %8.1f us %10.1f us %10.1f us" % (DIMS[i], DIMS[i + 1], iso[i], ch[i], chj[i], chf[i]))
    print("%-14s %8.1f us %8.1f us %10.1f us %10.1f us" % ("sum", sum(iso), sum(ch), sum(chj), sum(chf)))


if __name__ == "__main__":
    main()
