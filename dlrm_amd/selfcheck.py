"""`python -m dlrm_amd.selfcheck` — run the RCCL branches of dlrm_amd.ext_dist on ONE GPU and print one JSON line.

A one-rank "nccl" (= RCCL on ROCm) process group is forced through the distributed code path
(`ext_dist.init_distributed(force=True)`): `DLRM_Net.distributed_forward`, the asynchronous `alltoall()` / `.wait()` pair
(extend_distributed.py:389-486,541-576), `FlatDDP`'s `ReduceOp.AVG` all-reduce and torch DDP (dlrm_s_pytorch.py:1329-1336),
`reduce_scatter_rows` and `kjt_input_dist` all execute their RCCL calls as self-exchanges, and two training steps are compared with
the single-process (`sequential_forward`) steps of the same model on the same inputs.  `bench.py --gpus 1` runs this in a child
process (its `rccl_selfcheck` field); `tests/test_gpu_rccl.py` is the thorough version."""
from __future__ import annotations

import json
import os
import sys
import time

import numpy as np
import torch


def _steps(dev, wrap=None, chunks=1):
    import dlrm_amd
    from dlrm_amd import ops
    D, rows, B = 128, [50, 7, 3000, 11], 256
    F = len(rows) + 1
    ln_top = np.asarray([D + F * (F - 1) // 2, 64, 1])
    np.random.seed(7)
    model = dlrm_amd.DLRM_Net(D, np.asarray(rows), np.asarray([13, 64, D]), ln_top, "dot", sigmoid_top=ln_top.size - 2,
                              loss_function="bce").to(dev)
    model.emb_update_mode = ops.UPD_DETERMINISTIC
    model.a2a_chunks = chunks
    if wrap is not None:
        model.bot_l, model.top_l = wrap(model.bot_l, device_ids=[dev.index]), wrap(model.top_l, device_ids=[dev.index])
    opt = torch.optim.SGD([{"params": [p for e in model.emb_l for p in e.parameters()]}, {"params": model.bot_l.parameters()},
                           {"params": model.top_l.parameters()}], lr=0.1)
    g = torch.Generator().manual_seed(3)
    out = []
    for _ in range(2):
        X = torch.rand((B, 13), generator=g).to(dev)
        idx = torch.stack([torch.randint(0, n, (B,), generator=g) for n in rows]).to(dev)
        off = torch.arange(B).repeat(len(rows), 1).to(dev)
        T = torch.randint(0, 2, (B, 1), generator=g).float().to(dev)
        Z = model(X, off, idx)
        E = model.loss_fn(Z, T)
        opt.zero_grad()
        E.backward()
        opt.step()
        out.append(Z.detach().clone())
    out += [e.weight.detach().clone() for e in model.emb_l]
    inner = model.top_l.module if hasattr(model.top_l, "module") else model.top_l
    out += [p.detach().clone() for p in inner.parameters()]
    torch.cuda.synchronize()
    ops.check_index_errors(sync=True)
    return out


def run() -> dict:
    from dlrm_amd import ext_dist
    t0 = time.time()
    dev = torch.device("cuda", 0)
    torch.cuda.set_device(dev)
    ref = _steps(dev)
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    os.environ.setdefault("MASTER_PORT", str(29500 + os.getpid() % 2000))
    ext_dist.init_distributed(rank=0, local_rank=0, size=1, use_gpu=True, backend="nccl", force=True)
    res = {"backend": torch.distributed.get_backend(), "world_size": torch.distributed.get_world_size(), "legs": {}}
    worst = 0.0
    for name, wrap, chunks in (("alltoall+FlatDDP(AVG)", ext_dist.FlatDDP, 1), ("pipelined alltoall x2 + torch DDP", ext_dist.TorchDDP, 2)):
        got = _steps(dev, wrap, chunks)
        err = max(float(((a - b).abs() / (b.abs() + 1e-3)).max()) for a, b in zip(got, ref))
        res["legs"][name] = err
        worst = max(worst, err)
    x = torch.randn(8, 12, device=dev, requires_grad=True)
    y = ext_dist.reduce_scatter_rows(x)
    y.sum().backward()
    vals = torch.arange(18, device=dev, dtype=torch.int32)
    tw, rw = ext_dist.kjt_input_dist(vals, [2, 1, 3], [0, -1, 0], [1])
    res["collectives_identity"] = bool(torch.equal(y.detach(), x.detach()) and torch.equal(x.grad, torch.ones_like(x))
                                       and torch.equal(tw[0], vals[:6]) and torch.equal(tw[2], vals[9:]) and torch.equal(rw[1], vals[6:9]))
    torch.cuda.synchronize()
    torch.distributed.destroy_process_group()
    res["max_rel_err_vs_single_process"] = worst
    res["ok"] = bool(res["backend"] == "nccl" and worst <= 1e-5 and res["collectives_identity"])
    res["seconds"] = round(time.time() - t0, 2)
    res["what"] = ("one-rank RCCL group forced through distributed_forward / alltoall().wait() / FlatDDP ReduceOp.AVG / torch DDP / "
                   "reduce_scatter_tensor / all_gather_into_tensor / device all_to_all_single of ids; two SGD steps vs the single-process steps")
    return res


if __name__ == "__main__":
    try:
        print(json.dumps(run()))
    except BaseException as e:                                  # noqa: BLE001 - the caller reads the JSON line
        print(json.dumps({"ok": False, "error": "%s: %s" % (type(e).__name__, str(e)[:400])}))
        sys.exit(1)
