"""The RCCL ("nccl" backend on ROCm) branches of dlrm_amd.ext_dist, executed on ONE MI355X: a one-rank process group forced
through the distributed code path (ext_dist.init_distributed(..., force=True)), so every collective really runs on RCCL as
a self-exchange — asynchronous all_to_all_single work handles waited on the autograd thread (_A2AStart / _A2AWait,
extend_distributed.py:389-486), the pipelined exchange, torch DDP and FlatDDP with ReduceOp.AVG (dlrm_s_pytorch.py:1329-1336),
reduce_scatter_tensor / all_gather_into_tensor (reduce_scatter_rows) and the device all-to-all + all-gather of ids
(kjt_input_dist) — against the single-process result of the same model on the same inputs.  At world size 1 every exchange
is the identity, so the two must agree to fp32 round-off of the (different) kernel schedules; the gloo multi-rank tests
(test_gpu_dist.py, test_dist_gloo.py) cover the arithmetic of real exchanges, this file covers the backend."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.multiprocessing as mp

pytestmark = pytest.mark.gpu

_CFG = dict(D=128, rows=[50, 7, 3000, 11, 400], bot=[13, 64, 128], top=[64, 32, 1], B=64, lr=0.1, steps=2)


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _inputs(step):
    c = _CFG
    rng = np.random.default_rng(40 + step)
    X = rng.random((c["B"], 13)).astype(np.float32)
    idx = np.stack([rng.integers(0, n, size=c["B"]) for n in c["rows"]]).astype(np.int64)
    off = np.tile(np.arange(c["B"], dtype=np.int64), (len(c["rows"]), 1))
    T = rng.integers(0, 2, size=(c["B"], 1)).astype(np.float32)
    return X, off, idx, T


def _train(dev, wrap=None, chunks=1):
    """two SGD steps of a small DLRM_Net; returns predictions, losses and final parameters"""
    import dlrm_amd
    from dlrm_amd import ext_dist, ops
    c = _CFG
    F = len(c["rows"]) + 1
    ln_top = np.asarray([c["D"] + F * (F - 1) // 2] + c["top"])
    np.random.seed(11)
    model = dlrm_amd.DLRM_Net(c["D"], np.asarray(c["rows"]), np.asarray(c["bot"]), ln_top, "dot", sigmoid_top=ln_top.size - 2,
                              loss_function="bce").to(dev)
    model.emb_update_mode = ops.UPD_DETERMINISTIC
    model.a2a_chunks = chunks
    if wrap is not None:
        model.bot_l = wrap(model.bot_l, device_ids=[0])
        model.top_l = wrap(model.top_l, device_ids=[0])
    opt = torch.optim.SGD([{"params": [p for e in model.emb_l for p in e.parameters()], "lr": c["lr"]},
                           {"params": model.bot_l.parameters(), "lr": c["lr"]},
                           {"params": model.top_l.parameters(), "lr": c["lr"]}], lr=c["lr"])
    res = {}
    for s in range(c["steps"]):
        X, off, idx, T = _inputs(s)
        Z = model(torch.from_numpy(X).to(dev), torch.from_numpy(off).to(dev), torch.from_numpy(idx).to(dev))
        E = model.loss_fn(Z, torch.from_numpy(T).to(dev))
        res[f"s{s}.Z"] = Z.detach().cpu().numpy()
        res[f"s{s}.loss"] = float(E.detach())
        opt.zero_grad()
        E.backward()
        opt.step()
    torch.cuda.synchronize()
    ops.check_index_errors(sync=True)
    for j, e in enumerate(model.emb_l):
        res[f"emb.{j}"] = e.weight.detach().cpu().numpy()
    for tower, name in ((model.bot_l, "bot"), (model.top_l, "top")):
        inner = tower.module if hasattr(tower, "module") else tower
        for k, p in inner.named_parameters():
            res[f"{name}.{k}"] = p.detach().cpu().numpy()
    if wrap is ext_dist.FlatDDP:
        res["flat_avg"] = bool(model.top_l._avg)
    return res


def _sharded(dev, dist_on):
    """ShardedDLRM on ONE rank with a hand-made plan that shards table 2 ROW-WISE over the (one) rank: the forced group sends its
    ids through kjt_input_dist (all_to_all_single + all_gather_into_tensor on RCCL) and its partial sums through
    reduce_scatter_tensor; without the group the same model computes locally"""
    from dlrm_amd import ops, sharding
    from dlrm_amd.optim import FusedRWSAdagrad
    from dlrm_amd.torchrec_variant import ShardedDLRM
    rows, hot, D, B = [50, 7, 3000, 11], [3, 1, 7, 2], 16, 32
    cost = [sharding.table_cost(B, h, D) for h in hot]
    plan = sharding.ShardingPlan(1, [sharding.TableShard(0, "table", 0), sharding.TableShard(1, "table", 0),
                                     sharding.TableShard(2, "row", -1, ((0, rows[2]),)), sharding.TableShard(3, "table", 0)],
                                 cost, [sum(cost)], [0])
    np.random.seed(5)
    model = ShardedDLRM(rows, hot, D, 13, [32, 16], [48, 24, 1], B, plan=plan).to(dev)
    assert model.rw_tables == [2]
    opt = FusedRWSAdagrad(model.parameters(), lr=0.05)
    rng = np.random.default_rng(77)
    out = {}
    for s in range(2):
        X = torch.from_numpy(rng.random((B, 13)).astype(np.float32)).to(dev)
        values = torch.from_numpy(np.concatenate([rng.integers(0, n, size=B * h) for n, h in zip(rows, hot)]).astype(np.int32)).to(dev)
        y = torch.from_numpy(rng.integers(0, 2, size=(B, 1)).astype(np.float32)).to(dev)
        z = model(X, values)
        loss = model.loss_fn(z, y)
        out[f"s{s}.z"] = z.detach().cpu().numpy()
        opt.zero_grad()
        loss.backward()
        opt.step()
    torch.cuda.synchronize()
    ops.check_index_errors(sync=True)
    for j, e in enumerate(model.emb_l):
        out[f"emb.{j}"] = e.weight.detach().cpu().numpy()
    return out


def _worker(port, q):
    try:
        os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK="0", WORLD_SIZE="1", LOCAL_RANK="0")
        from dlrm_amd import ext_dist
        dev = torch.device("cuda:0")
        torch.cuda.set_device(dev)
        out = {"single": _train(dev), "sharded_single": _sharded(dev, False)}
        assert not ext_dist.is_distributed()
        ext_dist.init_distributed(rank=0, local_rank=0, size=1, use_gpu=True, backend="nccl", force=True)
        out["backend"] = torch.distributed.get_backend()
        out["is_distributed"] = ext_dist.is_distributed() and ext_dist.my_size == 1 and ext_dist.alltoall_supported
        out["ddp"] = _train(dev, wrap=ext_dist.TorchDDP)
        out["ddp_chunks2"] = _train(dev, wrap=ext_dist.TorchDDP, chunks=2)
        out["flat"] = _train(dev, wrap=ext_dist.FlatDDP)
        out["sharded_rccl"] = _sharded(dev, True)
        # the collectives on their own, through autograd
        x = torch.randn(8, 12, device=dev, requires_grad=True)
        y = ext_dist.reduce_scatter_rows(x)                     # reduce_scatter_tensor over one rank: y == x
        (y * 3.0).sum().backward()                              # backward: all_gather_into_tensor of the gradient
        out["rs_ok"] = bool(torch.equal(y.detach(), x.detach()) and torch.equal(x.grad, torch.full_like(x, 3.0)))
        g = ext_dist.all_gather(torch.arange(6, device=dev, dtype=torch.float32).view(3, 2), None)
        out["ag_ok"] = bool(torch.equal(g, torch.arange(6, device=dev, dtype=torch.float32).view(3, 2)))
        hot = [2, 1, 3]
        vals = torch.arange(4 * sum(hot), device=dev, dtype=torch.int32)           # Bl = 4, key-major
        tw, rw = ext_dist.kjt_input_dist(vals, hot, [0, -1, 0], [1])
        out["kjt_ok"] = bool(torch.equal(tw[0], vals[:8]) and torch.equal(tw[2], vals[12:24]) and torch.equal(rw[1], vals[8:12]))
        torch.cuda.synchronize()
        ext_dist.barrier()
        torch.distributed.destroy_process_group()
        q.put(("ok", out))
    except BaseException as e:                                  # noqa: BLE001 - reported to the parent, which fails the test
        import traceback
        q.put(("error", "%s\n%s" % (e, traceback.format_exc())))


@pytest.fixture(scope="module")
def rccl_run():
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    p = ctx.Process(target=_worker, args=(_free_port(), q))
    p.start()
    status, out = q.get(timeout=900)
    p.join(120)
    assert status == "ok", out
    assert p.exitcode == 0
    return out


def test_the_process_group_is_rccl_at_world_size_one(rccl_run):
    assert rccl_run["backend"] == "nccl" and rccl_run["is_distributed"]
    assert rccl_run["flat"]["flat_avg"] is True                  # FlatDDP took ReduceOp.AVG (the RCCL-only branch)


@pytest.mark.parametrize("leg", ["ddp", "ddp_chunks2", "flat"])
def test_distributed_forward_over_rccl_equals_the_single_process_step(rccl_run, leg):
    """distributed_forward + alltoall().wait() (async RCCL work handles; chunks2 = the pipelined exchange) + DDP / FlatDDP(AVG)
    vs sequential_forward: same predictions, losses, embedding tables and towers after two steps"""
    a, b = rccl_run["single"], rccl_run[leg]
    for k, v in a.items():
        if k.endswith(".loss"):
            assert abs(b[k] - v) <= 1e-6 * abs(v), (k, b[k], v)
        else:
            np.testing.assert_allclose(b[k], v, rtol=1e-5, atol=1e-6, err_msg=f"{leg}: {k}")


def test_row_wise_and_input_exchange_collectives_over_rccl(rccl_run):
    assert rccl_run["rs_ok"] and rccl_run["ag_ok"] and rccl_run["kjt_ok"]
    a, b = rccl_run["sharded_single"], rccl_run["sharded_rccl"]
    for k, v in a.items():
        np.testing.assert_allclose(b[k], v, rtol=1e-5, atol=1e-6, err_msg=k)
