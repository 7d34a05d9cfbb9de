"""world_size-2 gloo run (CPU) of dlrm_amd.ext_dist: partition arithmetic, the pooled-embedding
all-to-all layouts in both directions, all_gather, against the oracle's layout restatement."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.multiprocessing as mp

from oracle import oracle as O


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, size, port, T, B, D, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(size), LOCAL_RANK=str(rank))
    from dlrm_amd import ext_dist
    ext_dist.init_distributed(rank=rank, local_rank=rank, size=size, use_gpu=False, backend="gloo")
    assert ext_dist.my_size == size and ext_dist.my_rank == rank
    n_local, per_rank = ext_dist.get_split_lengths(T)
    tables = list(range(T))[ext_dist.get_my_slice(T)]
    # pooled[b, j*D + d] = 1000*table + b + d/100  (deterministic, rank independent)
    def pooled_of(t):
        b = torch.arange(B, dtype=torch.float32).view(B, 1)
        return 1000.0 * t + b + torch.arange(D, dtype=torch.float32).view(1, D) / 100.0
    packed = torch.cat([pooled_of(t) for t in tables], dim=1).requires_grad_(True)
    # zero-copy form: one packed block
    req = ext_dist.alltoall([packed], per_rank, emb_dim=D)
    outs = req.wait()
    loss = sum(((s + 1) * o).sum() for s, o in enumerate(outs))
    loss.backward()
    res = {"outs": [o.detach().numpy().copy() for o in outs], "grad": packed.grad.numpy().copy(), "tables": tables}
    # reference form: one tensor per local table
    ins = [pooled_of(t).requires_grad_(True) for t in tables]
    outs2 = ext_dist.alltoall(ins, per_rank).wait()
    sum(o.sum() for o in outs2).backward()
    res["outs2"] = [o.detach().numpy().copy() for o in outs2]
    res["grad2"] = [i.grad.numpy().copy() for i in ins]
    g = ext_dist.all_gather(torch.full((ext_dist.get_split_lengths(B)[0], 1), float(rank)), None)
    res["gather"] = g.numpy().copy()
    q.put((rank, res))
    ext_dist.barrier()
    torch.distributed.destroy_process_group()


@pytest.mark.parametrize("size,T,B,D", [(2, 3, 8, 4), (2, 4, 6, 2),
                                        # the real Criteo split: 26 tables over 8 ranks -> [4,4,3,3,3,3,3,3], B/N = 8
                                        (8, 26, 64, 4)])
def test_alltoall_layouts(size, T, B, D):
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, size, port, T, B, D, q)) for r in range(size)]
    for p in procs:
        p.start()
    results = dict(q.get(timeout=180) for _ in range(size))
    for p in procs:
        p.join(60)
        assert p.exitcode == 0
    # oracle layout: destination r receives from source s the rows of its batch slice of s's tables
    def pooled_of(t):
        b = np.arange(B, dtype=np.float32).reshape(B, 1)
        return 1000.0 * t + b + np.arange(D, dtype=np.float32).reshape(1, D) / 100.0
    pooled_by_rank = []
    for s in range(size):
        sl = O.my_slice(T, s, size)
        pooled_by_rank.append(np.concatenate([pooled_of(t) for t in range(T)[sl]], axis=1))
        assert results[s]["tables"] == list(range(T))[sl]
    if (size, T) == (8, 26):
        assert [len(results[s]["tables"]) for s in range(size)] == [4, 4, 3, 3, 3, 3, 3, 3]
    want = O.a2a_forward_layout(pooled_by_rank, size)
    for r in range(size):
        for s in range(size):
            assert np.array_equal(results[r]["outs"][s], want[r][s]), (r, s)
            assert np.array_equal(results[r]["outs2"][s], want[r][s]), (r, s)
        # backward: d loss / d pooled[b, :] = (source index of the destination that owns row b) + 1 == my rank + 1
        # on every row (each destination weights MY block by (my_rank + 1))
        assert np.array_equal(results[r]["grad"], np.full_like(results[r]["grad"], r + 1.0))
        for gi in results[r]["grad2"]:
            assert np.array_equal(gi, np.ones_like(gi))
        lb = B // size
        assert np.array_equal(results[r]["gather"].reshape(-1), np.repeat(np.arange(size, dtype=np.float32), lb))


def _chunk_worker(rank, size, port, T, B, D, C, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(size), LOCAL_RANK=str(rank))
    from dlrm_amd import ext_dist
    from dlrm_amd.functional import ChunkPackFunction
    ext_dist.init_distributed(rank=rank, local_rank=rank, size=size, use_gpu=False, backend="gloo")
    _, per_rank = ext_dist.get_split_lengths(T)
    tables = list(range(T))[ext_dist.get_my_slice(T)]
    g = torch.Generator().manual_seed(100 + rank)
    base = torch.randn(B, len(tables) * D, generator=g)
    # a different weight per (destination row, source rank, column) so that a mis-routed gradient row cannot cancel out
    def weights(n_rows, width, src):
        return (torch.arange(n_rows * width, dtype=torch.float32).view(n_rows, width) % 7 + 1.0) * (src + 1)
    Bl = B // size
    # single exchange
    E1 = base.clone().requires_grad_(True)
    outs1 = ext_dist.alltoall([E1], per_rank, emb_dim=D).wait()
    sum((o * weights(o.size(0), o.size(1), s)).sum() for s, o in enumerate(outs1)).backward()
    # pipelined: C chunk exchanges issued up front, consumed in order (DLRM_Net._pipelined_exchange_forward)
    E2 = base.clone().requires_grad_(True)
    sends = ChunkPackFunction.apply(E2, size, C)
    reqs = [ext_dist.alltoall([sends[c]], per_rank, emb_dim=D) for c in range(C)]
    Bc = Bl // C
    loss = 0.0
    chunks = []
    for c in range(C):
        ly = reqs[c].wait()
        chunks.append([o.detach().clone() for o in ly])
        for s, o in enumerate(ly):
            loss = loss + (o * weights(Bl, o.size(1), s)[c * Bc:(c + 1) * Bc]).sum()
    loss.backward()
    q.put((rank, {"outs1": [o.detach().numpy().copy() for o in outs1],
                  "outs2": [np.concatenate([chunks[c][s].numpy() for c in range(C)], axis=0) for s in range(size)],
                  "g1": E1.grad.numpy().copy(), "g2": E2.grad.numpy().copy()}))
    ext_dist.barrier()
    torch.distributed.destroy_process_group()


@pytest.mark.parametrize("size,T,B,D,C", [(2, 3, 8, 4, 2), (3, 7, 18, 2, 3), (2, 5, 16, 3, 4)])
def test_pipelined_exchange_equals_single_exchange(size, T, B, D, C):
    """The chunked all-to-all of the distributed forward (uneven table splits, 2 and 3 ranks): concatenating the chunk
    results reproduces the single exchange exactly, and the reverse exchanges deliver every gradient row to the same
    place (bit-identical gradients of the packed embeddings)."""
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_chunk_worker, args=(r, size, port, T, B, D, C, q)) for r in range(size)]
    for p in procs:
        p.start()
    results = dict(q.get(timeout=180) for _ in range(size))
    for p in procs:
        p.join(60)
        assert p.exitcode == 0
    for r in range(size):
        for s in range(size):
            assert np.array_equal(results[r]["outs1"][s], results[r]["outs2"][s]), (r, s)
        assert np.array_equal(results[r]["g1"], results[r]["g2"]), r
        assert np.abs(results[r]["g1"]).min() > 0


# ---------------------------------------------------------------------------------------------------------------------
# planned sharding, key-major id input distribution, row-wise shard collectives
# ---------------------------------------------------------------------------------------------------------------------
MLPERF_ROWS = [40000000, 39060, 17295, 7424, 20265, 3, 7122, 1543, 63, 40000000, 3067956, 405282, 10, 2209, 11938, 155, 4, 976, 14,
               40000000, 40000000, 40000000, 590152, 12973, 108, 36]
MLPERF_HOT = [3, 2, 1, 2, 6, 1, 1, 1, 1, 7, 3, 8, 1, 6, 9, 5, 1, 1, 1, 12, 100, 27, 10, 3, 1, 1]


def test_sharding_plan_balances_the_mlperf_v2_tables():
    """The reference's contiguous blocks leave one rank with 1.6x / 2.6x / 5.0x the mean lookup traffic at 2 / 4 / 8 ranks (the
    100-hot 40 M-row table); the planner shards the tables nobody can absorb row-wise and places the rest longest-first."""
    from dlrm_amd import sharding as S
    for world, ref_imb in ((2, 1.6), (4, 2.5), (8, 4.9)):
        p = S.plan(MLPERF_ROWS, MLPERF_HOT, 128, world, 65536)
        assert sorted(s.table for s in p.shards) == list(range(26))                     # every table placed exactly once
        assert 20 in p.row_wise() and p.imbalance() < 1.05
        assert min(p.tables_per_rank()) >= 1 and max(p.rank_bytes) <= 250e9
        for s in p.shards:
            if s.kind == "row":
                assert s.row_ranges[0][0] == 0 and s.row_ranges[-1][1] == MLPERF_ROWS[s.table]
                assert all(a[1] == b[0] for a, b in zip(s.row_ranges, s.row_ranges[1:]))
        own = S.reference_plan(26, world)
        rc = [sum(p.cost[t] for t in range(26) if own[t] == r) for r in range(world)]
        assert max(rc) / (sum(rc) / world) > ref_imb
    assert S.split_rows(10, 4) == ((0, 3), (3, 6), (6, 8), (8, 10))
    one = S.plan([5, 6], [1, 2], 16, 1, 8)
    assert one.row_wise() == [] and one.table_wise(0) == [0, 1]


def _kjt_worker(rank, size, port, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(size), LOCAL_RANK=str(rank))
    from dlrm_amd import ext_dist
    ext_dist.init_distributed(rank=rank, local_rank=rank, size=size, use_gpu=False, backend="gloo")
    hot = [2, 1, 3, 1, 2]
    owner = [1 % size, 0, -1, (size - 1), 0]                  # table 2 is row-wise
    Bl = 4
    # id of (table t, global sample g, slot j) = 10000*t + 10*g + j
    vals = []
    for t, h in enumerate(hot):
        for b in range(Bl):
            g = rank * Bl + b
            vals += [10000 * t + 10 * g + j for j in range(h)]
    tw, rw = ext_dist.kjt_input_dist(torch.tensor(vals, dtype=torch.int32), hot, owner, [2])
    # row-wise partial sums: rank r contributes (r + 1) * ones -> reduce-scatter gives sum, backward all-gathers
    x = torch.full((Bl * size, 3), float(rank + 1), requires_grad=True)
    y = ext_dist.reduce_scatter_rows(x * torch.arange(Bl * size, dtype=torch.float32).view(-1, 1))
    (y * (rank + 1)).sum().backward()
    q.put((rank, {"tw": {t: v.numpy().copy() for t, v in tw.items()}, "rw": {t: v.numpy().copy() for t, v in rw.items()},
                  "y": y.detach().numpy().copy(), "gx": x.grad.numpy().copy()}))
    ext_dist.barrier()
    torch.distributed.destroy_process_group()


@pytest.mark.parametrize("size", [2, 4])
def test_kjt_input_dist_and_row_wise_collectives(size):
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_kjt_worker, args=(r, size, port, q)) for r in range(size)]
    for p in procs:
        p.start()
    results = dict(q.get(timeout=180) for _ in range(size))
    for p in procs:
        p.join(60)
        assert p.exitcode == 0
    hot, Bl = [2, 1, 3, 1, 2], 4
    owner = [1 % size, 0, -1, (size - 1), 0]
    B = Bl * size
    want = lambda t: np.asarray([10000 * t + 10 * g + j for g in range(B) for j in range(hot[t])], dtype=np.int32)
    for r in range(size):
        assert sorted(results[r]["tw"]) == [t for t in range(5) if owner[t] == r]
        for t, v in results[r]["tw"].items():
            assert np.array_equal(v, want(t)), (r, t)                     # whole batch, global order, on the owner only
        assert list(results[r]["rw"]) == [2] and np.array_equal(results[r]["rw"][2], want(2))
        tot = sum(range(1, size + 1))
        rows = np.arange(r * Bl, (r + 1) * Bl, dtype=np.float32).reshape(-1, 1)
        assert np.array_equal(results[r]["y"], np.repeat(rows * tot, 3, axis=1))
        # d/dx[g] = (owner_of_row(g) + 1) * g   (the owner's upstream gradient, all-gathered back to every rank)
        g = np.arange(B, dtype=np.float32)
        assert np.array_equal(results[r]["gx"], np.repeat(((g // Bl + 1) * g).reshape(-1, 1), 3, axis=1))


def _kjt_mlperf_worker(rank, size, port, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(size), LOCAL_RANK=str(rank))
    from dlrm_amd import ext_dist
    from dlrm_amd import sharding as S
    ext_dist.init_distributed(rank=rank, local_rank=rank, size=size, use_gpu=False, backend="gloo")
    plan = S.plan(MLPERF_ROWS, MLPERF_HOT, 128, size, 65536)            # the plan bench.py --gpus N --workload mlperf_v2_multihot builds
    owner = [-1] * 26
    for sh in plan.shards:
        if sh.kind == "table":
            owner[sh.table] = sh.rank
    rw_tables = plan.row_wise()
    Bl = 3
    # id of (table t, global sample g, slot j): unique, so any mis-routed element is visible
    vals = []
    for t, h in enumerate(MLPERF_HOT):
        for b in range(Bl):
            g = rank * Bl + b
            vals += [1000000 * t + 1000 * g + j for j in range(h)]
    tw, rw = ext_dist.kjt_input_dist(torch.tensor(vals, dtype=torch.int64), MLPERF_HOT, owner, rw_tables)
    # row-wise pooling: rank r owns rows [lo, hi) of table 20; its partial sum of a [B, 2] block = the ids of ITS range, summed per sample
    lo, hi = [s_ for s_ in plan.shards if s_.table == rw_tables[0]][0].row_ranges[rank]
    ids = rw[rw_tables[0]].view(Bl * size, -1) % MLPERF_ROWS[rw_tables[0]]
    part = torch.where((ids >= lo) & (ids < hi), ids, torch.zeros_like(ids)).sum(1, keepdim=True).double().repeat(1, 2)
    y = ext_dist.reduce_scatter_rows(part)
    q.put((rank, {"tw": {t: v.numpy().copy() for t, v in tw.items()}, "rw": {t: v.numpy().copy() for t, v in rw.items()},
                  "y": y.numpy().copy(), "owner": owner, "rw_tables": rw_tables}))
    ext_dist.barrier()
    torch.distributed.destroy_process_group()


def test_kjt_input_dist_at_the_mlperf_v2_plan_on_eight_ranks():
    """the input distribution and the row-wise reduce-scatter at the plan the 8-GPU benchmark uses — 26 MLPerf-v2
    tables, multi-hot sizes 3,2,1,...,100,27,... (214 ids per sample), tables 20 and 21 row-wise over all 8 ranks, the other 24
    table-wise longest-first — on 8 gloo ranks: every table-wise owner receives the whole batch's ids of its tables in global
    order, every rank receives the row-wise tables' ids, and the per-range partial sums of a row-wise table add up to the full sum."""
    size = 8
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_kjt_mlperf_worker, args=(r, size, port, q)) for r in range(size)]
    for p in procs:
        p.start()
    results = dict(q.get(timeout=300) for _ in range(size))
    for p in procs:
        p.join(60)
        assert p.exitcode == 0
    Bl = 3
    B = Bl * size
    owner, rw_tables = results[0]["owner"], results[0]["rw_tables"]
    assert rw_tables == [20, 21] and all(o >= 0 for t, o in enumerate(owner) if t not in rw_tables)
    want = lambda t: np.asarray([1000000 * t + 1000 * g + j for g in range(B) for j in range(MLPERF_HOT[t])], dtype=np.int64)
    seen = set()
    for r in range(size):
        assert sorted(results[r]["tw"]) == [t for t in range(26) if owner[t] == r]
        for t, v in results[r]["tw"].items():
            assert np.array_equal(v, want(t)), (r, t)
            seen.add(t)
        assert sorted(results[r]["rw"]) == rw_tables
        for t in rw_tables:
            assert np.array_equal(results[r]["rw"][t], want(t)), (r, t)
        full = (want(20).reshape(B, -1) % MLPERF_ROWS[20]).sum(1)[r * Bl:(r + 1) * Bl].astype(np.float64)
        assert np.array_equal(results[r]["y"], np.repeat(full.reshape(-1, 1), 2, axis=1)), r
    assert len(seen) == 24


def _flat_ddp_worker(rank, size, port, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(size), LOCAL_RANK=str(rank))
    from dlrm_amd import ext_dist
    ext_dist.init_distributed(rank=rank, local_rank=rank, size=size, use_gpu=False, backend="gloo")

    def tower(seed):
        torch.manual_seed(seed)          # rank-dependent on purpose: construction must broadcast rank 0's parameters
        return torch.nn.Sequential(torch.nn.Linear(6, 9), torch.nn.ReLU(), torch.nn.Linear(9, 5), torch.nn.ReLU(), torch.nn.Linear(5, 1))

    a, b = ext_dist.DDP(tower(100 + rank)), ext_dist.FlatDDP(tower(200 + rank))
    with torch.no_grad():
        for pa, pb in zip(a.parameters(), b.parameters()):
            pb.copy_(pa)                 # same starting point (DDP broadcast rank 0's at construction, FlatDDP its own)
    oa, ob = torch.optim.SGD(a.parameters(), lr=0.1), torch.optim.SGD(b.parameters(), lr=0.1)
    g = torch.Generator().manual_seed(7 + rank)
    res = {"grads": [], "params": None, "in_flat": []}
    for step in range(3):
        x = torch.randn(8, 6, generator=g)
        for m, o in ((a, oa), (b, ob)):
            o.zero_grad()
            m(x).pow(2).mean().backward()
        res["grads"].append([(pa.grad - pb.grad).abs().max().item() for pa, pb in zip(a.parameters(), b.parameters())])
        res["in_flat"].append(all(p.grad.data_ptr() == b._view(i).data_ptr() for i, p in enumerate(b._params)))
        oa.step(); ob.step()
    # accumulation over two backward passes without zero_grad: DDP and FlatDDP must still agree
    x1, x2 = torch.randn(8, 6, generator=g), torch.randn(8, 6, generator=g)
    for m, o in ((a, oa), (b, ob)):
        o.zero_grad()
        m(x1).pow(2).mean().backward()
        m(x2).pow(2).mean().backward()
    res["accum"] = [(pa.grad - pb.grad).abs().max().item() for pa, pb in zip(a.parameters(), b.parameters())]
    res["params"] = [p.detach().numpy().copy() for p in b.parameters()]
    res["keys"] = list(b.state_dict().keys())
    # a tower whose second half gets no gradient must raise, not hang or silently skip the collective
    c = ext_dist.FlatDDP(tower(300), broadcast=False)
    try:
        c.module[0](torch.randn(2, 6)).sum().backward()
        res["partial"] = "no error"
    except RuntimeError as e:
        res["partial"] = str(e)
    q.put((rank, res))
    ext_dist.barrier()
    torch.distributed.destroy_process_group()


@pytest.mark.parametrize("size", [2, 3])
def test_flat_ddp_equals_ddp(size):
    """ext_dist.FlatDDP (one flat gradient buffer, one all-reduce launched by the last gradient hook, waited for at the end of
    backward) against torch's DistributedDataParallel on the same tower: averaged gradients over 3 SGD steps, gradient
    accumulation, identical parameters on all ranks, DDP-style state_dict keys."""
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_flat_ddp_worker, args=(r, size, port, q)) for r in range(size)]
    for p in procs:
        p.start()
    results = dict(q.get(timeout=180) for _ in range(size))
    for p in procs:
        p.join(60)
        assert p.exitcode == 0
    for r in range(size):
        for step, gs in enumerate(results[r]["grads"]):
            assert max(gs) < 1e-6, (r, step, gs)
        assert all(results[r]["in_flat"])
        assert max(results[r]["accum"]) < 1e-6, results[r]["accum"]
        assert results[r]["keys"][0] == "module.0.weight"
        assert "parameters received a gradient" in results[r]["partial"]
        for pa, pb in zip(results[0]["params"], results[r]["params"]):
            assert np.array_equal(pa, pb)


# ---------------------------------------------------------------------------------------------------------------------
# a ONE-rank group forced through the distributed path (ext_dist.init_distributed(force=True); tests/test_gpu_rccl.py runs the same
# switch on RCCL): every collective is a self-exchange, i.e. the identity
# ---------------------------------------------------------------------------------------------------------------------
def _forced_one_rank_worker(port, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK="0", WORLD_SIZE="1", LOCAL_RANK="0")
    from dlrm_amd import ext_dist
    assert not ext_dist.is_distributed()
    ext_dist.init_distributed(rank=0, local_rank=0, size=1, use_gpu=False, backend="gloo", force=True)
    res = {"dist": ext_dist.is_distributed(), "size": ext_dist.my_size, "slice": ext_dist.get_my_slice(26),
           "split": ext_dist.get_split_lengths(26)}
    B, D, T = 6, 4, 3
    packed = torch.arange(B * T * D, dtype=torch.float32).view(B, T * D).requires_grad_(True)
    outs = ext_dist.alltoall([packed], None, emb_dim=D).wait()
    (2.0 * outs[0]).sum().backward()
    res["a2a"] = bool(len(outs) == 1 and torch.equal(outs[0], packed.detach()) and torch.equal(packed.grad, torch.full_like(packed, 2.0)))
    x = torch.randn(4, 5, requires_grad=True)
    y = ext_dist.reduce_scatter_rows(x)
    y.sum().backward()
    res["rs"] = bool(torch.equal(y.detach(), x.detach()) and torch.equal(x.grad, torch.ones_like(x)))
    hot = [2, 1]
    vals = torch.arange(3 * 3, dtype=torch.int32)
    tw, rw = ext_dist.kjt_input_dist(vals, hot, [0, -1], [1])
    res["kjt"] = bool(torch.equal(tw[0], vals[:6]) and torch.equal(rw[1], vals[6:]))
    lin = torch.nn.Linear(3, 2)
    f = ext_dist.FlatDDP(lin)
    f(torch.ones(5, 3)).sum().backward()
    res["flat"] = bool(torch.allclose(lin.weight.grad, torch.full((2, 3), 5.0)))
    q.put(res)
    ext_dist.barrier()
    torch.distributed.destroy_process_group()


def test_forced_one_rank_group_takes_the_distributed_path():
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    p = ctx.Process(target=_forced_one_rank_worker, args=(_free_port(), q))
    p.start()
    res = q.get(timeout=180)
    p.join(60)
    assert p.exitcode == 0
    assert res["dist"] and res["size"] == 1 and res["slice"] == slice(0, 26, 1) and res["split"] == (26, None)
    assert res["a2a"] and res["rs"] and res["kjt"] and res["flat"]
