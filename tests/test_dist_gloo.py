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


@pytest.mark.parametrize("T,B,D", [(3, 8, 4), (4, 6, 2)])
def test_alltoall_layouts_two_ranks(T, B, D):
    size = 2
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
