"""HIP kernels (through the C ABI, dlrm_amd.ops) against the CPU oracle on identical seeded inputs.
Integer/byte-order contracts are bit-exact; MFMA fp32 GEMMs/dots use the tolerance written in each test."""
import os

import numpy as np
import pytest
import torch

from oracle import oracle as O

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

pytestmark = pytest.mark.gpu


def dev():
    assert torch.cuda.is_available(), "GPU tests need a real MI355X"
    return torch.device("cuda:0")


def ragged(rng, B, rows, max_len, empty_frac=0.2):
    lens = rng.integers(0, max_len + 1, size=B)
    lens[rng.random(B) < empty_frac] = 0
    off = np.concatenate([[0], np.cumsum(lens)[:-1]]).astype(np.int64)
    idx = rng.integers(0, rows, size=int(lens.sum())).astype(np.int64)
    return off, idx


def to_dev(a, dtype=None):
    t = torch.from_numpy(np.ascontiguousarray(a))
    if dtype is not None:
        t = t.to(dtype)
    return t.to(dev())


# ------------------------------------------------------------------------------------------ embeddings
@pytest.mark.parametrize("D", [1, 2, 12, 16, 64, 128, 200, 256, 512])
@pytest.mark.parametrize("idx_dtype", [torch.int64, torch.int32])
def test_emb_fwd_bit_exact(D, idx_dtype):
    from dlrm_amd import ops
    rng = np.random.default_rng(D)
    rows = [1, 3, 50, 1000, 7]
    B = 203
    Ws = [rng.standard_normal((n, D)).astype(np.float32) for n in rows]
    bags = [ragged(rng, B, n, 9) for n in rows]
    bags[1] = (np.arange(B, dtype=np.int64), rng.integers(0, 3, size=B).astype(np.int64))  # one-hot table
    bags[4] = (np.zeros(B, dtype=np.int64), np.zeros(0, dtype=np.int64))                    # all bags empty
    psw = [None, None, rng.standard_normal(bags[2][1].shape[0]).astype(np.float32), None, None]
    want = np.concatenate([O.emb_fwd(W, i, o, psw=w) for W, (o, i), w in zip(Ws, bags, psw)], axis=1)
    dW = [to_dev(W) for W in Ws]
    bb = ops.BagBatch([to_dev(o, idx_dtype) for o, _ in bags], [to_dev(i, idx_dtype) for _, i in bags],
                      [None if w is None else to_dev(w) for w in psw])
    # write into a strided slot of a wider buffer, like the [B, F*D] interaction buffer
    buf = torch.full((B, (len(rows) + 1) * D + 4), -7.0, device=dev())
    out = buf[:, D:D + len(rows) * D]
    ops.emb_fwd(dW, bb, out)
    torch.cuda.synchronize()
    got = out.cpu().numpy()
    assert np.array_equal(got, want)
    assert torch.all(buf[:, :D] == -7.0) and torch.all(buf[:, D + len(rows) * D:] == -7.0)


@pytest.mark.parametrize("idx_dtype", [torch.int64, torch.int32])
def test_emb_stacked_inputs_equal_lists(idx_dtype):
    """Stacked [T, B] offsets/indices (the reference's collate layout) take the pointer-arithmetic path of BagBatch:
    same results as the list form, also for a row-sliced (table-sharded) view."""
    from dlrm_amd import ops
    rng = np.random.default_rng(77)
    rows, B, D = [5, 1000, 33, 70000], 333, 32
    Ws = [to_dev(rng.standard_normal((n, D)).astype(np.float32)) for n in rows]
    idx = np.stack([rng.integers(0, n, size=B) for n in rows]).astype(np.int64)
    off = np.tile(np.arange(B, dtype=np.int64), (len(rows), 1))
    want = np.concatenate([O.emb_fwd(W.cpu().numpy(), idx[t], off[t]) for t, W in enumerate(Ws)], axis=1)
    I, Ofs = to_dev(idx, idx_dtype), to_dev(off, idx_dtype)
    out = torch.empty(B, len(rows) * D, device=dev())
    ops.emb_fwd(Ws, ops.BagBatch(Ofs, I), out)
    assert np.array_equal(out.cpu().numpy(), want)
    out2 = torch.empty(B, 2 * D, device=dev())
    ops.emb_fwd(Ws[1:3], ops.BagBatch(Ofs[1:3], I[1:3]), out2)
    assert np.array_equal(out2.cpu().numpy(), want[:, D:3 * D])
    dV = to_dev(rng.standard_normal((B, len(rows) * D)).astype(np.float32))
    W1 = [w.clone() for w in Ws]
    W2 = [w.clone() for w in Ws]
    ops.emb_bwd_sgd(W1, ops.BagBatch(Ofs, I), dV, 0.1, ops.UPD_DETERMINISTIC)
    ops.emb_bwd_sgd(W2, ops.BagBatch([Ofs[t] for t in range(4)], [I[t] for t in range(4)]), dV, 0.1, ops.UPD_DETERMINISTIC)
    for a, b in zip(W1, W2):
        assert torch.equal(a, b)


@pytest.mark.parametrize("D", [2, 16, 128, 200])
def test_emb_bwd_sgd_deterministic_bit_exact(D):
    from dlrm_amd import ops
    rng = np.random.default_rng(100 + D)
    rows = [3, 40, 500]
    B = 150
    Ws = [rng.standard_normal((n, D)).astype(np.float32) for n in rows]
    bags = [ragged(rng, B, n, 5, empty_frac=0.1) for n in rows]
    psw = [None, rng.standard_normal(bags[1][1].shape[0]).astype(np.float32), None]
    dV = rng.standard_normal((B, len(rows) * D)).astype(np.float32)
    want = [O.emb_bwd_sgd(W.copy(), i, o, np.ascontiguousarray(dV[:, t * D:(t + 1) * D]), 0.3, psw=w)
            for t, (W, (o, i), w) in enumerate(zip(Ws, bags, psw))]
    dW = [to_dev(W) for W in Ws]
    bb = ops.BagBatch([to_dev(o) for o, _ in bags], [to_dev(i) for _, i in bags],
                      [None if w is None else to_dev(w) for w in psw])
    ops.emb_bwd_sgd(dW, bb, to_dev(dV), 0.3, ops.UPD_DETERMINISTIC)
    torch.cuda.synchronize()
    for t in range(len(rows)):
        assert np.array_equal(dW[t].cpu().numpy(), want[t]), t


@pytest.mark.parametrize("D,rows", [(16, [3, 40, 5000]), (128, [4, 10, 130, 100000]), (6, [2, 9])])
@pytest.mark.parametrize("mode_name", ["atomic", "sorted"])
def test_emb_bwd_sgd_atomic_matches_oracle(D, rows, mode_name):
    """fast modes: duplicate-row sums are re-associated (fp32 atomics / LDS pre-reduction / sorted runs) -> tolerance"""
    from dlrm_amd import ops
    rng = np.random.default_rng(7 + D)
    B = 4096
    Ws = [rng.standard_normal((n, D)).astype(np.float32) for n in rows]
    bags = [(np.arange(B, dtype=np.int64), rng.integers(0, n, size=B).astype(np.int64)) for n in rows]
    bags[-1] = ragged(rng, B, rows[-1], 6)
    dV = (rng.standard_normal((B, len(rows) * D)) * 0.1).astype(np.float32)
    want = [O.emb_bwd_sgd(W.copy(), i, o, np.ascontiguousarray(dV[:, t * D:(t + 1) * D]), 0.05)
            for t, (W, (o, i)) in enumerate(zip(Ws, bags))]
    dW = [to_dev(W) for W in Ws]
    bb = ops.BagBatch([to_dev(o) for o, _ in bags], [to_dev(i) for _, i in bags])
    ops.emb_bwd_sgd(dW, bb, to_dev(dV), 0.05, ops.UPD_ATOMIC if mode_name == "atomic" else ops.UPD_SORTED)
    torch.cuda.synchronize()
    for t in range(len(rows)):
        np.testing.assert_allclose(dW[t].cpu().numpy(), want[t], rtol=1e-5, atol=2e-5, err_msg=str(t))


@pytest.mark.parametrize("D", [4, 16, 128, 200, 512])
@pytest.mark.parametrize("idx_dtype", [torch.int64, torch.int32])
def test_emb_bwd_sgd_sorted_large_tables_bit_exact(D, idx_dtype):
    """sorted mode: rows whose run sits inside one chunk follow the reference's per-lookup fma chain exactly;
    with large tables (few duplicates) every row does"""
    from dlrm_amd import ops
    rng = np.random.default_rng(31 + D)
    rows = [100003, 7, 250000]
    B = 2000
    Ws = [rng.standard_normal((n, D)).astype(np.float32) for n in rows]
    bags = [ragged(rng, B, n, 3, empty_frac=0.1) for n in rows]
    bags[1] = (np.zeros(B, dtype=np.int64), np.zeros(0, dtype=np.int64))      # a table without lookups
    dV = rng.standard_normal((B, len(rows) * D)).astype(np.float32)
    want = [O.emb_bwd_sgd(W.copy(), i, o, np.ascontiguousarray(dV[:, t * D:(t + 1) * D]), 0.2)
            for t, (W, (o, i)) in enumerate(zip(Ws, bags))]
    dW = [to_dev(W) for W in Ws]
    bb = ops.BagBatch([to_dev(o, idx_dtype) for o, _ in bags], [to_dev(i, idx_dtype) for _, i in bags])
    ops.emb_bwd_sgd(dW, bb, to_dev(dV), 0.2, ops.UPD_SORTED)
    torch.cuda.synchronize()
    for t in range(len(rows)):
        got = dW[t].cpu().numpy()
        # duplicates of one row can straddle a chunk boundary (atomic re-association): allow a few such rows
        bad = np.unique(np.nonzero(got != want[t])[0])
        touched = np.unique(bags[t][1]).size
        assert bad.size <= max(8, touched // 50), (t, bad.size, touched)
        np.testing.assert_allclose(got, want[t], rtol=1e-5, atol=1e-5)


CRITEO_TB_ROWS = [39884406, 39043, 17289, 7420, 20263, 3, 7120, 1543, 63, 38532951, 2953546, 403346, 10, 2208, 11938, 155,
                  4, 976, 14, 39979771, 25641295, 39664984, 585935, 12972, 108, 36]


@pytest.mark.parametrize("case", ["criteo_onehot", "ragged_hot_rows_int32", "wide_keys_three_rounds", "tiny_tables_many_tiles",
                                  "long_segment_general_sorter", "single_lookup", "mlperf_v2_100hot_segment"])
def test_lookup_sort_is_stable_and_exact(case, monkeypatch):
    """The (table, row) sort in front of the sort-based updates (csrc/seg_sort.h; dlrm_emb_sort_lookups) against numpy's stable
    argsort of the same keys — positions, keys and the bag of every position, exactly: one-hot Criteo tables (1- and 2-round tables
    mixed), ragged multi-hot bags with hot rows and an empty table (int32), 64-bit keys over three rounds, segments of many tiles
    over 1- and 2-row tables (every cursor hit by every lane), a 300 k-lookup segment (three tile groups) and the 6.55 M-lookup segment of
    the MLPerf-v2 batch's 100-hot table (narrower digits, 25 tile groups, hot row)."""
    from dlrm_amd import ops
    rng = np.random.default_rng(len(case))
    idx_dtype = torch.int64
    # ("mlperf_v2_100hot_segment" and "long_segment_general_sorter" hold table segments of more than 262144 lookups: the product library hands those
    # to the general sorter — DESIGN.md section 6, round 6: the segmented sorter's long-segment path exists in tuning builds only
    # (DLRM_HIP_LIB=... DLRM_SORT=own tools/sort_bench.py v2) — and the contract checked here is the same: stable, exact.)
    if case == "criteo_onehot":
        rows, B = CRITEO_TB_ROWS, 5000
        bags = [(np.arange(B, dtype=np.int64), rng.integers(0, n, size=B).astype(np.int64)) for n in rows]
    elif case == "ragged_hot_rows_int32":
        rows, B, idx_dtype = [3, 40, 9, 5000, 100003], 3000, torch.int32
        bags = [ragged(rng, B, n, 6) for n in rows]
        bags[2] = (np.zeros(B, dtype=np.int64), np.zeros(0, dtype=np.int64))
        bags[4][1][::3] = 77                                           # a hot row in a big table
    elif case == "wide_keys_three_rounds":
        rows, B = [1 << 34, 5, 1 << 27], 2000
        bags = [ragged(rng, B, n, 3, empty_frac=0.0) for n in rows]
    elif case == "tiny_tables_many_tiles":
        rows, B = [1, 2, 70000], 70000
        bags = [(np.arange(B, dtype=np.int64), rng.integers(0, n, size=B).astype(np.int64)) for n in rows]
    elif case == "long_segment_general_sorter":
        rows, B = [1000, 50], 300000
        bags = [(np.arange(B, dtype=np.int64), rng.integers(0, n, size=B).astype(np.int64)) for n in rows]
    elif case == "mlperf_v2_100hot_segment":
        # config 5's longest segments: the 100-hot 40 M-row table of a 65536-sample batch = 6.55 M lookups (3200 tiles, 25 tile groups,
        # 26 row bits in three 9/9/8-bit rounds) next to a 27-hot one and a one-hot tiny table; int32 ids as the KJT carries them
        rows, B, idx_dtype = [40000000, 3, 40000000], 65536, torch.int32
        hots = [100, 1, 27]
        bags = [(np.arange(B, dtype=np.int64) * h, rng.integers(0, n, size=B * h).astype(np.int64)) for n, h in zip(rows, hots)]
        bags[0][1][::7] = 12345                                         # a hot row: 936 k equal keys in input order
    else:
        rows, B = [7], 1
        bags = [(np.zeros(1, dtype=np.int64), np.asarray([5], dtype=np.int64))]
    bb = ops.BagBatch([to_dev(o, idx_dtype) for o, _ in bags], [to_dev(i, idx_dtype) for _, i in bags])
    pos, keys, bag_of, rb = ops.sort_lookups(rows, bb)
    torch.cuda.synchronize()
    assert rb == max(1, int(np.ceil(np.log2(max(rows))))) or (1 << rb) >= max(rows)
    want_keys = np.concatenate([(np.int64(t) << rb) | i for t, (_, i) in enumerate(bags)]) if sum(len(i) for _, i in bags) else np.zeros(0, np.int64)
    order = np.argsort(want_keys, kind="stable")
    assert np.array_equal(pos.cpu().numpy(), order), case
    assert np.array_equal(keys.cpu().numpy(), want_keys[order]), case
    want_bag = np.concatenate([np.searchsorted(o, np.arange(len(i)), side="right") - 1 for o, i in bags])
    assert np.array_equal(bag_of.cpu().numpy(), want_bag), case
    ops.check_index_errors(sync=True)


# ------------------------------------------------------------------------------------------ interaction
@pytest.mark.parametrize("F,D,itself", [(4, 16, False), (27, 128, False), (27, 16, False), (6, 12, True), (2, 2, False),
                                        (9, 64, False), (33, 32, False), (49, 8, True),
                                        # D = 128: LDS-DMA double-buffered path (odd / even / full / tiny F, self pairs)
                                        (27, 128, True), (32, 128, False), (3, 128, False), (1, 128, True), (16, 128, False)])
@pytest.mark.parametrize("B", [67, 5000])
def test_interact_fwd_bwd(F, D, itself, B):
    from dlrm_amd import ops
    if B > 67 and D != 128:
        pytest.skip("large batch only exercises the multi-sample-per-wave loop of the D = 128 path")
    rng = np.random.default_rng(F * 1000 + D)
    feat = rng.standard_normal((B, F, D)).astype(np.float32)
    want = O.interact_fwd(feat, itself)
    Wd = want.shape[1]
    ldr = (Wd + 3) & ~3
    x = to_dev(feat[:, 0, :])
    E = to_dev(feat[:, 1:, :].reshape(B, (F - 1) * D))
    R = torch.full((B, ldr), 3.0, device=dev())
    ops.interact_fwd([x, E], D, itself, R)
    torch.cuda.synchronize()
    got = R.cpu().numpy()
    # fp32 MFMA vs fp64-accumulated oracle: a 128-term fp32 dot of N(0,1) data carries ~1e-5 absolute round-off in its
    # worst element out of millions, so the absolute floor scales with the number of outputs checked
    atol = 1e-5 if B <= 67 else 3e-5
    np.testing.assert_allclose(got[:, :Wd], want, rtol=1e-5, atol=atol)
    assert np.array_equal(got[:, :D], feat[:, 0, :])                      # the copied x block is exact
    assert np.all(got[:, Wd:] == 0)
    dR = rng.standard_normal((B, Wd)).astype(np.float32)
    dwant = O.interact_bwd(feat, dR, itself)
    dRd = torch.zeros((B, ldr), device=dev())
    dRd[:, :Wd] = to_dev(dR)
    dx = torch.empty((B, D), device=dev())
    dE = torch.empty((B, (F - 1) * D), device=dev())
    ops.interact_bwd([x, E], D, itself, dRd, [dx, dE])
    torch.cuda.synchronize()
    np.testing.assert_allclose(dx.cpu().numpy(), dwant[:, 0, :], rtol=1e-5, atol=2 * atol)
    np.testing.assert_allclose(dE.cpu().numpy().reshape(B, F - 1, D), dwant[:, 1:, :], rtol=1e-5, atol=2 * atol)


@pytest.mark.parametrize("widths", [[4, 4, 3, 3, 3, 3, 3, 3], [13, 13], [1] * 26, [7, 7, 6, 6]],
                         ids=["config4_8ranks", "2ranks", "26ranks", "4ranks"])
@pytest.mark.parametrize("B", [8192, 333])
def test_interact_fwd_bwd_over_all_to_all_blocks(widths, B):
    """BASELINE configs[3] at its OWN shapes: after the forward all-to-all a rank's interaction reads the bottom-MLP output
    plus one receive block per peer (dlrm_s_pytorch.py:528-585; 26 tables over 8 ranks = widths 4,4,3,3,3,3,3,3,
    extend_distributed.py:47-62), each block its own allocation with its own leading dimension, D = 128, B/N = 8192 rows;
    the backward writes each block's gradient into a strided chunk of the reverse exchange's send buffer.  Checked directly
    against the oracle (dlrm_s_pytorch.py:483-504), forward and backward."""
    from dlrm_amd import ops
    D, T = 128, sum(widths)
    F = T + 1
    rng = np.random.default_rng(B + len(widths))
    feat = rng.standard_normal((B, F, D)).astype(np.float32)
    want = O.interact_fwd(feat, False)
    Wd = want.shape[1]
    ldr = (Wd + 3) & ~3
    # x lives inside a wider activation buffer (row pitch > D); every receive block has its own pitch: block j is padded by
    # 4 * (j % 3) floats so that no two consecutive blocks share a leading dimension
    xbuf = torch.full((B, D + 8), -7.0, device=dev())
    x = xbuf[:, :D]
    x.copy_(to_dev(feat[:, 0, :]))
    blocks, t0 = [x], 1
    for j, w in enumerate(widths):
        buf = torch.full((B, w * D + 4 * (j % 3)), -9.0, device=dev())
        blk = buf[:, :w * D]
        blk.copy_(to_dev(feat[:, t0:t0 + w, :].reshape(B, w * D)))
        blocks.append(blk)
        t0 += w
    R = torch.full((B, ldr), 3.0, device=dev())
    ops.interact_fwd(blocks, D, False, R)
    torch.cuda.synchronize()
    got = R.cpu().numpy()
    atol = 3e-5
    np.testing.assert_allclose(got[:, :Wd], want, rtol=1e-5, atol=atol)
    assert np.array_equal(got[:, :D], feat[:, 0, :])
    assert np.all(got[:, Wd:] == 0)
    # backward: the gradient of block j lands in a column chunk of ONE flat send buffer [B, T*D + pad] (the layout
    # ext_dist's reverse all-to-all sends in place), the gradient of x in its own strided buffer
    dR = rng.standard_normal((B, Wd)).astype(np.float32)
    dwant = O.interact_bwd(feat, dR, False)
    dRd = torch.zeros((B, ldr), device=dev())
    dRd[:, :Wd] = to_dev(dR)
    dxbuf = torch.full((B, D + 4), 5.0, device=dev())
    send = torch.full((B, T * D + 12), 6.0, device=dev())
    dblocks, c0 = [dxbuf[:, :D]], 0
    for w in widths:
        dblocks.append(send[:, c0:c0 + w * D])
        c0 += w * D
    ops.interact_bwd(blocks, D, False, dRd, dblocks)
    torch.cuda.synchronize()
    np.testing.assert_allclose(dxbuf[:, :D].cpu().numpy(), dwant[:, 0, :], rtol=1e-5, atol=2 * atol)
    np.testing.assert_allclose(send[:, :T * D].cpu().numpy().reshape(B, T, D), dwant[:, 1:, :], rtol=1e-5, atol=2 * atol)
    assert torch.all(dxbuf[:, D:] == 5.0) and torch.all(send[:, T * D:] == 6.0)      # nothing written outside the chunks
    assert torch.all(xbuf[:, D:] == -7.0)


# ------------------------------------------------------------------------------------------ MLP layers
@pytest.mark.parametrize("M,N,K,act", [(128, 512, 13, 1), (300, 16, 512, 1), (257, 1024, 479, 1), (64, 1, 256, 2),
                                       (1, 3, 2, 0), (513, 130, 36, 1), (1000, 128, 256, 1),
                                       # LDS-DMA fast path: 128-row tiles, 256-row tiles (edge tiles in M and N), split-K
                                       (512, 256, 64, 1), (4096, 200, 48, 2), (65536, 512, 256, 1), (66000, 384, 272, 1)])
@pytest.mark.parametrize("arith", ["f32", "bf16x6"])
def test_linear_fwd_bwd(M, N, K, act, arith):
    """both MLP arithmetics against the float64 oracle at the SAME fp32-class tolerances: "bf16x6" (exact 3-term bf16
    split of the fp32 operands, 6 bf16 MFMA products, fp32 accumulation) must not be distinguishable from fp32 MFMA"""
    _linear_fwd_bwd(M, N, K, act, arith)


def _linear_fwd_bwd(M, N, K, act, arith):
    from dlrm_amd import ops
    rng = np.random.default_rng(M + N + K)
    # asymmetric, non-identity data so that a transposed fragment cannot pass
    X = rng.standard_normal((M, K)).astype(np.float32)
    W = (rng.standard_normal((N, K)) / np.sqrt(K)).astype(np.float32)
    b = rng.standard_normal(N).astype(np.float32)
    Y = O.linear_fwd(X, W, b, act)
    ldx = (K + 3) & ~3
    Xd = torch.zeros((M, ldx), device=dev())[:, :K]
    Xd.copy_(to_dev(X))
    Wd, bd = to_dev(W), to_dev(b)
    Yd = torch.empty((M, N), device=dev())
    ops.linear_fwd(Xd, Wd, bd, act, Yd, arith)
    torch.cuda.synchronize()
    np.testing.assert_allclose(Yd.cpu().numpy(), Y, rtol=1e-5, atol=1e-5)

    dY = rng.standard_normal((M, N)).astype(np.float32)
    # the activation mask is taken from the forward output the GPU produced: an output within rounding of 0 may fall on
    # either side of the ReLU threshold, and a flipped mask bit is a forward-rounding artefact, not a backward error
    dX, dW, db = O.linear_bwd(X, W, act, Yd.cpu().numpy(), dY)
    dZd = torch.empty((M, N), device=dev())
    ops.act_bwd(to_dev(dY), Yd, act, dZd, None)
    dWd = torch.empty((N, K), device=dev())
    dbd = torch.full((N,), 7.0, device=dev())          # overwritten, not accumulated
    ops.linear_bwd_weight(dZd, Xd, dWd, dbd, arith=arith)
    dWa = torch.empty((N, K), device=dev())             # same GEMM, k-slices accumulated with atomics (no workspace)
    ops.linear_bwd_weight(dZd, Xd, dWa, None, use_workspace=False, arith=arith)
    dXd = torch.empty((M, ldx), device=dev())[:, :K]
    ops.linear_bwd_data(dZd, Wd, None, 0, dXd, arith)
    torch.cuda.synchronize()
    scale = max(1.0, float(np.abs(dW).max()))
    np.testing.assert_allclose(dbd.cpu().numpy(), db, rtol=1e-4, atol=1e-4 * max(1.0, float(np.abs(db).max())))
    np.testing.assert_allclose(dWd.cpu().numpy(), dW, rtol=1e-4, atol=1e-5 * scale)
    np.testing.assert_allclose(dWa.cpu().numpy(), dW, rtol=1e-4, atol=1e-5 * scale)
    np.testing.assert_allclose(dXd.cpu().numpy(), dX, rtol=1e-5, atol=1e-5)


@pytest.mark.parametrize("M,widths,acts,need_dx", [
    (2048, [13, 512, 256, 64, 16], [1, 1, 1, 1], False),          # Criteo-Kaggle bottom tower: unaligned input rows (52 bytes)
    (2048, [368, 512, 256, 1], [1, 1, 2], True),                  # ... top tower (the interaction's padded 367 columns), sigmoid head
    (128, [13, 512, 16], [1, 1], False), (128, [22, 512, 256, 1], [1, 1, 2], True),
    (1000, [7, 33, 130, 5], [1, 0, 2], True),                     # nothing aligned, ragged batch (1000 = 62 * 16 + 8), mixed activations
    (17, [64, 512, 512, 64], [1, 1, 1], True), (1, [4, 4], [0], True), (4096, [16, 64, 48], [2, 1], True)])
def test_small_batch_tower_kernels_match_oracle(M, widths, acts, need_dx):
    """dlrm_tower_fwd / _bwd / _wgrad (csrc/tower.hip: a whole MLP per launch, activations of 16 rows in LDS, the weight gradients of all
    layers from one launch with an in-order slice sum) against the float64 oracle applied layer by layer (O.linear_fwd / O.linear_bwd,
    dlrm_s_pytorch.py:208-246, 399-405), at the tolerances of the per-layer GEMM test; the weight gradient twice: same bits (deterministic)."""
    from dlrm_amd import ops
    rng = np.random.default_rng(M + sum(widths))
    L = len(acts)
    X = rng.standard_normal((M, widths[0])).astype(np.float32)
    Ws = [(rng.standard_normal((widths[l + 1], widths[l])) / np.sqrt(widths[l])).astype(np.float32) for l in range(L)]
    bs = [rng.standard_normal(widths[l + 1]).astype(np.float32) for l in range(L)]
    Xd = to_dev(X)
    Wd, bd = [to_dev(w) for w in Ws], [to_dev(b) for b in bs]
    outs = [torch.full((M, widths[l + 1]), 7.0, device=dev()) for l in range(L)]
    ops.tower_fwd(Xd, Wd, bd, acts, outs)
    torch.cuda.synchronize()
    cur = X
    for l in range(L):          # every layer checked on the input the GPU actually fed it (errors do not compound into the bars)
        want = O.linear_fwd(cur, Ws[l], bs[l], acts[l])
        np.testing.assert_allclose(outs[l].cpu().numpy(), want, rtol=1e-5, atol=1e-5, err_msg="layer %d forward" % l)
        cur = outs[l].cpu().numpy()
    dY = rng.standard_normal((M, widths[L])).astype(np.float32)
    dZs = [torch.full((M, widths[l + 1]), 7.0, device=dev()) for l in range(L)]
    dX = torch.full((M, widths[0]), 7.0, device=dev()) if need_dx else None
    ops.tower_bwd(to_dev(dY), Wd, acts, outs, dZs, dX)
    dWs = [torch.full((widths[l + 1], widths[l]), 7.0, device=dev()) for l in range(L)]
    dbs = [torch.full((widths[l + 1],), 7.0, device=dev()) for l in range(L)]
    ops.tower_wgrad(dZs, [Xd] + outs[:-1], dWs, dbs)
    dWs2 = [torch.empty_like(w) for w in dWs]
    dbs2 = [torch.empty_like(b) for b in dbs]
    ops.tower_wgrad(dZs, [Xd] + outs[:-1], dWs2, dbs2)
    torch.cuda.synchronize()
    g = dY
    for l in range(L - 1, -1, -1):
        inp = X if l == 0 else outs[l - 1].cpu().numpy()
        dXo, dWo, dbo = O.linear_bwd(inp, Ws[l], acts[l], outs[l].cpu().numpy(), g)
        # dZ of this layer as the GPU has it (mask from the GPU's own forward output, as in the per-layer test)
        y = outs[l].cpu().numpy().astype(np.float64)
        dz = g * ((y > 0) if acts[l] == 1 else (y * (1 - y)) if acts[l] == 2 else 1.0)
        np.testing.assert_allclose(dZs[l].cpu().numpy(), dz, rtol=1e-5, atol=1e-5, err_msg="layer %d dZ" % l)
        scale = max(1.0, float(np.abs(dWo).max()))
        np.testing.assert_allclose(dWs[l].cpu().numpy(), dWo, rtol=1e-4, atol=1e-5 * scale, err_msg="layer %d dW" % l)
        np.testing.assert_allclose(dbs[l].cpu().numpy(), dbo, rtol=1e-4, atol=1e-4 * max(1.0, float(np.abs(dbo).max())), err_msg="layer %d db" % l)
        assert torch.equal(dWs[l], dWs2[l]) and torch.equal(dbs[l], dbs2[l])
        if l == 0:
            if need_dx:
                np.testing.assert_allclose(dX.cpu().numpy(), dXo, rtol=1e-5, atol=1e-5)
        else:
            g = dZs[l].cpu().numpy().astype(np.float64) @ Ws[l].astype(np.float64)       # the gradient the next (lower) layer receives, from the GPU's dZ
            g = g.astype(np.float32)
    # the consumer-applied form: dY taken as dL/dz of the last layer
    dZb = [torch.empty_like(z) for z in dZs]
    ops.tower_bwd(dZs[L - 1].clone(), Wd, acts, outs, dZb, None, last_act_applied=True)
    torch.cuda.synchronize()
    assert all(torch.equal(a_, b_) for a_, b_ in zip(dZs, dZb))
    # a narrower first weight gradient: the input's trailing padding columns are dropped
    if widths[0] > 4:
        dW0 = torch.full((widths[1], widths[0] - 1), 7.0, device=dev())
        ops.tower_wgrad(dZs, [Xd] + outs[:-1], [dW0] + dWs2[1:], dbs2)
        torch.cuda.synchronize()
        assert torch.equal(dW0, dWs[0][:, :widths[0] - 1])
    with pytest.raises(RuntimeError):
        ops.tower_fwd(Xd, Wd, bd, acts, outs[:-1])


@pytest.mark.parametrize("M,N,K,Nn", [(65536, 512, 256, 128), (1000, 256, 64, 96), (4100, 200, 48, 64), (333, 130, 36, 16), (4096, 1, 256, 32)])
def test_relu_sign_bits_replace_the_fp32_mask(M, N, K, Nn):
    """dlrm_linear_fwd(relu_bits=...) stores one sign bit per output element in the documented 32 x 64-block layout (fast
    LDS-DMA kernel, any-shape fallback and the N = 1 matrix-vector path alike); dlrm_linear_bwd_data fed with those bits
    instead of the fp32 activation produces the IDENTICAL data gradient."""
    from dlrm_amd import ops
    rng = np.random.default_rng(M + N)
    X = to_dev(rng.standard_normal((M, K)).astype(np.float32))
    W = to_dev((rng.standard_normal((N, K)) / np.sqrt(K)).astype(np.float32))
    b = to_dev(rng.standard_normal(N).astype(np.float32))
    Y = torch.empty((M, N), device=dev())
    bits = ops.relu_bits_alloc(M, N, dev())
    bits.fill_(-1)
    ops.linear_fwd(X, W, b, 1, Y, relu_bits=bits)
    torch.cuda.synchronize()
    y = Y.cpu().numpy() > 0
    nblk = (N + 63) // 64
    words = bits.cpu().numpy().view(np.uint32).reshape(-1, nblk, 64)             # [row band, column block, lane]
    mp, npad = ((M + 31) // 32) * 32, nblk * 64
    yp = np.zeros((mp, npad), dtype=bool)
    yp[:M, :N] = y
    # element (32*mb + 4*it + l//16, 64*nb + 4*(l%16) + c) <-> bit 31 - (4*it + c) of word [mb, nb, l]
    e = yp.reshape(mp // 32, 8, 4, nblk, 16, 4)                                  # [mb, it, l//16, nb, l%16, c]
    e = e.transpose(0, 3, 2, 4, 1, 5).reshape(mp // 32, nblk, 64, 32)            # [mb, nb, l, 4*it + c]
    want = (e.astype(np.uint64) << (31 - np.arange(32, dtype=np.uint64))).sum(-1).astype(np.uint32)
    valid = np.zeros((mp, npad), dtype=bool)
    valid[:M, :N] = True                                                         # bits of elements outside the matrix are unspecified
    vm = valid.reshape(mp // 32, 8, 4, nblk, 16, 4).transpose(0, 3, 2, 4, 1, 5).reshape(mp // 32, nblk, 64, 32)
    vmask = (vm.astype(np.uint64) << (31 - np.arange(32, dtype=np.uint64))).sum(-1).astype(np.uint32)
    assert np.array_equal(words & vmask, want)
    # consumer: the next layer (N -> Nn) back-propagates into this activation
    W2 = to_dev((rng.standard_normal((Nn, N)) / np.sqrt(N)).astype(np.float32))
    dZ = to_dev(rng.standard_normal((M, Nn)).astype(np.float32))
    d_float = torch.empty((M, N), device=dev())
    d_bits = torch.empty((M, N), device=dev())
    ops.linear_bwd_data(dZ, W2, Y, 1, d_float)
    ops.linear_bwd_data(dZ, W2, Y, 1, d_bits, relu_bits=bits)
    assert torch.equal(d_float, d_bits)
    assert bool(torch.all(d_bits[~(Y > 0)] == 0))


@pytest.mark.parametrize("M,N,K,act", [(4096, 512, 256, 1), (1000, 320, 192, 1), (65536, 128, 256, 0), (300, 1024, 480, 1), (8192, 256, 512, 1)])
def test_straight_line_epilogue_equals_general_epilogue(M, N, K, act):
    """Round 6: `gemm3_kernel`'s straight-line epilogue (EPI 1 / 2: what the host selects for aligned plain-store calls — every layer of the
    headline) against the general epilogue of the same kernel, on the same operands: a bias vector that starts 4 bytes off a 16-byte
    boundary is all it takes to send the call down the general path (csrc/gemm.hip launch_gemm `fast_epi`).  Outputs AND ReLU sign bits must
    be identical bit for bit — same products, same order, same bias add.  (The data gradient's pair — fp32 mask = general, sign bits =
    straight-line — is `test_relu_sign_bits_replace_the_fp32_mask`.)"""
    from dlrm_amd import ops
    rng = np.random.default_rng(M + N + K)
    X = to_dev(rng.standard_normal((M, K)).astype(np.float32))
    W = to_dev((rng.standard_normal((N, K)) / np.sqrt(K)).astype(np.float32))
    b = rng.standard_normal(N).astype(np.float32)
    b_al = to_dev(b)
    holder = torch.empty(N + 1, device=dev())
    b_un = holder[1:]
    b_un.copy_(b_al)
    assert b_al.data_ptr() % 16 == 0 and b_un.data_ptr() % 16 == 4
    outs = []
    for bias in (b_al, b_un):
        Y = torch.full((M, N), 7.0, device=dev())
        bits = ops.relu_bits_alloc(M, N, dev()) if act == 1 else None
        if bits is not None:
            bits.fill_(0)
        ops.linear_fwd(X, W, bias, act, Y, relu_bits=bits)
        torch.cuda.synchronize()
        outs.append((Y, bits))
    assert torch.equal(outs[0][0], outs[1][0])
    if act == 1:
        # (bits of elements outside the matrix are unspecified: compare through the consumer — the next layer's data gradient)
        dZ = to_dev(rng.standard_normal((M, 64)).astype(np.float32))
        W2 = to_dev(rng.standard_normal((64, N)).astype(np.float32))
        d = [torch.empty((M, N), device=dev()) for _ in range(2)]
        for i in range(2):
            ops.linear_bwd_data(dZ, W2, outs[i][0], 1, d[i], relu_bits=outs[i][1])
        assert torch.equal(d[0], d[1])
    want = X.double().cpu().numpy() @ W.double().cpu().numpy().T + b.astype(np.float64)
    if act == 1:
        want = np.maximum(want, 0)
    np.testing.assert_allclose(outs[0][0].cpu().numpy(), want, rtol=2e-5, atol=2e-5)


def test_linear_bwd_data_fused_mask():
    """dgrad epilogue: previous layer's ReLU mask fused in (aligned and unaligned leading dimensions)"""
    from dlrm_amd import ops
    rng = np.random.default_rng(5)
    M, N, K = 777, 96, 200
    dZ = rng.standard_normal((M, N)).astype(np.float32)
    W = rng.standard_normal((N, K)).astype(np.float32)
    Xact = np.maximum(rng.standard_normal((M, K)), 0).astype(np.float32)
    want = (dZ.astype(np.float64) @ W.astype(np.float64)) * (Xact > 0)
    dXd = torch.empty((M, K), device=dev())
    ops.linear_bwd_data(to_dev(dZ), to_dev(W), to_dev(Xact), 1, dXd)
    torch.cuda.synchronize()
    np.testing.assert_allclose(dXd.cpu().numpy(), want, rtol=1e-5, atol=2e-5)
    # unaligned: K = 199 (ld 199) forces the scalar epilogue
    K2 = 199
    dX2 = torch.empty((M, K2), device=dev())
    ops.linear_bwd_data(to_dev(dZ), to_dev(W[:, :K2].copy()), to_dev(Xact[:, :K2].copy()), 1, dX2)
    np.testing.assert_allclose(dX2.cpu().numpy(), want[:, :K2], rtol=1e-5, atol=2e-5)
    # accumulate mode of the weight gradient
    X = rng.standard_normal((M, K)).astype(np.float32)
    dW = torch.ones((N, K), device=dev())
    db = torch.ones(N, device=dev())
    ops.linear_bwd_weight(to_dev(dZ), to_dev(X), dW, db, accumulate=True)
    np.testing.assert_allclose(dW.cpu().numpy(), 1.0 + dZ.astype(np.float64).T @ X.astype(np.float64), rtol=1e-4, atol=2e-3)
    np.testing.assert_allclose(db.cpu().numpy(), 1.0 + dZ.astype(np.float64).sum(0), rtol=1e-4, atol=1e-3)


@pytest.mark.parametrize("M,N,K", [(512, 256, 512), (4096, 1024, 480), (304, 128, 256)])
def test_linear_bf16_arithmetic(M, N, K):
    """"bf16" MLP arithmetic (BASELINE configs[4]): equals a float64 product of the bf16-ROUNDED operands (nearest even,
    the rounding of torch's bfloat16 conversion) up to fp32 accumulation error, for forward, data and weight gradient;
    and differs from the fp32 result by about the bf16 operand precision — it is a different arithmetic, not a bug.
    (Every reduction length here is a multiple of 16: other shapes run the any-shape fp32 kernel, include/dlrm_hip.h.)"""
    from dlrm_amd import ops

    def rb(a):        # round fp32 -> bf16 -> fp32 exactly like the kernel (and torch)
        return torch.from_numpy(a).to(torch.bfloat16).to(torch.float32).numpy().astype(np.float64)
    rng = np.random.default_rng(M + N + K)
    X = rng.standard_normal((M, K)).astype(np.float32)
    W = (rng.standard_normal((N, K)) / np.sqrt(K)).astype(np.float32)
    b = rng.standard_normal(N).astype(np.float32)
    dY = rng.standard_normal((M, N)).astype(np.float32)
    Xd, Wd, bd, dYd = to_dev(X), to_dev(W), to_dev(b), to_dev(dY)
    Y, dX, dW, db = (torch.empty((M, N), device=dev()), torch.empty((M, K), device=dev()), torch.empty((N, K), device=dev()),
                     torch.empty(N, device=dev()))
    ops.linear_fwd(Xd, Wd, bd, 0, Y, "bf16")
    ops.linear_bwd_data(dYd, Wd, None, 0, dX, "bf16")
    ops.linear_bwd_weight(dYd, Xd, dW, db, arith="bf16")
    torch.cuda.synchronize()
    np.testing.assert_allclose(Y.cpu().numpy(), rb(X) @ rb(W).T + b, rtol=2e-5, atol=2e-5)
    np.testing.assert_allclose(dX.cpu().numpy(), rb(dY) @ rb(W), rtol=2e-5, atol=2e-5)
    np.testing.assert_allclose(dW.cpu().numpy(), rb(dY).T @ rb(X), rtol=1e-4, atol=1e-4 * np.sqrt(M))
    np.testing.assert_allclose(db.cpu().numpy(), dY.astype(np.float64).sum(0), rtol=1e-4, atol=1e-3)   # bias grad stays fp32
    err = np.abs(Y.cpu().numpy() - (X.astype(np.float64) @ W.astype(np.float64).T + b)).max()
    assert 1e-4 < err < 0.1, err


@pytest.mark.parametrize("M,N,K,act", [(512, 256, 512, 1), (1000, 320, 192, 1), (4100, 1024, 1024, 0), (65536, 512, 256, 1), (777, 3456, 512, 0),
                                       (2048, 512, 3456, 1), (256, 192, 64, 2)])
def test_gemm_bf16_phased_kernel(M, N, K, act, monkeypatch):
    """dlrm_gemm_bf16 on the bf16-SHAPED kernel (csrc/gemm_bf16.hip: 256 x 256 x 64 tile, four phases per k-tile, two wave halves one
    barrier apart) — forward form (bias, activation, fp32 + bf16 results, ReLU sign bits out) and data-gradient form (sign-bit mask in):
      * against a float64 product of the SAME bf16 operands (the oracle's arithmetic: exact products, wide accumulation), and
      * BIT-IDENTICAL to the fp32-shaped kernel (DLRM_BF16_PHASED=0 is read once per process, so the reference result comes from
        dlrm_linear_fwd's in-loop rounding, ARITH bf16, which the storage kernels are tested to equal bit for bit): same MFMA, same k order.
    Shapes: ragged M and N (clamped rows / columns), N = 3456 and K = 3456 (the DCN-v2 products of config 5), one tile, many k-tiles."""
    from dlrm_amd import ops
    rng = np.random.default_rng(M + N + K)
    A = torch.from_numpy(rng.standard_normal((M, K)).astype(np.float32)).to(dev()).to(torch.bfloat16)
    B = torch.from_numpy((rng.standard_normal((N, K)) / np.sqrt(K)).astype(np.float32)).to(dev()).to(torch.bfloat16)
    bias = to_dev(rng.standard_normal(N).astype(np.float32))
    Cf = torch.full((M, N), 7.0, device=dev())
    Cb = torch.zeros((M, N), dtype=torch.bfloat16, device=dev())
    bits = ops.relu_bits_alloc(M, N, dev()) if act == 1 else None
    ops.gemm_bf16(A, B, bias, act, Cf, Cb, relu_bits_out=bits)
    torch.cuda.synchronize()
    want = A.double().cpu().numpy() @ B.double().cpu().numpy().T + bias.double().cpu().numpy()
    if act == 1:
        want = np.maximum(want, 0.0)
    elif act == 2:
        want = 1.0 / (1.0 + np.exp(-want))
    got = Cf.cpu().numpy()
    np.testing.assert_allclose(got, want, rtol=2e-5, atol=3e-5)
    assert torch.equal(Cb, Cf.to(torch.bfloat16))                              # the bf16 copy is the rounding of the fp32 result
    # the same product through dlrm_linear_fwd with in-loop rounding of fp32 operands that ARE bf16 values: bit-identical
    Y2 = torch.empty((M, N), device=dev())
    ops.linear_fwd(A.float(), B.float(), bias, act, Y2, "bf16")
    assert torch.equal(Y2, Cf)
    if act == 1:
        # the sign bits drive the data gradient of the NEXT layer: dX = (dY . W) * (Y > 0), here with Y = Cf [M, N], dY [M, N2], W [N2, N]
        N2 = 256
        dY = torch.from_numpy(rng.standard_normal((M, N2)).astype(np.float32)).to(dev()).to(torch.bfloat16)
        Wt = torch.from_numpy((rng.standard_normal((N, N2)) / 16).astype(np.float32)).to(dev()).to(torch.bfloat16)     # = W^T [N, N2]
        dX = torch.empty((M, N), device=dev())
        dXb = torch.empty((M, N), dtype=torch.bfloat16, device=dev())
        ops.gemm_bf16(dY, Wt, None, 0, dX, dXb, relu_bits_in=bits, category="linear_bwd_data")
        torch.cuda.synchronize()
        wantd = (dY.double().cpu().numpy() @ Wt.double().cpu().numpy().T) * (got > 0)
        np.testing.assert_allclose(dX.cpu().numpy(), wantd, rtol=2e-5, atol=3e-5)
        assert torch.equal(dXb, dX.to(torch.bfloat16))


@pytest.mark.parametrize("M,N,K", [(4096, 1024, 1024), (65536, 512, 256), (1024, 320, 192), (8192, 128, 256), (2048, 3456, 512), (256, 64, 64),
                                   (16384, 1024, 480), (4096, 1024, 479)])
def test_linear_bwd_weight_bf16_from_stored_operands(M, N, K):
    """dlrm_linear_bwd_weight_bf16 (csrc/gemm_bf16.hip, weight-gradient form: both operands k-strided, fragments by ds_read_b64_tr_b16, batch
    split into fp32 slabs summed in slice order) against a float64 product of the SAME bf16 operands (AddmmBackward's weight / bias
    branch, dlrm_s_pytorch.py:1613): dW = dZ^T X, db = column sums of dZ; overwrite and accumulate forms; run-to-run bit-identical.
    Shapes: ragged output tiles (N, K not multiples of 256), one tile, the 3456-wide DCN-v2 product, K = 480 (the padded interaction width)."""
    from dlrm_amd import ops
    rng = np.random.default_rng(M + N + K)
    dZ = torch.from_numpy(rng.standard_normal((M, N)).astype(np.float32)).to(dev()).to(torch.bfloat16)
    Kp = (K + 31) & ~31                                        # activations are stored at widths that are multiples of 32; columns K.. are zero
    X = torch.zeros((M, Kp), dtype=torch.bfloat16, device=dev())
    X[:, :K] = torch.from_numpy(rng.standard_normal((M, K)).astype(np.float32)).to(dev()).to(torch.bfloat16)
    assert ops.linear_bwd_weight_bf16_ok(M, N, K, dZ, X)
    dW = torch.full((N, K), 7.0, device=dev())
    db = torch.full((N,), 7.0, device=dev())
    ops.linear_bwd_weight_bf16(dZ, X, dW, db)
    torch.cuda.synchronize()
    want = dZ.double().cpu().numpy().T @ X[:, :K].double().cpu().numpy()
    wdb = dZ.double().cpu().numpy().sum(0)
    np.testing.assert_allclose(dW.cpu().numpy(), want, rtol=1e-4, atol=2e-4 * np.sqrt(M))
    np.testing.assert_allclose(db.cpu().numpy(), wdb, rtol=1e-4, atol=2e-4 * np.sqrt(M))
    dW2, db2 = dW.clone(), db.clone()
    ops.linear_bwd_weight_bf16(dZ, X, dW2, db2, accumulate=True)
    dW3, db3 = torch.empty_like(dW), torch.empty_like(db)
    ops.linear_bwd_weight_bf16(dZ, X, dW3, db3)
    torch.cuda.synchronize()
    assert torch.equal(dW3, dW) and torch.equal(db3, db)                       # deterministic
    np.testing.assert_allclose(dW2.cpu().numpy(), 2 * want, rtol=1e-4, atol=4e-4 * np.sqrt(M))
    np.testing.assert_allclose(db2.cpu().numpy(), 2 * wdb, rtol=1e-4, atol=4e-4 * np.sqrt(M))
    # same product as the fp32-storage weight gradient with in-loop rounding (ARITH bf16) of operands that already are bf16 values
    dW4 = torch.empty((N, K), device=dev())
    ops.linear_bwd_weight(dZ.float(), X[:, :K].float().contiguous() if K % 4 else X.float()[:, :K], dW4, None, arith="bf16")
    np.testing.assert_allclose(dW.cpu().numpy(), dW4.cpu().numpy(), rtol=1e-4, atol=2e-4 * np.sqrt(M))


def test_bf16x3_split_is_exact_and_padded():
    """dlrm_split_bf16x3 / _transposed: the three truncation planes of an fp32 matrix (gemm.hip split3) — h + m + l == x EXACTLY for every
    finite value (tiny, huge, negative, zero), h = the upper 16 bits of x; padding columns are zero in all planes; the 8-wide and the
    scalar kernel, ragged widths."""
    from dlrm_amd import ops
    g = torch.Generator(device="cpu").manual_seed(5)
    for M, N, Np in ((300, 13, 16), (65, 479, 480), (1000, 256, 256), (513, 3456, 3456), (129, 480, 512), (3, 8, 64)):
        x = torch.randn(M, N, generator=g) * torch.exp(torch.randn(M, N, generator=g) * 8)          # 30 binary orders of magnitude
        x[0, 0], x[-1, -1] = 0.0, -1.00390625
        x = x.to(dev())
        p = ops.split_bf16x3(x, Np)
        assert tuple(p.shape) == (3, M, Np)
        assert torch.equal((p[0, :, :N].float() + p[1, :, :N].float()) + p[2, :, :N].float(), x), (M, N, Np)
        assert torch.equal(p[:, :, :N].contiguous().view(torch.int16), _planes_ref(x).view(torch.int16))
        assert torch.equal(p[0, :, :N].view(torch.int16).int() & 0xffff, (x.view(torch.int32) >> 16) & 0xffff)
        assert not p[:, :, N:].view(torch.int16).any()
    for R, C_, Rp in ((1024, 480, 1024), (100, 37, 128), (1, 256, 32)):
        w = torch.randn(R, C_, generator=g).to(dev())
        p = ops.split_bf16x3_transposed(w, Rp)
        assert tuple(p.shape) == (3, C_, Rp)
        assert torch.equal((p[0, :, :R].float() + p[1, :, :R].float()) + p[2, :, :R].float(), w.t())
        assert not p[:, :, R:].view(torch.int16).any()


def _planes_ref(x):
    """the truncation planes of an fp32 tensor in plain torch: [3, ...] bf16 (every conversion below is exact)"""
    h = (x.view(torch.int32) & -65536).view(torch.float32)
    r = x - h
    m = (r.view(torch.int32) & -65536).view(torch.float32)
    return torch.stack([h.to(torch.bfloat16), m.to(torch.bfloat16), (r - m).to(torch.bfloat16)])


@pytest.mark.parametrize("M,N,K,act", [(512, 256, 512, 1), (1000, 320, 192, 1), (4100, 1024, 1024, 0), (65536, 512, 256, 1), (777, 452, 208, 1),
                                       (2048, 512, 3456, 0), (256, 192, 64, 2), (16384, 1024, 480, 1), (8192, 512, 16, 1)])
def test_gemm_bf16x6_from_planes_equals_in_loop_split(M, N, K, act):
    """dlrm_gemm_bf16x6 (csrc/gemm_bf16.hip PL = 3: operands as three pre-split bf16 planes, six MFMAs per 16 k in the four-phase pipeline)
    is BIT-IDENTICAL to dlrm_linear_fwd / dlrm_linear_bwd_data with DLRM_ARITH_BF16X6, which split the same fp32 operands inside their
    k-loops (same six products, same order, same k partition) — and therefore inherits their parity with the fp64 oracle (test_linear_fwd_bwd):
      * forward: bias, activation, fp32 result, the planes of the result (== split of the fp32 result), ReLU sign bits;
      * planes-only output (the lean tower's hidden layers, 16-byte stores);
      * data gradient: product with a transposed-weight planes operand, masked by the sign bits.
    Shapes: ragged M / N, one tile, K = 3456, K = 480 (the padded interaction width), K = 16 (ONE k-tile: the 13 -> 16 dense-feature layer)."""
    from dlrm_amd import ops
    rng = np.random.default_rng(M + N + K)
    assert ops.gemm_bf16x6_ok(M, N, K)
    X = to_dev(rng.standard_normal((M, K)).astype(np.float32))
    W = to_dev((rng.standard_normal((N, K)) / np.sqrt(K)).astype(np.float32))
    bias = to_dev(rng.standard_normal(N).astype(np.float32))
    Y_ref = torch.empty((M, N), device=dev())
    bits_ref = ops.relu_bits_alloc(M, N, dev()).zero_() if act == 1 else None          # (zeroed: words of ragged edge blocks are compared too)
    ops.linear_fwd(X, W, bias, act, Y_ref, "bf16x6", relu_bits=bits_ref)
    X3, W3 = ops.split_bf16x3(X, K), ops.split_bf16x3(W, K)
    Y = torch.full((M, N), 7.0, device=dev())
    Y3 = torch.zeros((3, M, N), dtype=torch.bfloat16, device=dev())
    bits = ops.relu_bits_alloc(M, N, dev()).zero_() if act == 1 else None
    ops.gemm_bf16x6(X3, W3, bias, act, Y, Y3, relu_bits_out=bits)
    torch.cuda.synchronize()
    assert torch.equal(Y, Y_ref)
    assert torch.equal(Y3.view(torch.int16), _planes_ref(Y).view(torch.int16))
    if bits is not None and M % 32 == 0 and N % 64 == 0:       # (bits of rows / columns past the edge are unspecified; ragged shapes: checked through the data gradient below)
        assert torch.equal(bits, bits_ref)
    if N % 8 == 0:
        Y3b = torch.zeros((3, M, N), dtype=torch.bfloat16, device=dev())
        bits_b = ops.relu_bits_alloc(M, N, dev()).zero_() if act == 1 else None
        ops.gemm_bf16x6(X3, W3, bias, act, None, Y3b, relu_bits_out=bits_b)
        assert torch.equal(Y3b, Y3)
        if bits_b is not None and M % 32 == 0 and N % 64 == 0:
            assert torch.equal(bits_b, bits)
    if act == 1:
        # data gradient of the NEXT layer: dX = (dY . W2) * (Y > 0) with Y [M, N], dY [M, N2], W2 [N2, N]
        N2 = 256
        dY = to_dev(rng.standard_normal((M, N2)).astype(np.float32))
        W2 = to_dev((rng.standard_normal((N2, N)) / 16).astype(np.float32))
        dX_ref = torch.empty((M, N), device=dev())
        ops.linear_bwd_data(dY, W2, Y_ref, 1, dX_ref, "bf16x6", relu_bits=bits_ref)      # (small shapes mask by the fp32 activation, large ones by the bits)
        dX = torch.empty((M, N), device=dev())
        dX3 = torch.zeros((3, M, N), dtype=torch.bfloat16, device=dev())
        ops.gemm_bf16x6(ops.split_bf16x3(dY, N2), ops.split_bf16x3_transposed(W2, N2), None, 0, dX, dX3, relu_bits_in=bits, category="linear_bwd_data")
        torch.cuda.synchronize()
        assert torch.equal(dX, dX_ref)
        assert torch.equal(dX3.view(torch.int16), _planes_ref(dX).view(torch.int16))


@pytest.mark.parametrize("M,N,K", [(4096, 1024, 1024), (65536, 512, 256), (1024, 320, 192), (8192, 128, 256), (2048, 3456, 512), (256, 64, 64),
                                   (16384, 1024, 480), (4096, 1024, 479)])
def test_linear_bwd_weight_bf16x6_from_planes(M, N, K):
    """dlrm_linear_bwd_weight_bf16x6 (weight-gradient form of the planes kernel: both operands k-strided through ds_read_b64_tr_b16, six MFMAs per
    16 batch rows, fp32 slabs summed in slice order) against a float64 product of the fp32 operands (AddmmBackward's weight / bias branch,
    dlrm_s_pytorch.py:1613) at the tolerance the fp32 MFMA weight gradient is held to, and against the in-loop-split kernel; overwrite /
    accumulate; run-to-run bit-identical."""
    from dlrm_amd import ops
    rng = np.random.default_rng(M + N + K)
    dZ = to_dev(rng.standard_normal((M, N)).astype(np.float32))
    Kp = (K + 15) & ~15
    X = torch.zeros((M, Kp), device=dev())
    X[:, :K] = to_dev(rng.standard_normal((M, K)).astype(np.float32))
    dZ3, X3 = ops.split_bf16x3(dZ, N), ops.split_bf16x3(X, Kp)
    dW = torch.full((N, K), 7.0, device=dev())
    db = torch.full((N,), 7.0, device=dev())
    ops.linear_bwd_weight_bf16x6(dZ3, X3, dW, db)
    torch.cuda.synchronize()
    want = dZ.double().cpu().numpy().T @ X[:, :K].double().cpu().numpy()
    wdb = dZ.double().cpu().numpy().sum(0)
    np.testing.assert_allclose(dW.cpu().numpy(), want, rtol=1e-4, atol=1e-5 * np.sqrt(M))
    np.testing.assert_allclose(db.cpu().numpy(), wdb, rtol=1e-4, atol=1e-5 * np.sqrt(M))
    dW2, db2 = dW.clone(), db.clone()
    ops.linear_bwd_weight_bf16x6(dZ3, X3, dW2, db2, accumulate=True)
    dW3, db3 = torch.empty_like(dW), torch.empty_like(db)
    ops.linear_bwd_weight_bf16x6(dZ3, X3, dW3, db3)
    torch.cuda.synchronize()
    assert torch.equal(dW3, dW) and torch.equal(db3, db)                       # deterministic
    np.testing.assert_allclose(dW2.cpu().numpy(), 2 * want, rtol=1e-4, atol=2e-5 * np.sqrt(M))
    np.testing.assert_allclose(db2.cpu().numpy(), 2 * wdb, rtol=1e-4, atol=2e-5 * np.sqrt(M))
    dW4 = torch.empty((N, K), device=dev())
    ops.linear_bwd_weight(dZ, X[:, :K].contiguous() if K % 4 else X[:, :K], dW4, None, arith="bf16x6")
    np.testing.assert_allclose(dW.cpu().numpy(), dW4.cpu().numpy(), rtol=1e-4, atol=1e-5 * np.sqrt(M))


def test_bf16_casts_are_round_to_nearest_even_and_padded():
    """dlrm_cast_bf16 / dlrm_cast_bf16_transposed against torch's fp32 -> bfloat16 conversion (round to nearest even), including the
    zero padding columns and odd shapes"""
    from dlrm_amd import ops
    g = torch.Generator(device="cpu").manual_seed(3)
    for M, N, Np in ((300, 13, 32), (65, 479, 480), (1000, 256, 256), (7, 1, 2), (513, 3456, 3456), (129, 480, 512), (3, 8, 64)):   # last three: the 8-wide kernel, with padding groups
        x = (torch.randn(M, N, generator=g) * 3).to(dev())
        x[0, 0] = 1.00390625          # exactly half way between two bf16 values: ties to even
        got = ops.cast_bf16(x, Np)
        want = torch.zeros(M, Np, dtype=torch.bfloat16, device=dev())
        want[:, :N] = x.to(torch.bfloat16)
        assert torch.equal(got.view(torch.int16), want.view(torch.int16)), (M, N, Np)
    for R, C_, Rp in ((1024, 480, 1024), (100, 37, 128), (1, 256, 32)):
        w = torch.randn(R, C_, generator=g).to(dev())
        got = ops.cast_bf16_transposed(w, Rp)
        want = torch.zeros(C_, Rp, dtype=torch.bfloat16, device=dev())
        want[:, :R] = w.t().to(torch.bfloat16)
        assert torch.equal(got.view(torch.int16), want.view(torch.int16)), (R, C_, Rp)


def test_bf16_multi_cast_equals_the_single_casts():
    """dlrm_cast_bf16_multi (round 5: the bf16 copies of ALL weights of a tower — row-major for the forward GEMMs, transposed for the data
    gradients — in one launch) against dlrm_cast_bf16 / dlrm_cast_bf16_transposed tensor by tensor: same bits, same zero padding; odd shapes,
    a strided source, only-one-copy requests, more tensors than one launch holds."""
    from dlrm_amd import ops
    rng = np.random.default_rng(77)
    shapes = [(512, 16), (256, 512), (128, 256), (1024, 480), (33, 70), (1, 5), (64, 64)] * 3          # 21 tensors > DLRM_CAST_MULTI_MAX = 16
    items, want = [], []
    for k, (R, C_) in enumerate(shapes):
        wide = to_dev(rng.standard_normal((R, C_ + 3)).astype(np.float32))
        src = wide[:, 1:1 + C_] if k % 4 == 1 else to_dev(rng.standard_normal((R, C_)).astype(np.float32))
        cpad = ops.round_bf16_k(C_) if k % 3 != 2 else None
        rpad = ((R + 7) & ~7) if k % 3 != 1 else None
        items.append((src, cpad, rpad))
        want.append((ops.cast_bf16(src.contiguous(), cpad) if cpad else None, ops.cast_bf16_transposed(src.contiguous(), rpad) if rpad else None))
    got = ops.cast_bf16_multi(items)
    torch.cuda.synchronize()
    assert len(got) == len(items)
    for k, ((d, dT), (w, wT)) in enumerate(zip(got, want)):
        assert (d is None) == (w is None) and (dT is None) == (wT is None), k
        if w is not None:
            assert d.shape == w.shape and torch.equal(d.view(torch.int16), w.view(torch.int16)), k
        if wT is not None:
            assert dT.shape == wT.shape and torch.equal(dT.view(torch.int16), wT.view(torch.int16)), k


@pytest.mark.parametrize("ln,B", [([13, 512, 256, 128], 4096), ([479, 1024, 1024, 512, 256, 1], 2048), ([13, 64, 48, 16], 300), ([96, 64, 32], 129),
                                  ([64, 256, 128, 64], 65536)])
def test_bf16_storage_tower_is_bit_identical_to_in_loop_rounding(ln, B):
    """arith "bf16" with bf16 STORAGE (dlrm_gemm_bf16: activations / weights read as bf16 copies, nothing converted in the k-loop, the
    data gradient over a transposed bf16 weight copy) against the in-loop rounding path of rounds 1-2 (dlrm_linear_fwd / _bwd_data with
    DLRM_ARITH_BF16): same operand rounding, same accumulation order -> outputs, input gradient and every parameter gradient equal
    bit for bit; and both within bf16 tolerance of an fp64 reference.  (Layer widths are multiples of 16: for other widths the in-loop
    path falls back to the any-shape fp32 kernel and is MORE precise than bf16, so there is nothing to be identical to; 48 exercises
    the zero padding of a reduction length to the next multiple of 32.)"""
    from dlrm_amd import functional, ops
    from dlrm_amd.functional import MLPFunction
    rng = np.random.default_rng(sum(ln))
    L = len(ln) - 1
    params = []
    for i in range(L):
        params += [to_dev((rng.standard_normal((ln[i + 1], ln[i])) * np.sqrt(2 / (ln[i] + ln[i + 1]))).astype(np.float32)).requires_grad_(True),
                   to_dev((rng.standard_normal(ln[i + 1]) * 0.1).astype(np.float32)).requires_grad_(True)]
    acts = tuple([ops.ACT_RELU] * (L - 1) + [ops.ACT_SIGMOID if ln[-1] == 1 else ops.ACT_RELU])
    x0 = to_dev(rng.random((B, ln[0])).astype(np.float32))
    dy = to_dev(rng.standard_normal((B, ln[-1])).astype(np.float32))
    results = []
    saved = functional.BF16_STORAGE, functional.BF16_LEAN
    try:
        for storage, lean in ((False, False), (True, False), (True, True)):
            functional.BF16_STORAGE, functional.BF16_LEAN = storage, lean
            x = x0.clone().requires_grad_(True)
            for p in params:
                p.grad = None
            y = MLPFunction.apply(x, acts, None, ops.arith_code("bf16"), *params)
            y.backward(dy)
            torch.cuda.synchronize()
            results.append((y.detach().clone(), x.grad.clone(), [p.grad.clone() for p in params]))
    finally:
        functional.BF16_STORAGE, functional.BF16_LEAN = saved
    (y0, dx0, g0), (y1, dx1, g1), (y2, dx2, g2) = results
    assert torch.equal(y0, y1), float((y0 - y1).abs().max())
    assert torch.equal(dx0, dx1), float((dx0 - dx1).abs().max())
    for a, b in zip(g0, g1):
        assert torch.equal(a, b)
    # LEAN storage (hidden activations / gradients only as bf16 + sign bits; weight gradient from the bf16 operands as stored, k-strided
    # reads): every forward and data-gradient product has the same operands and the same k order -> y and dx still bit-identical; the
    # weight gradient sums the same bf16 products in another slice order and the bias gradient sums bf16-rounded dZ -> fp32 round-off /
    # one bf16 rounding per term apart
    assert torch.equal(y0, y2), float((y0 - y2).abs().max())
    assert torch.equal(dx0, dx2), float((dx0 - dx2).abs().max())
    for k, (a, b) in enumerate(zip(g0, g2)):
        scale = float(a.abs().max())
        np.testing.assert_allclose(b.cpu().numpy(), a.cpu().numpy(), rtol=1e-3, atol=(2e-3 if k % 2 else 2e-5) * scale, err_msg="param %d" % k)
    # and the arithmetic is what it claims to be: bf16 operands, fp32 accumulation
    h = x0.double().cpu().numpy()
    for i in range(L):
        h = h @ params[2 * i].detach().double().cpu().numpy().T + params[2 * i + 1].detach().double().cpu().numpy()
        h = 1 / (1 + np.exp(-h)) if acts[i] == ops.ACT_SIGMOID else np.maximum(h, 0)
    np.testing.assert_allclose(y1.cpu().numpy(), h, rtol=6e-2, atol=6e-2 * float(np.abs(h).max()))


@pytest.mark.parametrize("ln,B", [([13, 512, 256, 128], 65536), ([479, 1024, 1024, 512, 256, 1], 4096), ([480, 1024, 512, 256], 2048), ([96, 64, 32], 512),
                                  ([256, 320, 192, 64], 1000)])
def test_bf16x6_planes_tower_equals_in_loop_split(ln, B):
    """arith "bf16x6" with PLANES (functional.BF16X6_PLANES: activations / gradients / weights of the GEMM layers travel as three pre-split bf16
    planes, hidden layers keep planes + sign bits only, weight gradients from the planes as stored) against the same tower with every GEMM
    splitting its fp32 operands in its k-loop (the kernels of rounds 1-3): every forward and data-gradient product has the same six bf16
    terms in the same order -> output and input gradient BIT-IDENTICAL; weight / bias gradients sum the same products in another slice
    order -> fp32 round-off apart.  Towers: the two Terabyte towers (13 -> 16 first layer on the small-K kernel, a 128-wide and a 1-wide
    layer on the fp32-storage kernels in between planes layers), widths the planes kernel refuses (N < 192), ragged batch."""
    from dlrm_amd import functional, ops
    from dlrm_amd.functional import MLPFunction
    rng = np.random.default_rng(sum(ln) + B)
    L = len(ln) - 1
    params = []
    for i in range(L):
        params += [to_dev((rng.standard_normal((ln[i + 1], ln[i])) * np.sqrt(2 / (ln[i] + ln[i + 1]))).astype(np.float32)).requires_grad_(True),
                   to_dev((rng.standard_normal(ln[i + 1]) * 0.1).astype(np.float32)).requires_grad_(True)]
    acts = tuple([ops.ACT_RELU] * (L - 1) + [ops.ACT_SIGMOID if ln[-1] == 1 else ops.ACT_RELU])
    x0 = to_dev(rng.random((B, ln[0])).astype(np.float32))
    dy = to_dev(rng.standard_normal((B, ln[-1])).astype(np.float32))
    results = []
    saved = functional.BF16X6_PLANES, functional._PlaneStore.gemm
    calls = [0]

    def counting_gemm(*a, **k):
        calls[0] += 1
        return ops.gemm_bf16x6(*a, **k)
    try:
        functional._PlaneStore.gemm = staticmethod(counting_gemm)
        for planes in (False, True):
            functional.BF16X6_PLANES = planes
            x = x0.clone().requires_grad_(True)
            for p in params:
                p.grad = None
            y = MLPFunction.apply(x, acts, None, ops.arith_code("bf16x6"), *params)
            y.backward(dy)
            torch.cuda.synchronize()
            results.append((y.detach().clone(), x.grad.clone(), [p.grad.clone() for p in params]))
    finally:
        functional.BF16X6_PLANES, functional._PlaneStore.gemm = saved[0], staticmethod(saved[1])
    eligible = sum(1 for i in range(L) if ln[i + 1] >= 192 and B >= 256)
    assert (calls[0] > 0) == (eligible > 0), (calls, eligible)              # the planes kernels really ran (forward + data gradients)
    (y0, dx0, g0), (y1, dx1, g1) = results
    assert torch.equal(y0, y1), float((y0 - y1).abs().max())
    assert torch.equal(dx0, dx1), float((dx0 - dx1).abs().max())
    for k, (a, b) in enumerate(zip(g0, g1)):
        scale = float(a.abs().max())
        np.testing.assert_allclose(b.cpu().numpy(), a.cpu().numpy(), rtol=1e-4, atol=2e-6 * scale * max(1.0, np.sqrt(B / 256)), err_msg="param %d" % k)
    # and against fp64: the fp32 round-off class
    h = x0.double().cpu().numpy()
    for i in range(L):
        h = h @ params[2 * i].detach().double().cpu().numpy().T + params[2 * i + 1].detach().double().cpu().numpy()
        h = 1 / (1 + np.exp(-h)) if acts[i] == ops.ACT_SIGMOID else np.maximum(h, 0)
    np.testing.assert_allclose(y1.cpu().numpy(), h, rtol=1e-5, atol=1e-5 * float(np.abs(h).max()))


@pytest.mark.parametrize("arith", ["bf16", "bf16x6"])
@pytest.mark.parametrize("ln", [[128, 256, 200, 64], [128, 256, 72, 64], [64, 320, 200, 72, 32]])
def test_lean_towers_keep_the_fp32_activation_where_the_data_gradient_falls_back(ln, arith):
    """hidden widths the bf16 / plane DATA-gradient GEMM refuses (200, 72: multiples of 8 that are not multiples of 32)
    while the bf16 WEIGHT gradient takes them.  The lean plan used to drop the fp32 activation below such a layer on the strength of
    the weight gradient alone; the fp32-storage data-gradient fallback then ran without its ReLU mask and every gradient below was
    wrong.  The plan now keeps the fp32 copy (need32 looks at dg16 as well) and ops.linear_bwd_data refuses a derivative without its
    activation.  Against an fp64 restatement with the ReLU masks applied: input gradient and all parameter gradients."""
    from dlrm_amd import ops
    from dlrm_amd.functional import MLPFunction
    rng = np.random.default_rng(sum(ln))
    B, L = 2048, len(ln) - 1
    params = []
    for i in range(L):
        params += [to_dev((rng.standard_normal((ln[i + 1], ln[i])) * np.sqrt(2 / (ln[i] + ln[i + 1]))).astype(np.float32)).requires_grad_(True),
                   to_dev((rng.standard_normal(ln[i + 1]) * 0.1).astype(np.float32)).requires_grad_(True)]
    acts = tuple([ops.ACT_RELU] * L)
    x = to_dev(rng.random((B, ln[0])).astype(np.float32)).requires_grad_(True)
    dy = to_dev(rng.standard_normal((B, ln[-1])).astype(np.float32))
    y = MLPFunction.apply(x, acts, None, ops.arith_code(arith), *params)
    y.backward(dy)
    torch.cuda.synchronize()
    hs = [x.detach().double().cpu().numpy()]
    for i in range(L):
        hs.append(np.maximum(hs[-1] @ params[2 * i].detach().double().cpu().numpy().T + params[2 * i + 1].detach().double().cpu().numpy(), 0))
    g = dy.double().cpu().numpy()
    # bf16: operands rounded to 8 bits; a pre-activation within that rounding of zero flips its ReLU unit against the fp64 restatement, which
    # moves that SAMPLE's input-gradient row by O(1) and single parameter-gradient entries by a few per cent (measured: 0.1 % of the dx
    # entries, 2 of 51200 dW entries) — so bf16 is compared in the Frobenius norm; the bug this test guards against (a data gradient
    # without its ReLU mask) changes about half of ALL entries below the layer by their own size (relative norm error ~0.7).  The bar of 0.2:
    # these gradients are sums of 2048 random-sign terms, which amplifies the 2^-9 operand rounding — bf16 vs fp32 arithmetic of the SAME
    # kernels differs by 0.04-0.10 in this norm for every tensor, at widths 192 / 224 as at 200 (tools/probes/lean_width_probe.py)
    def close(got, want, what):
        if arith == "bf16":
            err = float(np.linalg.norm(got.astype(np.float64) - want) / max(np.linalg.norm(want), 1e-30))
            assert err <= 0.2, (what, err)
        else:
            np.testing.assert_allclose(got, want, rtol=2e-5, atol=2e-5 * float(np.abs(want).max()), err_msg=what)
    close(y.detach().cpu().numpy(), hs[-1], "y")
    for i in range(L - 1, -1, -1):
        g = g * (hs[i + 1] > 0)
        close(params[2 * i].grad.cpu().numpy(), g.T @ hs[i], "dW %d" % i)
        close(params[2 * i + 1].grad.cpu().numpy(), g.sum(0), "db %d" % i)
        g = g @ params[2 * i].detach().double().cpu().numpy()
    close(x.grad.cpu().numpy(), g, "dx")
    with pytest.raises(RuntimeError, match="needs the fp32 activation"):
        ops.linear_bwd_data(dy, params[2 * (L - 1)].detach(), None, ops.ACT_RELU, torch.empty((B, ln[-2]), device=dev()))


@pytest.mark.parametrize("M,N,K0", [(5000, 512, 13), (300, 64, 479), (4096, 1024, 479), (33, 8, 5), (8192, 128, 14)])
def test_linear_weight_gradient_of_padded_input(M, N, K0):
    """First MLP layers run on a zero-padded input (13 -> 16 dense features, 479 -> 480 interaction outputs):
    ops.pad_cols builds the operand, dlrm_linear_bwd_weight_padded writes the gradient at the parameter's true width."""
    from dlrm_amd import ops
    rng = np.random.default_rng(M + N + K0)
    Kp = (K0 + 3) & ~3
    X = rng.standard_normal((M, K0)).astype(np.float32)
    dZ = rng.standard_normal((M, N)).astype(np.float32)
    wide = torch.full((M, K0 + 2), 9.0, device=dev())
    wide[:, 1:1 + K0] = to_dev(X)
    Xp = ops.pad_cols(wide[:, 1:1 + K0], Kp)                         # strided source
    assert Xp.shape == (M, Kp) and Xp.is_contiguous()
    assert torch.equal(Xp[:, :K0], to_dev(X)) and bool((Xp[:, K0:] == 0).all())
    dW = torch.full((N, K0), 5.0, device=dev())
    db = torch.full((N,), 5.0, device=dev())
    ops.linear_bwd_weight(to_dev(dZ), Xp, dW, db)
    want = dZ.astype(np.float64).T @ X.astype(np.float64)
    tol = 1e-5 * max(1.0, float(np.abs(want).max())) * 10
    np.testing.assert_allclose(dW.cpu().numpy(), want, rtol=1e-4, atol=tol)
    np.testing.assert_allclose(db.cpu().numpy(), dZ.astype(np.float64).sum(0), rtol=1e-4, atol=tol)
    ops.linear_bwd_weight(to_dev(dZ), Xp, dW, db, accumulate=True)
    np.testing.assert_allclose(dW.cpu().numpy(), 2 * want, rtol=1e-4, atol=2 * tol)
    with pytest.raises(RuntimeError):                                # the narrow form needs the slab workspace
        ops.linear_bwd_weight(to_dev(dZ), Xp, dW, db, use_workspace=False)


@pytest.mark.parametrize("M,K,act", [(5000, 256, 2), (333, 64, 1), (70001, 1024, 0), (129, 12, 2), (64, 100, 1)])
def test_linear_single_output_layer(M, K, act):
    """N == 1 (the last top-MLP layer): the matrix-vector kernels (gemv.hip) for K % 4 == 0 <= 1024, the GEMM path
    otherwise — forward, masked data gradient, weight + bias gradient with and without accumulation."""
    from dlrm_amd import ops
    arith = "f32"
    rng = np.random.default_rng(M + K)
    X = rng.standard_normal((M, K)).astype(np.float32)
    W = (rng.standard_normal((1, K)) / np.sqrt(K)).astype(np.float32)
    b = rng.standard_normal(1).astype(np.float32)
    Xd, Wd, bd = to_dev(X), to_dev(W), to_dev(b)
    Yd = torch.empty((M, 1), device=dev())
    ops.linear_fwd(Xd, Wd, bd, act, Yd, arith)
    np.testing.assert_allclose(Yd.cpu().numpy(), O.linear_fwd(X, W, b, act), rtol=1e-5, atol=1e-5)
    dZ = rng.standard_normal((M, 1)).astype(np.float32)
    mask = np.maximum(rng.standard_normal((M, K)), 0).astype(np.float32)
    dXd = torch.empty((M, K), device=dev())
    ops.linear_bwd_data(to_dev(dZ), Wd, to_dev(mask), 1, dXd)
    np.testing.assert_allclose(dXd.cpu().numpy(), (dZ.astype(np.float64) @ W.astype(np.float64)) * (mask > 0), rtol=1e-5, atol=1e-6)
    ops.linear_bwd_data(to_dev(dZ), Wd, None, 0, dXd)
    np.testing.assert_allclose(dXd.cpu().numpy(), dZ.astype(np.float64) @ W.astype(np.float64), rtol=1e-5, atol=1e-6)
    want_dW = dZ.astype(np.float64).T @ X.astype(np.float64)
    want_db = dZ.astype(np.float64).sum(0)
    tol = 1e-5 * max(1.0, float(np.abs(want_dW).max())) * 10
    dWd, dbd = torch.full((1, K), 3.0, device=dev()), torch.full((1,), 3.0, device=dev())
    ops.linear_bwd_weight(to_dev(dZ), Xd, dWd, dbd)                       # overwrite
    np.testing.assert_allclose(dWd.cpu().numpy(), want_dW, rtol=1e-4, atol=tol)
    np.testing.assert_allclose(dbd.cpu().numpy(), want_db, rtol=1e-4, atol=tol)
    first = dWd.clone()
    ops.linear_bwd_weight(to_dev(dZ), Xd, dWd, dbd, accumulate=True)      # accumulate on top
    np.testing.assert_allclose(dWd.cpu().numpy(), 2 * want_dW, rtol=1e-4, atol=2 * tol)
    np.testing.assert_allclose(dbd.cpu().numpy(), 2 * want_db, rtol=1e-4, atol=2 * tol)
    again = torch.empty((1, K), device=dev())
    ops.linear_bwd_weight(to_dev(dZ), Xd, again, None)
    assert torch.equal(again, first)                                     # fixed-order partial sums: deterministic


@pytest.mark.parametrize("M,N,K,act", [(5000, 512, 16, 1), (4096, 64, 8, 1), (70001, 16, 4, 0), (4100, 1024, 12, 2)])
def test_linear_small_reduction_layer(M, N, K, act):
    """K <= 16 and M >= 4096 (the first bottom-MLP layer at training batch sizes): the weight/bias gradient runs the
    streaming kernel of smallk.hip (forward stays on the GEMM kernel) — both against the oracle / a float64 restatement,
    accumulate mode, run-to-run bit identity (fixed-order partial sums), strided destination."""
    from dlrm_amd import ops
    arith = "f32"
    rng = np.random.default_rng(M + N + K)
    X = rng.standard_normal((M, K)).astype(np.float32)
    W = (rng.standard_normal((N, K)) / np.sqrt(K)).astype(np.float32)
    b = rng.standard_normal(N).astype(np.float32)
    Xd, Wd, bd = to_dev(X), to_dev(W), to_dev(b)
    Yd = torch.empty((M, N), device=dev())
    ops.linear_fwd(Xd, Wd, bd, act, Yd, arith)
    np.testing.assert_allclose(Yd.cpu().numpy(), O.linear_fwd(X, W, b, act), rtol=1e-5, atol=1e-5)
    dZ = rng.standard_normal((M, N)).astype(np.float32)
    want_dW = dZ.astype(np.float64).T @ X.astype(np.float64)
    want_db = dZ.astype(np.float64).sum(0)
    tol = 1e-5 * max(1.0, float(np.abs(want_dW).max())) * 10
    dZd = to_dev(dZ)
    dWd, dbd = torch.full((N, K), 3.0, device=dev()), torch.full((N,), 3.0, device=dev())
    ops.linear_bwd_weight(dZd, Xd, dWd, dbd, arith=arith)
    np.testing.assert_allclose(dWd.cpu().numpy(), want_dW, rtol=1e-4, atol=tol)
    np.testing.assert_allclose(dbd.cpu().numpy(), want_db, rtol=1e-4, atol=tol)
    first = dWd.clone()
    ops.linear_bwd_weight(dZd, Xd, dWd, dbd, accumulate=True)
    np.testing.assert_allclose(dWd.cpu().numpy(), 2 * want_dW, rtol=1e-4, atol=2 * tol)
    np.testing.assert_allclose(dbd.cpu().numpy(), 2 * want_db, rtol=1e-4, atol=2 * tol)
    again = torch.empty((N, K), device=dev())
    ops.linear_bwd_weight(dZd, Xd, again, None)
    assert torch.equal(again, first)
    # strided destination (a slot of a wider buffer, like the [B, (1+T)*D] feature buffer)
    wide = torch.full((M, N + 8), -1.0, device=dev())
    ops.linear_fwd(Xd, Wd, bd, act, wide[:, 4:4 + N])
    assert torch.equal(wide[:, 4:4 + N], Yd) and torch.all(wide[:, :4] == -1.0) and torch.all(wide[:, 4 + N:] == -1.0)


# ------------------------------------------------------------------------------------------ loss / SGD
@pytest.mark.parametrize("B", [1, 7, 128, 65536])
def test_bce_and_mse(B):
    from dlrm_amd import ops
    rng = np.random.default_rng(B)
    p = rng.random(B).astype(np.float32)
    p[: min(B, 3)] = [0.0, 1.0, 1e-30][: min(B, 3)]          # clamp paths (log -> -100, denominator -> 1e-12)
    t = np.round(rng.random(B)).astype(np.float32)
    want, dwant = O.bce(p, t)
    loss, dp = ops.bce_loss(to_dev(p), to_dev(t), None, 1.0, True)
    torch.cuda.synchronize()
    assert abs(float(loss.cpu()) - want) <= 1e-5 * max(abs(want), 1e-6)
    np.testing.assert_allclose(dp.cpu().numpy(), dwant, rtol=1e-5, atol=1e-12)
    want, dwant = O.mse(p, t)
    loss, dp = ops.mse_loss(to_dev(p), to_dev(t), 1.0, True)
    assert abs(float(loss.cpu()) - want) <= 1e-5 * max(abs(want), 1e-6)
    np.testing.assert_allclose(dp.cpu().numpy(), dwant, rtol=1e-6, atol=1e-12)


def test_sgd_dense():
    from dlrm_amd import ops
    rng = np.random.default_rng(9)
    for n in (1, 5, 1024, 2368897):
        w = rng.standard_normal(n).astype(np.float32)
        g = rng.standard_normal(n).astype(np.float32)
        wd = to_dev(w)
        ops.sgd_dense(wd, to_dev(g), 0.1)
        want = (w.astype(np.float64) - 0.1 * g.astype(np.float64))
        np.testing.assert_allclose(wd.cpu().numpy(), want, rtol=1e-6, atol=1e-7)


def test_copy_blocks_is_cat_and_split():
    """dlrm_copy_blocks: torch.cat / split along dim 1 over strided views, e.g. the all-to-all receive blocks
    [b_local, T_s*D] of every source rank -> one [b_local, T*D] matrix (the "cat" interaction in distributed mode)."""
    from dlrm_amd import ops
    rng = np.random.default_rng(3)
    for widths, M in (([8, 4, 12], 5), ([3, 5, 1], 33), ([128] * 5, 1000)):
        chunks = [rng.standard_normal((M, w)).astype(np.float32) for w in widths]
        srcs = [to_dev(c) for c in chunks]
        out = torch.full((M, sum(widths) + 4), -7.0, device=dev())
        dsts, o = [], 0
        for w in widths:
            dsts.append(out[:, o:o + w]); o += w
        ops.copy_blocks(srcs, dsts)
        assert np.array_equal(out[:, :o].cpu().numpy(), np.concatenate(chunks, axis=1)) and bool(torch.all(out[:, o:] == -7.0))
        back = [torch.empty((M, w), device=dev()) for w in widths]
        ops.copy_blocks(dsts, back)                      # the split direction
        for bk, c in zip(back, chunks):
            assert np.array_equal(bk.cpu().numpy(), c)


@pytest.mark.parametrize("idx_dtype", [torch.int64, torch.int32])
def test_out_of_range_index_is_skipped_and_reported(idx_dtype):
    """An index outside [0, rows) — the reference's EmbeddingBag raises on it — must never be dereferenced: forward
    treats the lookup as absent, every update mode leaves all tables untouched by it (the sorted path must not let it spill
    into another table's key bits), and the host raises IndexError from the pinned error block."""
    from dlrm_amd import ops
    rng = np.random.default_rng(9)
    D, rows, B = 16, [5, 64, 7], 40
    Ws = [rng.standard_normal((n, D)).astype(np.float32) for n in rows]
    idx = [rng.integers(0, n, size=B) for n in rows]
    bad = [i.copy() for i in idx]
    bad[0][3] = 5                      # == rows
    bad[1][17] = 1 << 20               # far beyond 2^row_bits
    bad[2][0] = -1 if idx_dtype == torch.int64 else 2 ** 31 - 1
    off = [np.arange(B)] * 3
    dout = rng.standard_normal((B, 3 * D)).astype(np.float32)

    def run(indices, fn):
        Wd = [to_dev(w.copy()) for w in Ws]
        bags = ops.BagBatch([to_dev(o).to(idx_dtype) for o in off], [to_dev(i).to(idx_dtype) for i in indices])
        fn(Wd, bags)
        torch.cuda.synchronize()
        return Wd

    ops.check_index_errors(sync=True)   # clean slate
    # forward: bad lookups pool to zero, the others are unaffected
    out = torch.empty((B, 3 * D), device=dev())
    run(bad, lambda Wd, bags: ops.emb_fwd(Wd, bags, out))
    with pytest.raises(IndexError, match="out of range"):
        ops.check_index_errors(sync=True)
    ops.check_index_errors(sync=True)   # the block was reset by the raise
    want = np.concatenate([Ws[t][idx[t]] for t in range(3)], axis=1)
    want[3, 0:D] = 0; want[17, D:2 * D] = 0; want[0, 2 * D:3 * D] = 0
    assert np.array_equal(out.cpu().numpy(), want)
    # updates: identical to the update with the bad lookups REMOVED (their gradient rows zeroed), in every mode
    dz = dout.copy()
    dz[3, 0:D] = 0; dz[17, D:2 * D] = 0; dz[0, 2 * D:3 * D] = 0
    for mode in (ops.UPD_SORTED, ops.UPD_ATOMIC, ops.UPD_DETERMINISTIC):
        got = run(bad, lambda Wd, bags: ops.emb_bwd_sgd(Wd, bags, to_dev(dout), 0.5, mode))
        with pytest.raises(IndexError):
            ops.check_index_errors(sync=True)
        ref = run(idx, lambda Wd, bags: ops.emb_bwd_sgd(Wd, bags, to_dev(dz), 0.5, mode))
        ops.check_index_errors(sync=True)
        for g, r in zip(got, ref):
            np.testing.assert_allclose(g.cpu().numpy(), r.cpu().numpy(), rtol=1e-6, atol=1e-6)
    st = [torch.zeros(n, device=dev()) for n in rows]
    got = run(bad, lambda Wd, bags: ops.emb_bwd_rowwise_adagrad(Wd, st, bags, to_dev(dout), 0.1, 1e-8))
    with pytest.raises(IndexError):
        ops.check_index_errors(sync=True)
    st2 = [torch.zeros(n, device=dev()) for n in rows]
    ref = run(idx, lambda Wd, bags: ops.emb_bwd_rowwise_adagrad(Wd, st2, bags, to_dev(dz), 0.1, 1e-8))
    for g, r in zip(got + st, ref + st2):
        np.testing.assert_allclose(g.cpu().numpy(), r.cpu().numpy(), rtol=1e-5, atol=1e-6)


def test_emb_bwd_coo_is_the_reference_sparse_gradient():
    """dlrm_emb_bwd_coo: values[i] = psw[i] * dout[bag(i)] per lookup in input order (EmbeddingBagBackward's COO values,
    dlrm_s_pytorch.py:1613) — ragged, empty and weighted bags — and, through torch.sparse_coo_tensor + torch.optim.SGD,
    the bit-exact reference update."""
    from dlrm_amd import ops
    rng = np.random.default_rng(21)
    D, rows, B = 12, [9, 300], 50
    lens = [rng.integers(0, 5, size=B) for _ in rows]
    off = [np.concatenate([[0], np.cumsum(l)[:-1]]) for l in lens]
    idx = [rng.integers(0, n, size=int(l.sum())) for n, l in zip(rows, lens)]
    psw = [rng.standard_normal(i.size).astype(np.float32) for i in idx]
    dout = rng.standard_normal((B, 2 * D)).astype(np.float32)
    for weights in (None, psw):
        bags = ops.BagBatch([to_dev(o) for o in off], [to_dev(i) for i in idx], None if weights is None else [to_dev(w) for w in weights])
        vals = ops.emb_bwd_coo(bags, to_dev(dout), D)
        for t in range(2):
            bag_of = np.repeat(np.arange(B), lens[t])
            want = dout[bag_of, t * D:(t + 1) * D]
            if weights is not None:
                want = want * weights[t][:, None]
            assert np.array_equal(vals[t].cpu().numpy(), want.astype(np.float32)), t


@pytest.mark.parametrize("T,B,idx_dtype,itself", [(26, 5000, torch.int64, False), (26, 4099, torch.int32, False), (3, 777, torch.int64, True),
                                                  (26, 300, torch.int64, True), (7, 64, torch.int32, True), (2, 5, torch.int64, False),
                                                  (26, 65536, torch.int64, False)])
def test_gather_interaction_is_bit_identical_to_the_two_kernels(T, B, idx_dtype, itself):
    """dlrm_interact_fwd_gather / _bwd_gather (one lookup per bag, D = 128: the interaction kernel fetches the embedding rows
    itself) against dlrm_emb_fwd + dlrm_interact_fwd / _bwd: the SAME bits for R, dx and the embedding-row gradients; a
    non-one-hot bag layout and an out-of-range index are reported through the error block."""
    from dlrm_amd import ops
    rng = np.random.default_rng(T * B)
    D = 128
    rows = [int(r) for r in rng.integers(1, 5000, size=T)]
    Ws = [to_dev(rng.standard_normal((n, D)).astype(np.float32)) for n in rows]
    idx = torch.stack([to_dev(rng.integers(0, n, size=B)) for n in rows]).to(idx_dtype)
    off = torch.arange(B, device=dev()).repeat(T, 1).to(idx_dtype)
    x = to_dev(rng.standard_normal((B, D)).astype(np.float32))
    bags = ops.BagBatch(off, idx)
    F = T + 1
    W_ = ops.interact_out_width(F, D, itself)
    ldr = (W_ + 3) & ~3
    # reference route: pooled embeddings into a feature buffer, then the plain interaction kernels
    feat = torch.empty((B, F * D), device=dev())
    feat[:, :D] = x
    ops.emb_fwd(Ws, bags, feat[:, D:])
    R0 = torch.empty((B, ldr), device=dev())
    ops.interact_fwd([feat[:, :D], feat[:, D:]], D, itself, R0)
    R1 = torch.full((B, ldr), 7.0, device=dev())
    ops.interact_fwd_gather(x, Ws, bags, D, itself, R1)
    assert torch.equal(R0, R1)
    # ... and DIRECTLY against the oracle (O.emb_fwd + O.interact_fwd, dlrm_s_pytorch.py:407-462, 483-504): this test stands on its
    # own, whatever else ran before it
    idx_np = idx.cpu().numpy().astype(np.int64)
    Bo = min(B, 6000)
    feat_o = np.empty((Bo, F, D), dtype=np.float32)
    feat_o[:, 0] = x.cpu().numpy()[:Bo]
    for t in range(T):
        feat_o[:, 1 + t] = O.emb_fwd(Ws[t].cpu().numpy(), idx_np[t, :Bo], np.arange(Bo, dtype=np.int64))
    want = O.interact_fwd(feat_o, itself)
    np.testing.assert_allclose(R1.cpu().numpy()[:Bo, :W_], want, rtol=1e-5, atol=3e-5)
    dR = to_dev(rng.standard_normal((B, ldr)).astype(np.float32))
    d0 = torch.empty((B, F * D), device=dev())
    ops.interact_bwd([feat[:, :D], feat[:, D:]], D, itself, dR, [d0[:, :D], d0[:, D:]])
    dx, dE = torch.empty((B, D), device=dev()), torch.empty((B, T * D), device=dev())
    ops.interact_bwd_gather(x, Ws, bags, D, itself, dR, dx, dE)
    assert torch.equal(d0[:, :D], dx) and torch.equal(d0[:, D:], dE)
    ops.check_index_errors(sync=True)
    dwant = O.interact_bwd(feat_o, dR.cpu().numpy()[:Bo, :W_], itself)
    np.testing.assert_allclose(dx.cpu().numpy()[:Bo], dwant[:, 0, :], rtol=1e-5, atol=6e-5)
    np.testing.assert_allclose(dE.cpu().numpy()[:Bo].reshape(Bo, T, D), dwant[:, 1:, :], rtol=1e-5, atol=6e-5)
    # violations are reported
    bad_off = off.clone(); bad_off[T - 1, min(5, B - 1)] = min(5, B - 1) - 1
    ops.interact_fwd_gather(x, Ws, ops.BagBatch(bad_off, idx), D, itself, R1)
    with pytest.raises(IndexError, match="one-lookup-per-bag"):
        ops.check_index_errors(sync=True)
    bad_idx = idx.clone(); bad_idx[0, 3] = rows[0]
    ops.interact_fwd_gather(x, Ws, ops.BagBatch(off, bad_idx), D, itself, R1)
    with pytest.raises(IndexError, match="out of range"):
        ops.check_index_errors(sync=True)
    if idx_dtype == torch.int64:                       # the selectors travel as two dwords: the high one counts
        bad_idx = idx.clone(); bad_idx[T - 1, B - 1] = (1 << 32) + 1
        ops.interact_bwd_gather(x, Ws, ops.BagBatch(off, bad_idx), D, itself, dR, dx, dE)
        with pytest.raises(IndexError, match="out of range"):
            ops.check_index_errors(sync=True)
    assert not ops.gather_ok(29, 128) and ops.gather_ok(27, 128)      # the dR row of F > 27 (with self pairs) exceeds the gather backward's image


@pytest.mark.parametrize("T,B,idx_dtype,itself,lr_on_device,relu_x", [
    (26, 5000, torch.int64, False, False, True), (26, 4099, torch.int32, False, True, False), (3, 777, torch.int64, True, False, False),
    (7, 64, torch.int32, True, True, True), (2, 5, torch.int64, False, False, False), (1, 1, torch.int64, False, False, False),
    (26, 65536, torch.int64, False, True, True)])
def test_single_lookup_rows_are_updated_inside_the_fused_backward(T, B, idx_dtype, itself, lr_on_device, relu_x):
    """ABI 17 (dlrm_emb_presort + dlrm_interact_bwd_gather_sgd + dlrm_emb_bwd_sgd_presorted) against dlrm_interact_bwd_gather +
    dlrm_emb_bwd_sgd(DLRM_UPD_SORTED) — EmbeddingBagBackward + SGD.step of the reference (dlrm_s_pytorch.py:1613,1620): the SAME bits in every
    table and in dx; the mask of single lookups against numpy's own count of each (table, row); the gradient rows of the single lookups are
    not written, all others are the plain backward's; under a launch predicate that does NOT hold (a ragged batch) nothing is updated by the
    backward call and the presorted update applies every lookup; tables mixing rows looked up once, often and never."""
    from dlrm_amd import ops
    rng = np.random.default_rng(T * B + 17)
    D = 128
    # rows per table from "almost every lookup single" down to "every row hot"
    rows = [max(1, int(B * f)) for f in rng.choice([40.0, 8.0, 1.5, 0.3, 0.02] if B < 60000 else [6.0, 2.0, 1.0, 0.3, 0.02], size=T)]
    rows[0] = max(rows[0], 3)
    gen = torch.Generator(device=dev()); gen.manual_seed(T * B + 18)
    W0 = [torch.randn((n, D), device=dev(), generator=gen) for n in rows]
    idx = torch.stack([to_dev(rng.integers(0, n, size=B)) for n in rows]).to(idx_dtype)
    off = torch.arange(B, device=dev()).repeat(T, 1).to(idx_dtype)
    x = to_dev(rng.standard_normal((B, D)).astype(np.float32))
    if relu_x:
        x = torch.relu(x)
    bags = ops.BagBatch(off, idx)
    mode = int(itself) | (ops.INTERACT_RELU_X if relu_x else 0)
    ldr = (ops.interact_out_width(T + 1, D, itself) + 3) & ~3
    dR = to_dev(rng.standard_normal((B, ldr)).astype(np.float32))
    lr = 0.37
    lr_arg = torch.full((1,), lr, device=dev()) if lr_on_device else lr

    # the two calls of rounds 3-6
    Wa = [w.clone() for w in W0]
    dx_a, dE_a = torch.empty((B, D), device=dev()), torch.empty((B, T * D), device=dev())
    ops.interact_bwd_gather(x, Wa, bags, D, mode, dR, dx_a, dE_a)
    ops.emb_bwd_sgd(Wa, bags, dE_a, lr_arg, ops.UPD_SORTED)

    # presort -> backward that takes the single rows -> the rest from the sorted workspace
    Wb = [w.clone() for w in W0]
    assert ops.presort_ok(Wb, bags)
    pre = ops.emb_presort(Wb, bags, lr_arg)
    idx_np = idx.cpu().numpy().astype(np.int64)
    want_mask = np.zeros(B, dtype=np.uint32)
    for t in range(T):
        cnt = np.bincount(idx_np[t], minlength=rows[t])
        want_mask |= (cnt[idx_np[t]] == 1).astype(np.uint32) << np.uint32(t)
    got_mask = pre.mask.cpu().numpy().view(np.uint32)
    assert np.array_equal(got_mask, want_mask)
    once_or_never = [torch.from_numpy(np.bincount(idx_np[t], minlength=rows[t]) <= 1).to(dev()) for t in range(T)]

    def same_tables(Wx):
        # rows looked up at most once: the same bits.  Rows looked up several times: the sorted update adds the partial sums of a run that
        # crosses a 64-entry group boundary with fp32 atomics (DLRM_UPD_SORTED's contract: hot rows are re-associated), in either path
        for t in range(T):
            assert torch.equal(Wa[t][once_or_never[t]], Wx[t][once_or_never[t]]), t
            assert torch.allclose(Wa[t], Wx[t], rtol=1e-5, atol=1e-5), t
    dx_b = torch.empty((B, D), device=dev())
    dE_b = torch.full((B, T * D), float("nan"), device=dev())
    ops.interact_bwd_gather(x, Wb, bags, D, mode, dR, dx_b, dE_b, presorted=pre)
    single = torch.from_numpy(((want_mask[:, None] >> np.arange(T, dtype=np.uint32)[None, :]) & 1).astype(bool)).to(dev())     # [B, T]
    dEb3, dEa3 = dE_b.view(B, T, D), dE_a.view(B, T, D)
    assert torch.isnan(dEb3[single]).all()                               # gradient rows of single lookups: never written
    assert torch.equal(dEb3[~single], dEa3[~single])
    assert torch.equal(dx_a, dx_b)
    # single rows already hold their update, every other row is still untouched
    for t in range(T):
        rows_single = idx[t][single[:, t]].long()
        assert torch.equal(Wb[t][rows_single], Wa[t][rows_single]), t
        keep = torch.ones(rows[t], dtype=torch.bool, device=dev()); keep[rows_single] = False
        assert torch.equal(Wb[t][keep], W0[t][keep]), t
    ops.emb_bwd_sgd_presorted(Wb, bags, dE_b, pre)
    same_tables(Wb)
    ops.check_index_errors(sync=True)

    # a launch predicate that does not hold: the fused backward returns at once (the two-kernel form wrote every gradient row) and the
    # presorted update applies EVERY lookup — with skip_singles asked for
    flag = torch.ones(1, dtype=torch.int32, device=dev())
    Wc = [w.clone() for w in W0]
    pre_c = ops.emb_presort(Wc, bags, lr_arg, pred=(flag, 0))
    dx_c, dE_c = torch.full((B, D), 3.0, device=dev()), torch.full((B, T * D), 5.0, device=dev())
    ops.interact_bwd_gather(x, Wc, bags, D, mode, dR, dx_c, dE_c, pred=(flag, 0), presorted=pre_c)
    assert bool((dx_c == 3.0).all()) and bool((dE_c == 5.0).all())
    for t in range(T):
        assert torch.equal(Wc[t], W0[t])
    ops.emb_bwd_sgd_presorted(Wc, bags, dE_a, pre_c)
    same_tables(Wc)
    # ... and one that holds, through the predicated entry
    flag.zero_()
    Wd = [w.clone() for w in W0]
    pre_d = ops.emb_presort(Wd, bags, lr_arg, pred=(flag, 0))
    dE_d = torch.empty((B, T * D), device=dev())
    ops.interact_bwd_gather(x, Wd, bags, D, mode, dR, dx_c, dE_d, pred=(flag, 0), presorted=pre_d)
    ops.emb_bwd_sgd_presorted(Wd, bags, dE_d, pre_d)
    same_tables(Wd)
    assert torch.equal(dx_a, dx_c)
    # skip_singles = False on a plain backward's gradient rows: the second half of the sorted update alone
    We = [w.clone() for w in W0]
    ops.emb_bwd_sgd_presorted(We, bags, dE_a, ops.emb_presort(We, bags, lr_arg), skip_singles=False)
    same_tables(We)
    # an out-of-range lookup is never a single one: it is skipped by both halves and reported
    if B >= 5:
        bad = idx.clone(); bad[0, 3] = rows[0]
        bb = ops.BagBatch(off, bad)
        Wf = [w.clone() for w in W0]
        pre_f = ops.emb_presort(Wf, bb, lr_arg)
        assert not (int(pre_f.mask[3].item()) & 1)
        with pytest.raises(IndexError, match="out of range"):
            ops.check_index_errors(sync=True)


@pytest.mark.parametrize("T,B,D,mode", [(26, 4099, 128, 0), (3, 777, 128, 1), (26, 1000, 128, 2), (26, 300, 16, 0), (8, 513, 64, 1), (2, 5, 36, 0)])
def test_interaction_backward_applies_the_relu_derivative_of_feature_0(T, B, D, mode):
    """`mode | INTERACT_RELU_X`: the backward kernels (LDS-DMA form at D = 128 with and without the fused lookups, the generic form
    elsewhere) multiply the gradient of feature 0 by [feature 0 > 0] — the same bits as dlrm_act_bwd(ReLU) applied afterwards, which
    is what the bottom tower's backward otherwise starts with (dlrm_s_pytorch.py:238-241, 483-504); every other gradient row unchanged."""
    from dlrm_amd import ops
    rng = np.random.default_rng(T * B + D)
    F = T + 1
    x = torch.relu(to_dev(rng.standard_normal((B, D)).astype(np.float32)))          # a ReLU output: about half zeros
    x[0, 0] = 0.0; x[B - 1, D - 1] = -0.0
    E = to_dev(rng.standard_normal((B, T * D)).astype(np.float32))
    W_ = ops.interact_out_width(F, D, mode)
    ldr = (W_ + 3) & ~3
    dR = to_dev(rng.standard_normal((B, ldr)).astype(np.float32))
    d0 = [torch.empty((B, D), device=dev()), torch.empty((B, T * D), device=dev())]
    d1 = [torch.full((B, D), 9.0, device=dev()), torch.full((B, T * D), 9.0, device=dev())]
    ops.interact_bwd([x, E], D, mode, dR, d0)
    ops.interact_bwd([x, E], D, mode | ops.INTERACT_RELU_X, dR, d1)
    want = torch.empty_like(d0[0])
    ops.act_bwd(d0[0], x, ops.ACT_RELU, want, None)
    assert torch.equal(d1[0], want) and torch.equal(d1[1], d0[1])
    assert float(want.abs().sum()) > 0 and not torch.equal(want, d0[0])
    if D == 128 and ops.gather_ok(F, D):
        rows = [int(r) for r in rng.integers(1, 3000, size=T)]
        Ws = [to_dev(rng.standard_normal((n, D)).astype(np.float32)) for n in rows]
        idx = torch.stack([to_dev(rng.integers(0, n, size=B)) for n in rows])
        bags = ops.BagBatch(torch.arange(B, device=dev()).repeat(T, 1), idx)
        dx0, dE0 = torch.empty((B, D), device=dev()), torch.empty((B, T * D), device=dev())
        dx1, dE1 = torch.empty((B, D), device=dev()), torch.empty((B, T * D), device=dev())
        ops.interact_bwd_gather(x, Ws, bags, D, mode, dR, dx0, dE0)
        ops.interact_bwd_gather(x, Ws, bags, D, mode | ops.INTERACT_RELU_X, dR, dx1, dE1)
        ops.act_bwd(dx0, x, ops.ACT_RELU, want, None)
        assert torch.equal(dx1, want) and torch.equal(dE1, dE0)
        ops.check_index_errors(sync=True)
    with pytest.raises(RuntimeError):
        ops.interact_bwd([x, E], D, 3 | ops.INTERACT_RELU_X, dR, d1)                 # mode 3 does not exist, with or without the flag


# ------------------------------------------------------------------------------------------ config 5 inputs: Multihot
@pytest.mark.parametrize("id_dtype", [torch.int64, torch.int32])
def test_multihot_expand_matches_reference_class(id_dtype):
    """dlrm_multihot_expand with the REFERENCE's lookup tables uploaded: values (int32, table-major KJT layout), cumulative
    offsets (int64) — bit-exact against the outputs of torchrec_dlrm/multi_hot.py's own class (fixture multihot_tables.npz),
    for the constructor's batch size and a different one; local offsets = b * h_t; pooled embeddings through the views
    handed to the model equal the oracle's EmbeddingBag of the reference's values."""
    from conftest import load_golden
    from dlrm_amd import ops
    from dlrm_amd.multihot import Multihot
    d, meta = load_golden("multihot_tables")
    for c in meta["cases"]:
        tabs = [d[f"{c['tag']}.table{k}"] for k in range(len(c["sizes"]))]
        mh = Multihot.from_host_tables(tabs, c["batches"][-1], device=dev())
        assert mh.lookups_per_sample == sum(c["sizes"])
        for b in c["batches"]:
            ids = to_dev(d[f"{c['tag']}.b{b}.ids"]).to(id_dtype)
            values, off_g, off_l = mh.expand(ids)
            torch.cuda.synchronize()
            assert values.dtype == torch.int32 and np.array_equal(values.cpu().numpy(), d[f"{c['tag']}.b{b}.values"])
            assert off_g.dtype == torch.int64 and np.array_equal(off_g.cpu().numpy(), d[f"{c['tag']}.b{b}.offsets"])
            for t, h in enumerate(c["sizes"]):
                assert np.array_equal(off_l[t].cpu().numpy(), np.arange(b) * h)
            # straight into the embedding kernel (int32 indices, h_t lookups per bag)
            D = 8
            rng = np.random.default_rng(1)
            Ws = [rng.standard_normal((n, D)).astype(np.float32) for n in c["n_emb"]]
            lS_o, lS_i = mh.to_model_inputs(ids)
            out = torch.empty((b, len(Ws) * D), device=dev())
            ops.emb_fwd([to_dev(w) for w in Ws], ops.BagBatch(lS_o, lS_i), out)
            ref_v, ref_o = d[f"{c['tag']}.b{b}.values"], d[f"{c['tag']}.b{b}.offsets"]
            for t in range(len(Ws)):
                lo, hi = ref_o[t * b], ref_o[(t + 1) * b]
                want = O.emb_fwd(Ws[t], ref_v[lo:hi].astype(np.int64), (ref_o[t * b:(t + 1) * b] - lo).astype(np.int64))
                assert np.array_equal(out[:, t * D:(t + 1) * D].cpu().numpy(), want), (c["tag"], b, t)
    ops.check_index_errors(sync=True)
    # an id outside the table is reported (F.embedding raises in the reference) and never dereferenced
    bad = to_dev(d["uniform.b32.ids"]).to(id_dtype).clone()
    bad[2, 5] = 3                                                     # table 2 has 3 rows
    c0 = meta["cases"][0]
    mh = Multihot.from_host_tables([d[f"uniform.table{k}"] for k in range(len(c0["sizes"]))], 32, device=dev())
    mh.expand(bad)
    with pytest.raises(IndexError, match="table 2"):
        ops.check_index_errors(sync=True)


def test_multihot_generated_tables_match_philox_oracle():
    """dlrm_multihot_gen_table == oracle.philox_multihot_table: column 0 the id, uniform columns bit-exact (integer
    multiply-shift of a Philox word), pareto columns equal up to the last-bit differences of exp/log1p between the device
    and numpy (the value is a huge heavy-tailed double truncated to int32) and both with the reference's distribution."""
    from dlrm_amd.multihot import Multihot
    sizes, n_emb = [3, 1, 12, 5], [1000, 7, 50000, 3]
    mh = Multihot(sizes, n_emb, 16, dist_type="uniform", device=dev(), seed=99)
    for t, (h, n) in enumerate(zip(sizes, n_emb)):
        got = mh.multi_hot_tables_l[t].cpu().numpy()
        assert got.dtype == np.int32 and np.array_equal(got, O.philox_multihot_table(t, n, h, 0, 99)), t
        assert got.min() >= 0 and got.max() < n
    big = mh.multi_hot_tables_l[2].cpu().numpy()[:, 1:].reshape(-1)
    assert abs(big.mean() / 50000 - 0.5) < 5e-3                       # uniform on [0, n)
    mp_ = Multihot(sizes, n_emb, 16, dist_type="pareto", device=dev(), seed=5)
    for t, (h, n) in enumerate(zip(sizes, n_emb)):
        got = mp_.multi_hot_tables_l[t].cpu().numpy()
        want = O.philox_multihot_table(t, n, h, 1, 5)
        assert got.min() >= 0 and got.max() < n and np.array_equal(got[:, 0], np.arange(n))
        assert np.mean(got != want) < 1e-4, (t, np.mean(got != want))
    with pytest.raises(ValueError):
        Multihot(sizes, n_emb, 16, dist_type="zipf", device=dev())


# ------------------------------------------------------------------------------------------ K4: row-wise Adagrad
@pytest.mark.parametrize("D,rows,B,max_len", [(16, [3, 40, 5000], 300, 5), (128, [4, 10, 130, 100000], 700, 3),
                                              (6, [2, 9], 97, 4), (64, [1, 3], 5000, 2), (200, [7, 300], 260, 6)])
@pytest.mark.parametrize("idx_dtype", [torch.int64, torch.int32])
def test_emb_bwd_rowwise_adagrad_matches_oracle(D, rows, B, max_len, idx_dtype):
    """Fused backward + row-wise sparse Adagrad vs the C oracle (pinned to optim/rwsadagrad.py by
    tests/test_oracle_golden.py).  Tables with 1-4 rows make runs of thousands of duplicates that cross many
    64-entry groups (edge buffers + fix-up pass); two consecutive steps exercise the accumulated state."""
    from dlrm_amd import ops
    rng = np.random.default_rng(1000 + D + B)
    Ws = [rng.standard_normal((n, D)).astype(np.float32) for n in rows]
    moms = [np.full(n, 0.01, dtype=np.float32) for n in rows]
    dW = [to_dev(W) for W in Ws]
    dM = [to_dev(m) for m in moms]
    for step in range(2):
        bags = [ragged(rng, B, n, max_len, empty_frac=0.1) for n in rows]
        psw = [None] * len(rows)
        psw[-1] = rng.standard_normal(bags[-1][1].shape[0]).astype(np.float32)
        dV = (rng.standard_normal((B, len(rows) * D)) * 0.1).astype(np.float32)
        clr, eps = 0.05 / (1 + step * 0.1), 1e-8
        for t, (W, m, (o, i), w) in enumerate(zip(Ws, moms, bags, psw)):
            O.emb_bwd_rowwise_adagrad(W, m, i, o, np.ascontiguousarray(dV[:, t * D:(t + 1) * D]), clr, eps, psw=w)
        bb = ops.BagBatch([to_dev(o, idx_dtype) for o, _ in bags], [to_dev(i, idx_dtype) for _, i in bags],
                          [None if w is None else to_dev(w) for w in psw])
        ops.emb_bwd_rowwise_adagrad(dW, dM, bb, to_dev(dV), clr, eps)
        torch.cuda.synchronize()
        for t in range(len(rows)):
            # hot rows sum thousands of duplicates: fp32 re-association (per 64 lookups) vs the sequential oracle
            np.testing.assert_allclose(dM[t].cpu().numpy(), moms[t], rtol=2e-4, atol=1e-7, err_msg=f"mom {t} step {step}")
            np.testing.assert_allclose(dW[t].cpu().numpy(), Ws[t], rtol=2e-4, atol=2e-5, err_msg=f"W {t} step {step}")


def test_emb_bwd_rowwise_adagrad_is_deterministic_and_touches_only_looked_up_rows():
    from dlrm_amd import ops
    rng = np.random.default_rng(5)
    D, rows, B = 32, [5, 2000], 4000
    bags = [ragged(rng, B, n, 3) for n in rows]
    dV = to_dev((rng.standard_normal((B, len(rows) * D)) * 0.1).astype(np.float32))
    W0 = [rng.standard_normal((n, D)).astype(np.float32) for n in rows]
    bb = ops.BagBatch([to_dev(o) for o, _ in bags], [to_dev(i) for _, i in bags])
    results = []
    for _ in range(2):
        dW = [to_dev(W) for W in W0]
        dM = [torch.zeros(n, device=dev()) for n in rows]
        ops.emb_bwd_rowwise_adagrad(dW, dM, bb, dV, 0.1, 1e-8)
        torch.cuda.synchronize()
        results.append(([w.cpu().numpy() for w in dW], [m.cpu().numpy() for m in dM]))
    for a, b in zip(results[0][0] + results[0][1], results[1][0] + results[1][1]):
        assert np.array_equal(a, b)                       # fixed summation order: bit-identical run to run
    touched = np.zeros(rows[1], dtype=bool)
    touched[bags[1][1]] = True
    assert np.array_equal(results[0][0][1][~touched], W0[1][~touched])
    assert np.all(results[0][1][1][~touched] == 0)


def test_dense_optimizer_kernels():
    from dlrm_amd import ops
    rng = np.random.default_rng(9)
    sizes = [1, 3, 4, 13, 4096, 4097, 512 * 13, 1024 * 1024 + 5]
    ws = [rng.standard_normal(n).astype(np.float32) for n in sizes]
    gs = [rng.standard_normal(n).astype(np.float32) for n in sizes]
    big = torch.empty(sum(sizes) + 8 * len(sizes) + 1, device=dev())     # odd offsets: unaligned tensors too
    dws, o = [], 1
    for w in ws:
        v = big[o:o + w.size]; v.copy_(torch.from_numpy(w)); dws.append(v); o += w.size + 3
    dgs = [to_dev(g) for g in gs]
    ops.sgd_dense_multi(dws, dgs, 0.37)
    torch.cuda.synchronize()
    for w, g, dw in zip(ws, gs, dws):
        want = (w.astype(np.float64) - 0.37 * g.astype(np.float64))
        np.testing.assert_allclose(dw.cpu().numpy(), want, rtol=1e-6, atol=1e-6)
    # dense Adagrad (optim/rwsadagrad.py:145-148)
    w, g, sm = ws[-1].copy(), gs[-1], np.abs(rng.standard_normal(sizes[-1])).astype(np.float32)
    dw, dsum = to_dev(w), to_dev(sm)
    ops.adagrad_dense(dw, dsum, to_dev(g), 0.05, 1e-10)
    torch.cuda.synchronize()
    s2 = sm.astype(np.float64) + g.astype(np.float64) ** 2
    np.testing.assert_allclose(dsum.cpu().numpy(), s2, rtol=1e-6)
    np.testing.assert_allclose(dw.cpu().numpy(), w - 0.05 * g / (np.sqrt(s2) + 1e-10), rtol=1e-5, atol=1e-6)


def test_learning_rate_from_a_device_scalar_is_bit_identical_to_by_value():
    """include/dlrm_hip.h "LEARNING RATES" (ABI 15): every update entry point takes its step size by value OR reads it from a device float when
    the kernel runs (what a captured HIP graph needs to follow the reference's per-iteration LRPolicyScheduler, dlrm_s_pytorch.py:169-203).
    Both forms multiply with the same fp32 value: deterministic / row-wise-Adagrad / dense kernels must agree BIT FOR BIT, the sorted and
    atomic embedding updates up to the order of their atomic adds; a later write to the scalar (dlrm_set_f32) is what the next launch sees."""
    from dlrm_amd import ops
    rng = np.random.default_rng(17)
    D, rows, B = 32, [7, 3000, 90], 2500
    bags = [ragged(rng, B, n, 3) for n in rows]
    dV = to_dev((rng.standard_normal((B, len(rows) * D)) * 0.1).astype(np.float32))
    W0 = [rng.standard_normal((n, D)).astype(np.float32) for n in rows]
    bb = ops.BagBatch([to_dev(o) for o, _ in bags], [to_dev(i) for _, i in bags])
    lr = 0.3125 + 1e-3                                           # not exactly representable: float(lr) is what both forms must use
    lr_t = torch.zeros(1, device=dev())
    ops.set_f32([lr_t], [lr])
    assert float(lr_t) == float(np.float32(lr))
    for mode, exact in ((ops.UPD_DETERMINISTIC, True), (ops.UPD_SORTED, False), (ops.UPD_ATOMIC, False)):
        got = []
        for step in (lr, lr_t):
            dW = [to_dev(W) for W in W0]
            ops.emb_bwd_sgd(dW, bb, dV, step, mode)
            torch.cuda.synchronize()
            got.append([w.cpu().numpy() for w in dW])
        for a, b in zip(*got):
            if exact:
                assert np.array_equal(a, b)
            else:
                np.testing.assert_allclose(a, b, rtol=1e-5, atol=1e-6)
    got = []
    for step in (lr, lr_t):
        dW = [to_dev(W) for W in W0]
        dM = [torch.zeros(n, device=dev()) for n in rows]
        ops.emb_bwd_rowwise_adagrad(dW, dM, bb, dV, step, 1e-8)
        torch.cuda.synchronize()
        got.append([w.cpu().numpy() for w in dW] + [m.cpu().numpy() for m in dM])
    for a, b in zip(*got):
        assert np.array_equal(a, b)
    n = 4097
    w0, g0, s0 = rng.standard_normal(n).astype(np.float32), rng.standard_normal(n).astype(np.float32), np.abs(rng.standard_normal(n)).astype(np.float32)
    res = []
    for step in (lr, lr_t):
        w1, w2, w3, sm = to_dev(w0), to_dev(w0), to_dev(w0), to_dev(s0)
        ops.sgd_dense(w1, to_dev(g0), step)
        ops.sgd_dense_multi([w2], [to_dev(g0)], step)
        ops.adagrad_dense(w3, sm, to_dev(g0), step, 1e-10)
        torch.cuda.synchronize()
        res.append([t.cpu().numpy() for t in (w1, w2, w3, sm)])
    for a, b in zip(*res):
        assert np.array_equal(a, b)
    # the scalar is read when the kernel RUNS: rewrite it, launch again with the same tensor
    ops.set_f32([lr_t], [0.0])
    w = to_dev(w0)
    ops.sgd_dense(w, to_dev(g0), lr_t)
    torch.cuda.synchronize()
    assert np.array_equal(w.cpu().numpy(), w0)
    with pytest.raises(RuntimeError):
        ops.sgd_dense(w, to_dev(g0), torch.zeros(1, dtype=torch.float64, device=dev()))


# ------------------------------------------------------------------------------------------ evaluation metrics
def test_binary_metrics_match_scikit_learn_golden():
    from conftest import load_golden
    from dlrm_amd import ops
    d, meta = load_golden("metrics_sklearn")
    for case in meta["cases"]:
        s, t = d[case["tag"] + ".scores"], d[case["tag"] + ".targets"]
        m = ops.binary_metrics(to_dev(s), to_dev(t))
        for k in ("recall", "precision", "f1", "ap", "roc_auc", "accuracy"):
            assert abs(m[k] - case[k]) <= 1e-9, (case["tag"], k, m[k], case[k])
        assert m["round_matches"] == case["round_matches"]
        o = O.binary_metrics(s, t)
        assert (m["tp"], m["fp"], m["fn"], m["tn"]) == (o["tp"], o["fp"], o["fn"], o["tn"])


def test_binary_metrics_large_and_degenerate():
    from dlrm_amd import ops
    rng = np.random.default_rng(3)
    n = 3_000_000
    t = (rng.random(n) < 0.25).astype(np.float32)
    s = (1 / (1 + np.exp(-(1.5 * (t - 0.3) + rng.standard_normal(n))))).astype(np.float32)
    s[::7] = np.round(s[::7], 2)                       # long tie groups
    m, o = ops.binary_metrics(to_dev(s), to_dev(t)), O.binary_metrics(s, t)
    assert abs(m["roc_auc"] - o["roc_auc"]) < 1e-10 and abs(m["ap"] - o["ap"]) < 1e-10
    assert m["round_matches"] == o["round_matches"] and m["positives"] == o["positives"]
    one_class = ops.binary_metrics(to_dev(s[:100]), torch.zeros(100, device=dev()))
    assert np.isnan(one_class["roc_auc"]) and np.isnan(one_class["ap"]) and one_class["recall"] == 0.0
    const = ops.binary_metrics(torch.full((1000,), 0.5, device=dev()), to_dev(t[:1000]))
    assert abs(const["roc_auc"] - 0.5) < 1e-12       # one threshold: the ROC curve is the diagonal


# ------------------------------------------------------------------------------------------ synthetic input generator
@pytest.mark.parametrize("P,fixed", [(1, True), (4, True), (10, False), (37, False)])
@pytest.mark.parametrize("idx_dtype", [torch.int64, torch.int32])
def test_datagen_bit_exact_against_oracle(P, fixed, idx_dtype):
    """Device generator vs oracle.philox_bags / philox_dense: the integer bookkeeping (bag lengths, offsets, sorted unique
    indices) must agree bit for bit; the transformation itself is pinned to the reference generator on CPU."""
    from dlrm_amd.datagen import UniformBatchGenerator
    rows = [1, 2, 3, 1000, 39884406, 7]
    B = 777
    gen = UniformBatchGenerator(13, rows, P, fixed, round_targets=True, seed=5, device=dev(), index_dtype=idx_dtype)
    X, lS_o, lS_i, T = gen.batch(B, batch_no=3)
    torch.cuda.synchronize()
    assert np.array_equal(X.cpu().numpy().reshape(-1), O.philox_dense(B * 13, gen._seed(3, 1)))
    assert np.array_equal(T.cpu().numpy().reshape(-1), O.philox_dense(B, gen._seed(3, 2), round_values=True))
    for t, n in enumerate(rows):
        off, idx = O.philox_bags(t, n, B, P, fixed, gen._seed(3, 16))
        assert lS_o[t].dtype == idx_dtype and lS_i[t].dtype == idx_dtype
        assert np.array_equal(lS_o[t].cpu().numpy().astype(np.int64), off), t
        assert np.array_equal(lS_i[t].cpu().numpy().astype(np.int64), idx), t


def test_datagen_full_batch_properties():
    """MLPerf batch (65536) x 26 Criteo tables, reference default pooling (up to 10 lookups): size-independent
    properties — offsets are the running sum of bag lengths, bags are sorted + unique + in range, bag lengths and
    indices follow the reference's distributions — and the batch feeds the embedding kernel."""
    from dlrm_amd import ops
    from dlrm_amd.datagen import UniformBatchGenerator
    rows = [39884406, 39043, 17289, 7420, 20263, 3, 7120, 1543, 63, 38532951, 2953546, 403346, 10, 2208, 11938, 155, 4, 976,
            14, 39979771, 25641295, 39664984, 585935, 12972, 108, 36]
    B = 65536
    gen = UniformBatchGenerator(13, rows, 10, False, seed=727, device=dev())
    X, lS_o, lS_i, T = gen.batch(B, 0)
    X2, lS_o2, lS_i2, _ = gen.batch(B, 1)
    assert not torch.equal(X, X2) and not torch.equal(lS_i[0][:1000], lS_i2[0][:1000])      # batches differ
    Xr, lS_or, lS_ir, _ = gen.batch(B, 0)
    assert torch.equal(X, Xr) and all(torch.equal(a, b) for a, b in zip(lS_i, lS_ir))        # and are reproducible
    assert 0.49 < float(X.mean()) < 0.51 and float(X.min()) >= 0 and float(X.max()) <= 1
    assert set(torch.unique(T).tolist()) <= {0.0, 1.0} and 0.49 < float(T.mean()) < 0.51
    for t, n in enumerate(rows):
        off, idx = lS_o[t], lS_i[t]
        lens = torch.diff(torch.cat([off, torch.tensor([idx.numel()], device=off.device)]))
        assert int(off[0]) == 0 and int(lens.min()) >= 1 and int(lens.max()) <= min(n, 10)
        assert int(idx.min()) >= 0 and int(idx.max()) <= n - 1
        # strictly increasing inside every bag: the only non-increasing steps are at bag starts
        dec = (idx[1:] <= idx[:-1]).nonzero().flatten() + 1
        starts = torch.zeros(idx.numel(), dtype=torch.bool, device=idx.device)
        starts[off] = True
        assert bool(starts[dec].all()), t
        if n >= 100000:    # mean pooling of round(max(1, U*10)) is ~5.05 before duplicates are removed (rare in a large table)
            assert 4.9 < float(lens.float().mean()) < 5.2, (t, float(lens.float().mean()))
            assert abs(float(idx.double().mean()) / (n - 1) - 0.5) < 0.01
    # the generated batch drives the embedding kernel (indices are in range: no fault, pooled sums are finite)
    D = 16
    Ws = [torch.randn(min(n, 50000), D, device=dev()) for n in rows]
    capped = [torch.remainder(i, w.size(0)) for i, w in zip(lS_i, Ws)]
    out = torch.empty(B, len(rows) * D, device=dev())
    ops.emb_fwd(Ws, ops.BagBatch(lS_o, capped), out)
    assert bool(torch.isfinite(out).all())


# ------------------------------------------------------------------------------------------ Criteo binary reader
@pytest.mark.parametrize("idx_dtype", [torch.int64, torch.int32])
@pytest.mark.parametrize("mir", [-1, 40000000, 1000])
def test_criteo_bin_batches_match_reference(tmp_path, idx_dtype, mir):
    """dlrm_amd.criteo_bin.CriteoBinBatches (pinned staging + H2D of the raw block + device conversion kernel) against
    the golden output of the reference's CriteoBinDataset: ids/offsets/targets exact, log(x+1) within 1 ulp; sequential
    iteration with prefetch, random access, the short last batch."""
    from conftest import load_golden
    from dlrm_amd.criteo_bin import CriteoBinBatches
    d, meta = load_golden("criteo_bin")
    f = tmp_path / "day.bin"
    d["raw"].tofile(str(f))
    ds = CriteoBinBatches(str(f), 300, max_ind_range=mir, device=dev(), index_dtype=idx_dtype)
    assert len(ds) == 4
    got = list(ds)
    assert len(got) == 4
    for i, (X, lS_o, lS_i, T) in enumerate(got + [ds.batch(2)]):
        i = 2 if i == 4 else i
        tag = f"m{mir}.b{i}"
        assert lS_i.dtype == idx_dtype and lS_i.shape == d[tag + ".lS_i"].shape
        assert np.array_equal(lS_i.cpu().numpy().astype(np.int64), d[tag + ".lS_i"]), (tag, "ids")
        assert np.array_equal(lS_o.cpu().numpy().astype(np.int64), d[tag + ".lS_o"])
        assert np.array_equal(T.cpu().numpy(), d[tag + ".T"])
        np.testing.assert_allclose(X.cpu().numpy(), d[tag + ".X"], rtol=3e-7, atol=1e-7)
    # a converted batch drives the embedding kernel directly (stacked [26, B] layout)
    from dlrm_amd import ops
    X, lS_o, lS_i, T = got[0]
    rows = 1000 if mir == 1000 else 5000
    Ws = [torch.randn(rows, 8, device=dev()) for _ in range(26)]
    out = torch.empty(X.size(0), 26 * 8, device=dev())
    ops.emb_fwd(Ws, ops.BagBatch(lS_o, torch.remainder(lS_i, rows)), out)
    assert bool(torch.isfinite(out).all())


@pytest.mark.gpu
@pytest.mark.parametrize("T,B", [(26, 1000), (3, 67)])
def test_launch_predicates_choose_the_implementation_on_the_device(T, B):
    """ABI 16 (include/dlrm_hip.h: dlrm_*_pred, dlrm_offsets_iota_flags): a launch given a predicate runs iff (flag != 0) == nonzero, and
    what it then writes are the bits of the call without one; a launch whose predicate fails leaves its outputs untouched.  The proof leaves
    its verdict in a device word (0 = one lookup per bag) and in a pinned host word the host looks at later — never waiting — so a tensor
    object that comes back is known (ops.offsets_iota_state)."""
    from dlrm_amd import ops
    rng = np.random.default_rng(T + B)
    D = 128
    rows = [int(r) for r in rng.integers(1, 5000, size=T)]
    Ws = [to_dev(rng.standard_normal((n, D)).astype(np.float32)) for n in rows]
    idx = torch.stack([to_dev(rng.integers(0, n, size=B)) for n in rows])
    off = torch.arange(B, device=dev()).repeat(T, 1)
    x = to_dev(rng.standard_normal((B, D)).astype(np.float32))
    bags = ops.BagBatch(off, idx)
    F = T + 1
    ldr = (ops.interact_out_width(F, D, False) + 3) & ~3
    zero, one = torch.zeros(1, dtype=torch.int32, device=dev()), torch.ones(1, dtype=torch.int32, device=dev())
    # references without predicates
    ly0 = torch.empty((B, T * D), device=dev()); ops.emb_fwd(Ws, bags, ly0)
    R0 = torch.empty((B, ldr), device=dev()); ops.interact_fwd_gather(x, Ws, bags, D, False, R0)
    dR = to_dev(rng.standard_normal((B, ldr)).astype(np.float32))
    dx0, dE0 = torch.empty((B, D), device=dev()), torch.empty((B, T * D), device=dev())
    ops.interact_bwd_gather(x, Ws, bags, D, ops.INTERACT_RELU_X, dR, dx0, dE0)
    for flag, nz, runs in ((zero, 0, True), (one, 0, False), (one, 1, True), (zero, 1, False)):
        ly = torch.full((B, T * D), -7.0, device=dev()); ops.emb_fwd(Ws, bags, ly, pred=(flag, nz))
        Rg = torch.full((B, ldr), -7.0, device=dev()); ops.interact_fwd_gather(x, Ws, bags, D, False, Rg, pred=(flag, nz))
        Rp = torch.full((B, ldr), -7.0, device=dev()); ops.interact_fwd((x, ly0), D, False, Rp, pred=(flag, nz))
        dxg, dEg = torch.full((B, D), -7.0, device=dev()), torch.full((B, T * D), -7.0, device=dev())
        ops.interact_bwd_gather(x, Ws, bags, D, ops.INTERACT_RELU_X, dR, dxg, dEg, pred=(flag, nz))
        dxp, dEp = torch.full((B, D), -7.0, device=dev()), torch.full((B, T * D), -7.0, device=dev())
        ops.interact_bwd((x, ly0), D, ops.INTERACT_RELU_X, dR, (dxp, dEp), pred=(flag, nz))
        torch.cuda.synchronize()
        if runs:
            assert torch.equal(ly, ly0) and torch.equal(Rg, R0) and torch.equal(Rp, R0)
            assert torch.equal(dxg, dx0) and torch.equal(dEg, dE0) and torch.equal(dxp, dx0) and torch.equal(dEp, dE0)
        else:
            for t_ in (ly, Rg, Rp, dxg, dEg, dxp, dEp):
                assert bool((t_ == -7.0).all())
    ops.check_index_errors(sync=True)
    # the proof that stays on the device
    s0 = dict(ops.IOTA_STATS)
    fresh = torch.arange(B, device=dev()).repeat(T, 1)
    f1 = ops.offsets_iota_state(fresh)
    assert isinstance(f1, torch.Tensor) and f1.dtype == torch.int32 and f1.numel() == 1
    bad = fresh.clone(); bad[T - 1, 3] = 2; bad[0, 5] = 4
    f2 = ops.offsets_iota_state(bad)
    torch.cuda.synchronize()
    assert int(f1.item()) == 0 and int(f2.item()) == 2
    assert ops.IOTA_STATS["device_predicates"] == s0["device_predicates"] + 2 and ops.IOTA_STATS["checked"] == s0["checked"]
    assert ops.offsets_iota_state(fresh) is True and ops.offsets_iota_state(bad) is False          # known by now: no second pass
    assert ops.IOTA_STATS["device_predicates"] == s0["device_predicates"] + 2 and ops.IOTA_STATS["cached"] == s0["cached"] + 2
    lst = [torch.arange(B, device=dev(), dtype=torch.int32) for _ in range(T)]                      # list form, int32
    f3 = ops.offsets_iota_state(lst)
    torch.cuda.synchronize()
    assert int(f3.item()) == 0


@pytest.mark.gpu
@pytest.mark.parametrize("M,K", [(65536, 256), (5000, 256), (96, 64), (777, 480), (300, 1024)])
@pytest.mark.parametrize("act,xact", [(2, 1), (0, 1), (2, 0), (1, 2)])
def test_head_layer_backward_in_one_pass_is_the_three_calls(M, K, act, xact):
    """dlrm_linear_head_bwd (the 256 -> 1 head of the top tower: dz = dY * act'(Y), dW, db, dX = (dz W) * xact'(X) in one pass over X)
    against dlrm_act_bwd + dlrm_linear_bwd_weight + dlrm_linear_bwd_data with N = 1: the SAME bits, plain and accumulating, and directly
    against numpy in float64; a shape outside the fast path launches nothing and says so.  (activation codes: 0 none, 1 ReLU, 2 sigmoid)"""
    from dlrm_amd import ops
    from dlrm_amd import _lib
    assert (_lib.ACT_NONE, _lib.ACT_RELU, _lib.ACT_SIGMOID) == (0, 1, 2)
    rng = np.random.default_rng(M + K + act)
    X = rng.standard_normal((M, K)).astype(np.float32)
    if xact == 2:
        X = (1.0 / (1.0 + np.exp(-X))).astype(np.float32)
    W = rng.standard_normal((1, K)).astype(np.float32)
    Y = (1.0 / (1.0 + np.exp(-rng.standard_normal((M, 1))))).astype(np.float32)
    dY = rng.standard_normal((M, 1)).astype(np.float32)
    Xd, Wd, Yd, dYd = to_dev(X), to_dev(W), to_dev(Y), to_dev(dY)
    # the three calls
    dZ = torch.empty((M, 1), device=dev())
    ops.act_bwd(dYd, Yd, act, dZ, None)
    dW0, db0 = torch.empty((1, K), device=dev()), torch.empty((1,), device=dev())
    ops.linear_bwd_weight(dZ, Xd, dW0, db0)
    dX0 = torch.empty((M, K), device=dev())
    ops.linear_bwd_data(dZ, Wd, Xd if xact != 0 else None, xact, dX0, ops.arith_code("f32"))
    # one pass
    dW1, db1, dX1 = torch.full((1, K), 7.0, device=dev()), torch.full((1,), 7.0, device=dev()), torch.full((M, K), 7.0, device=dev())
    assert ops.linear_head_bwd(dYd, Yd, act, Xd, Wd, xact, dX1, dW1, db1) is True
    torch.cuda.synchronize()
    assert torch.equal(dW0, dW1) and torch.equal(db0, db1) and torch.equal(dX0, dX1)
    # accumulate
    ops.linear_bwd_weight(dZ, Xd, dW0, db0, accumulate=True)
    assert ops.linear_head_bwd(dYd, Yd, act, Xd, Wd, xact, None, dW1, db1, accumulate=True) is True
    assert torch.equal(dW0, dW1) and torch.equal(db0, db1)
    # numpy, float64
    y64, x64 = Y.astype(np.float64), X.astype(np.float64)
    dz = dY.astype(np.float64) * ((1 - y64) * y64 if act == 2 else (y64 > 0) if act == 1 else 1.0)
    dxw = dz * W.astype(np.float64)
    dxw = dxw * ((x64 > 0) if xact == 1 else ((1 - x64) * x64) if xact == 2 else 1.0)
    np.testing.assert_allclose(dX1.cpu().numpy(), dxw, rtol=2e-5, atol=1e-6)
    np.testing.assert_allclose(dW1.cpu().numpy() / 2, dz.T @ x64, rtol=2e-4, atol=2e-4 * np.sqrt(M))
    np.testing.assert_allclose(db1.cpu().numpy() / 2, dz.sum(0), rtol=2e-4, atol=2e-4 * np.sqrt(M))
    # outside the fast path: nothing happens
    if K == 64:
        Xo = to_dev(rng.standard_normal((M, 62)).astype(np.float32))
        dWo = torch.full((1, 62), 7.0, device=dev())
        assert ops.linear_head_bwd(dYd, Yd, act, Xo, to_dev(W[:, :62].copy()), xact, None, dWo, None) is False
        torch.cuda.synchronize()
        assert bool((dWo == 7.0).all())
