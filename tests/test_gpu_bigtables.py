"""The embedding kernels at the headline's REAL table sizes: the five Criteo-Terabyte tables with 25.6-40.0 M rows
(`tools/visualize.py:1195-1223` of the reference; 13-20 GB each at D = 128, 94 GB together) are allocated at full size on the
device and filled with a CLOSED FORM of (table, row, column); every lookup is drawn from the TOP EIGHTH of its table — byte offsets of
11-20 GB from the table base, where a 32-bit row * stride product or a narrowed index would wrap — plus the very last and the very
first row.  Checked, all through the C ABI:

  * dlrm_emb_fwd (one-hot and ragged multi-lookup bags, int64 / int32 indices): EXACT against the closed form (every value is a
    multiple of 2^-10 below 2^9, so any summation order is exact);
  * dlrm_interact_fwd_gather / _bwd_gather (the headline's fused lookup + interaction): BIT-IDENTICAL to dlrm_interact_fwd / _bwd over
    a feature buffer built from the closed form (small addresses only);
  * dlrm_emb_bwd_sgd in modes SORTED, ATOMIC and DETERMINISTIC with duplicate rows: the WHOLE table afterwards equals closed form +
    torch scatter of -lr * g, chunk by chunk (lr = 2^-1 and gradients multiples of 2^-6 make fma(-lr, g, w) exact in any order): touched
    rows exact, every other row of the 94 GB untouched;
  * dlrm_emb_bwd_rowwise_adagrad: touched rows and accumulator entries within the tolerance written below of a torch restatement
    of optim/rwsadagrad.py:117-143, every other row / accumulator entry untouched (whole-table comparison).

`index_select` of torch on the same device serves as a second witness for the gathers."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

BIG_ROWS = [39884406, 38532951, 39979771, 25641295, 39664984]      # bench.py CRITEO_TB_ROWS[0, 9, 19, 20, 21]
D = 128
CHUNK = 1 << 21                                                    # rows per fill / verify chunk (1 GiB of fp32)


def dev():
    assert torch.cuda.is_available(), "GPU tests need a real MI355X"
    return torch.device("cuda:0")


def closed_rows(t: int, rows: torch.Tensor) -> torch.Tensor:
    """W_t[r, c] = (((r * 1000003 + c * 7919 + t * 104729) mod 2^20) - 2^19) * 2^-10 for the given int64 row numbers -> [n, D] fp32"""
    r = rows.to(torch.int64).view(-1, 1)
    c = torch.arange(D, device=rows.device, dtype=torch.int64).view(1, D)
    v = (r * 1000003 + c * 7919 + t * 104729) & 0xFFFFF
    return (v - (1 << 19)).to(torch.float32) * (1.0 / 1024.0)


def fill(t: int, W: torch.Tensor) -> None:
    n = W.size(0)
    for r0 in range(0, n, CHUNK):
        r1 = min(r0 + CHUNK, n)
        W[r0:r1] = closed_rows(t, torch.arange(r0, r1, device=W.device))


@pytest.fixture(scope="module")
def tables():
    free, _total = torch.cuda.mem_get_info()
    need = sum(BIG_ROWS) * D * 4
    if free < need + (24 << 30):
        pytest.skip("needs %.0f GB of free HBM for the five full-size tables, %.0f GB free" % (need / 1e9 + 24, free / 1e9))
    Ws = [torch.empty((n, D), dtype=torch.float32, device=dev()) for n in BIG_ROWS]
    for t, W in enumerate(Ws):
        fill(t, W)
    torch.cuda.synchronize()
    yield Ws
    del Ws
    torch.cuda.empty_cache()


def high_indices(rng, n_rows: int, count: int, distinct_pool: int = 0) -> np.ndarray:
    """`count` row numbers from the top eighth of a table (with duplicates when drawn from a pool), the last and the first row included"""
    lo = n_rows - n_rows // 8
    if distinct_pool:
        pool = rng.integers(lo, n_rows, size=distinct_pool)
        idx = pool[rng.integers(0, distinct_pool, size=count)]
    else:
        idx = rng.integers(lo, n_rows, size=count)
    idx[0], idx[1], idx[-1] = n_rows - 1, 0, n_rows - 1
    return idx.astype(np.int64)


def to_dev(a, dtype=None):
    t = torch.from_numpy(np.ascontiguousarray(a)).to(dev())
    return t if dtype is None else t.to(dtype)


def assert_tables_are_closed_form_plus(tables, deltas=None, skip_rows=None):
    """every chunk of every table == closed form (+ deltas[t] = (rows int64 [m], values [m, D]) scatter-added); rows listed in
    skip_rows[t] are excluded from the comparison (checked separately with a tolerance)"""
    for t, W in enumerate(tables):
        n = W.size(0)
        d_rows, d_vals = deltas[t] if deltas is not None else (None, None)
        s_rows = skip_rows[t] if skip_rows is not None else None
        for r0 in range(0, n, CHUNK):
            r1 = min(r0 + CHUNK, n)
            want = closed_rows(t, torch.arange(r0, r1, device=W.device))
            got = W[r0:r1]
            if d_rows is not None:
                m = (d_rows >= r0) & (d_rows < r1)
                if bool(m.any()):
                    want.index_add_(0, d_rows[m] - r0, d_vals[m])
            if s_rows is not None:
                m = (s_rows >= r0) & (s_rows < r1)
                if bool(m.any()):
                    got = got.clone()
                    got[s_rows[m] - r0] = 0.0
                    want[s_rows[m] - r0] = 0.0
            assert torch.equal(got, want), "table %d rows [%d, %d) differ from the expected contents" % (t, r0, r1)


@pytest.mark.parametrize("idx_dtype", [torch.int64, torch.int32])
def test_emb_fwd_reads_the_right_rows_beyond_4_gib(tables, idx_dtype):
    from dlrm_amd import ops
    rng = np.random.default_rng(31)
    B, T = 16384, len(tables)
    # one lookup per bag
    idx = [high_indices(rng, n, B) for n in BIG_ROWS]
    I = torch.stack([to_dev(i, idx_dtype) for i in idx])
    Ofs = torch.arange(B, device=dev()).repeat(T, 1).to(idx_dtype)
    out = torch.empty((B, T * D), device=dev())
    ops.emb_fwd(tables, ops.BagBatch(Ofs, I), out)
    for t in range(T):
        assert torch.equal(out[:, t * D:(t + 1) * D], closed_rows(t, to_dev(idx[t]))), t
        assert torch.equal(out[:, t * D:(t + 1) * D], tables[t].index_select(0, to_dev(idx[t]))), t       # second witness
    # ragged bags of 0-4 lookups (sums of multiples of 2^-10 below 2^11: exact in any order)
    offs, idxs, want = [], [], []
    for t, n in enumerate(BIG_ROWS):
        lens = rng.integers(0, 5, size=B)
        off = np.concatenate([[0], np.cumsum(lens)[:-1]]).astype(np.int64)
        ii = high_indices(rng, n, int(lens.sum()))
        bag_of = torch.repeat_interleave(torch.arange(B, device=dev()), to_dev(lens))
        want.append(torch.zeros((B, D), device=dev()).index_add_(0, bag_of, closed_rows(t, to_dev(ii))))
        offs.append(to_dev(off, idx_dtype)); idxs.append(to_dev(ii, idx_dtype))
    ops.emb_fwd(tables, ops.BagBatch(offs, idxs), out)
    ops.check_index_errors(sync=True)
    for t in range(T):
        assert torch.equal(out[:, t * D:(t + 1) * D], want[t]), t


@pytest.mark.parametrize("idx_dtype", [torch.int64, torch.int32])
def test_fused_lookup_interaction_reads_the_right_rows_beyond_4_gib(tables, idx_dtype):
    """the headline's forward / backward kernels (dlrm_interact_fwd_gather / _bwd_gather) on the five full-size tables"""
    from dlrm_amd import ops
    rng = np.random.default_rng(32)
    B, T = 8192, len(tables)
    F = T + 1
    idx = [high_indices(rng, n, B) for n in BIG_ROWS]
    I = torch.stack([to_dev(i, idx_dtype) for i in idx])
    Ofs = torch.arange(B, device=dev()).repeat(T, 1).to(idx_dtype)
    bags = ops.BagBatch(Ofs, I)
    x = to_dev(rng.standard_normal((B, D)).astype(np.float32))
    feat = torch.empty((B, F * D), device=dev())
    feat[:, :D] = x
    for t in range(T):
        feat[:, (1 + t) * D:(2 + t) * D] = closed_rows(t, to_dev(idx[t]))
    Wd = ops.interact_out_width(F, D, False)
    ldr = (Wd + 3) & ~3
    R0 = torch.empty((B, ldr), device=dev())
    ops.interact_fwd([feat[:, :D], feat[:, D:]], D, False, R0)
    R1 = torch.full((B, ldr), 7.0, device=dev())
    ops.interact_fwd_gather(x, tables, bags, D, False, R1)
    assert torch.equal(R0, R1)
    dR = to_dev(rng.standard_normal((B, ldr)).astype(np.float32))
    d0 = torch.empty((B, F * D), device=dev())
    ops.interact_bwd([feat[:, :D], feat[:, D:]], D, False, dR, [d0[:, :D], d0[:, D:]])
    dx, dE = torch.empty((B, D), device=dev()), torch.empty((B, T * D), device=dev())
    ops.interact_bwd_gather(x, tables, bags, D, False, dR, dx, dE)
    ops.check_index_errors(sync=True)
    assert torch.equal(d0[:, :D], dx) and torch.equal(d0[:, D:], dE)


@pytest.mark.parametrize("mode_name", ["sorted", "atomic", "deterministic"])
def test_emb_bwd_sgd_updates_the_right_rows_beyond_4_gib(tables, mode_name):
    from dlrm_amd import ops
    mode = {"sorted": ops.UPD_SORTED, "atomic": ops.UPD_ATOMIC, "deterministic": ops.UPD_DETERMINISTIC}[mode_name]
    rng = np.random.default_rng(33)
    B, T, lr = 65536, len(tables), 0.5
    idx = [high_indices(rng, n, B, distinct_pool=B // 2) for n in BIG_ROWS]            # about half the lookups hit a duplicate row
    I = torch.stack([to_dev(i) for i in idx])
    Ofs = torch.arange(B, device=dev()).repeat(T, 1)
    g = to_dev(rng.integers(-8, 9, size=(B, T * D)).astype(np.float32) / 64.0)
    try:
        ops.emb_bwd_sgd(tables, ops.BagBatch(Ofs, I), g, lr, mode)
        ops.check_index_errors(sync=True)
        deltas = [(to_dev(idx[t]), -lr * g[:, t * D:(t + 1) * D]) for t in range(T)]
        assert_tables_are_closed_form_plus(tables, deltas)
    finally:
        for t, W in enumerate(tables):
            fill(t, W)


def test_emb_bwd_rowwise_adagrad_updates_the_right_rows_beyond_4_gib(tables):
    from dlrm_amd import ops
    rng = np.random.default_rng(34)
    B, T, clr, eps = 65536, len(tables), 0.05, 1e-8
    idx = [high_indices(rng, n, B, distinct_pool=B // 2) for n in BIG_ROWS]
    I = torch.stack([to_dev(i, torch.int32) for i in idx])                              # (config 5 carries int32 ids)
    Ofs = torch.arange(B, device=dev(), dtype=torch.int32).repeat(T, 1)
    g = to_dev(rng.integers(-8, 9, size=(B, T * D)).astype(np.float32) / 64.0)
    moms = [torch.zeros(n, device=dev()) for n in BIG_ROWS]
    try:
        ops.emb_bwd_rowwise_adagrad(tables, moms, ops.BagBatch(Ofs, I), g, clr, eps)
        ops.check_index_errors(sync=True)
        touched = []
        for t in range(T):
            rows_u, inv = torch.unique(to_dev(idx[t]), return_inverse=True)
            gsum = torch.zeros((rows_u.numel(), D), device=dev()).index_add_(0, inv, g[:, t * D:(t + 1) * D])    # exact (multiples of 2^-6)
            mom = (gsum.double() ** 2).mean(dim=1)                                      # optim/rwsadagrad.py:131-133, state starts at 0
            want = closed_rows(t, rows_u).double() - clr * gsum.double() / (mom.sqrt() + eps).unsqueeze(1)
            got = tables[t].index_select(0, rows_u)
            torch.testing.assert_close(got.double(), want, rtol=2e-6, atol=2e-6)
            torch.testing.assert_close(moms[t].index_select(0, rows_u).double(), mom, rtol=2e-6, atol=1e-9)
            assert int(torch.count_nonzero(moms[t])) == int(torch.count_nonzero(mom))   # no other accumulator entry was written
            touched.append(rows_u)
        assert_tables_are_closed_form_plus(tables, None, skip_rows=touched)
    finally:
        for t, W in enumerate(tables):
            fill(t, W)
