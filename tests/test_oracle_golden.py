"""The CPU oracle (oracle/) against the golden vectors produced by the live reference.
This is what makes the oracle PINNED: every number here came out of facebookresearch/dlrm itself
(oracle/make_golden.py)."""
import json
import os

import numpy as np
import pytest

from conftest import GOLDEN, ROOT, golden_batches, load_golden, params_with_prefix
from oracle import oracle as O

# cat_wbce_clamp: "cat" interaction + --loss-threshold clamp + --loss-function=wbce (the remaining --arch-* surface)
# learned_pooling: --weighted-pooling=learned (per-row pooling weights as parameters, state_dict keys v_W_l.{k})
TRAIN_FIXTURES = ["config1_b128", "cli_default_mse", "self_interact_d12", "multihot_hotrows", "kaggle_b2048", "cat_wbce_clamp",
                  "learned_pooling"]


def model_options(meta):
    return dict(interaction=meta.get("interaction", "dot"), loss_threshold=meta.get("loss_threshold", 0.0),
                loss_ws=meta.get("loss_ws"))


@pytest.mark.parametrize("name", TRAIN_FIXTURES)
def test_training_steps_match_reference(name):
    d, meta = load_golden(name)
    model = O.OracleDLRM(params_with_prefix(d, "init"), sigmoid_top=meta["sigmoid_top"],
                         self_interaction=meta["itself"], loss=meta["loss"], **model_options(meta))
    for s, (X, lS_o, lS_i, T) in enumerate(golden_batches(d, meta)):
        loss, Z = model.train_step(X, lS_o, lS_i, T, meta["lr"])
        np.testing.assert_allclose(Z, d[f"s{s}.Z"], rtol=2e-5, atol=1e-6)
        assert abs(loss - d["losses"][s]) <= 1e-5 * abs(d["losses"][s]), (s, loss, d["losses"][s])
        if s == 0:
            for k, v in params_with_prefix(d, "after1").items():
                np.testing.assert_allclose(model.p[k], v, rtol=1e-4, atol=2e-6, err_msg=k)
    for k, v in params_with_prefix(d, "final").items():
        np.testing.assert_allclose(model.p[k], v, rtol=1e-4, atol=5e-6, err_msg=k)


def test_embedding_bag_forward_is_bit_exact():
    """in-order fp32 sum == torch's EmbeddingBag CPU kernel, bit for bit"""
    import torch
    rng = np.random.default_rng(0)
    for D in (1, 2, 12, 16, 128):
        E = rng.standard_normal((97, D)).astype(np.float32)
        lens = rng.integers(0, 9, size=41)
        off = np.concatenate([[0], np.cumsum(lens)[:-1]]).astype(np.int64)
        idx = rng.integers(0, 97, size=int(lens.sum())).astype(np.int64)
        got = O.emb_fwd(E, idx, off)
        bag = torch.nn.EmbeddingBag(97, D, mode="sum", _weight=torch.tensor(E))
        want = bag(torch.tensor(idx), torch.tensor(off)).detach().numpy()
        assert np.array_equal(got, want), D
        w = rng.standard_normal(idx.shape[0]).astype(np.float32)
        got = O.emb_fwd(E, idx, off, psw=w)
        want = bag(torch.tensor(idx), torch.tensor(off), per_sample_weights=torch.tensor(w)).detach().numpy()
        assert np.array_equal(got, want), ("weighted", D)


def test_sparse_sgd_is_bit_exact():
    """per-lookup in-order fma chain == p.add_(uncoalesced sparse grad, alpha=-lr) on torch CPU"""
    import torch
    rng = np.random.default_rng(1)
    E = rng.standard_normal((5, 16)).astype(np.float32)
    B = 300                                   # ~60 duplicates per row
    off = np.arange(B, dtype=np.int64)
    idx = rng.integers(0, 5, size=B).astype(np.int64)
    dV = rng.standard_normal((B, 16)).astype(np.float32)
    bag = torch.nn.EmbeddingBag(5, 16, mode="sum", sparse=True, _weight=torch.tensor(E.copy()))
    out = bag(torch.tensor(idx), torch.tensor(off))
    out.backward(torch.tensor(dV))
    torch.optim.SGD(bag.parameters(), lr=0.37).step()
    got = O.emb_bwd_sgd(E.copy(), idx, off, dV, 0.37)
    assert np.array_equal(got, bag.weight.detach().numpy())


def test_coo_gradient_of_reference_matches_oracle_backward():
    d, meta = load_golden("config1_b128")
    model = O.OracleDLRM(params_with_prefix(d, "init"), sigmoid_top=meta["sigmoid_top"])
    X, lS_o, lS_i, T = golden_batches(d, meta)[0]
    Z = model.forward(X, lS_o, lS_i)
    _, dp = O.bce(Z, T)
    bot, feat, R, top = model._cache
    dR, gtop = model._mlp_bwd(top, model._tower("top_l", model.ntop, model.sigmoid_top), dp.reshape(Z.shape), True)
    dfeat = O.interact_bwd(feat, dR)
    # reference COO grad of table 0: indices are the input indices verbatim, values = dV[bag(i)]
    assert np.array_equal(d["s0.emb0_grad_indices"][0], lS_i[0])
    bag_of = np.searchsorted(lS_o[0], np.arange(lS_i[0].shape[0]), side="right") - 1
    np.testing.assert_allclose(dfeat[:, 1, :][bag_of], d["s0.emb0_grad_values"], rtol=1e-4, atol=1e-7)
    g = dict((i, (dW, db)) for i, dW, db in gtop)
    np.testing.assert_allclose(g[0][0], d["s0.top0_weight_grad"], rtol=1e-4, atol=1e-7)


def test_rowwise_adagrad_matches_reference():
    d, meta = load_golden("rwsadagrad_tiny")
    # the fixture stores the coalesced sparse grads RWSAdagrad consumed at step 0; rebuild dV-equivalent
    # inputs: one bag per coalesced row with the stored value as its gradient
    for k in range(3):
        E = d[f"init.emb_l.{k}.weight"].copy()
        rows = d[f"s0.emb{k}_cgrad_indices"][0].astype(np.int64)
        vals = d[f"s0.emb{k}_cgrad_values"]
        mom = np.zeros(E.shape[0], dtype=np.float32)
        O.emb_bwd_rowwise_adagrad(E, mom, rows, np.arange(rows.shape[0], dtype=np.int64), vals, meta["lr"], meta["eps"])
        np.testing.assert_allclose(E, d[f"after1.emb_l.{k}.weight"], rtol=1e-5, atol=1e-7)
        np.testing.assert_allclose(mom, d[f"after1.mom{k}"], rtol=1e-5, atol=1e-9)


def test_bookkeeping_is_bit_exact():
    with open(os.path.join(GOLDEN, "bookkeeping.json")) as f:
        book = json.load(f)
    for case in book["partition"]:
        for rank, want in enumerate(case["ranks"]):
            sl = O.my_slice(case["n"], rank, case["size"])
            assert [sl.start, sl.stop, sl.step] == want["slice"]
            mine, splits = O.split_lengths(case["n"], rank, case["size"])
            assert mine == want["my_len"] and splits == want["splits"]
    for key, want in book["pairs"].items():
        F = int(key.split("_")[0][1:])
        li, lj = O.pair_order(F, key.endswith("self1"))
        assert li.tolist() == want["li"] and lj.tolist() == want["lj"]


def test_distributed_reference_semantics():
    """2-rank reference run: rank outputs are the batch slices of the single-process forward, the mean of
    the rank losses is the single-process loss, and (the reference's quirk) embedding rows
    move N x as far as in the single-process run while MLP parameters move identically."""
    d, meta = load_golden("dist2_tiny")
    N, B = meta["size"], meta["B"]
    for r in range(N):
        sl = O.my_slice(B, r, N)
        np.testing.assert_allclose(d[f"rank{r}.s0.Z"], d["single.s0.Z"][sl], rtol=1e-5, atol=1e-7)
    mean_loss = np.mean([d[f"rank{r}.s0.loss"] for r in range(N)])
    assert abs(mean_loss - d["single.s0.loss"]) < 1e-6
    assert d["rank0.n_emb_per_rank"].tolist() == [2, 1]
    assert d["rank0.local_emb_indices"].tolist() == [0, 1] and d["rank1.local_emb_indices"].tolist() == [2]
    # oracle, one step: embedding lr scaled by N reproduces the distributed tables
    model = O.OracleDLRM(params_with_prefix(d, "init"), sigmoid_top=meta["sigmoid_top"])
    X, lS_o, lS_i, T = golden_batches(d, meta)[0]
    model.train_step(X, lS_o, lS_i, T, meta["lr"], emb_lr_scale=float(N))
    m2 = O.OracleDLRM(params_with_prefix(d, "init"), sigmoid_top=meta["sigmoid_top"])
    for s, (X, lS_o, lS_i, T) in enumerate(golden_batches(d, meta)):
        m2.train_step(X, lS_o, lS_i, T, meta["lr"], emb_lr_scale=float(N))
    owner = {0: 0, 1: 0, 2: 1}
    for g in range(3):
        np.testing.assert_allclose(m2.p[f"emb_l.{g}.weight"], d[f"rank{owner[g]}.final.emb_l.{g}.weight"], rtol=2e-4,
                                   atol=2e-6)
    for k in ("bot_l.0.weight", "top_l.2.bias", "top_l.0.weight"):
        np.testing.assert_allclose(m2.p[k], d[f"rank0.final.{k}"], rtol=2e-4, atol=2e-6)


@pytest.mark.parametrize("name", TRAIN_FIXTURES)
def test_torch_port_is_bit_identical_to_reference(name):
    """oracle/torch_port.py (bench.py's cpu_baseline) makes the reference's own operator calls: same bits."""
    import torch
    from oracle.torch_port import TorchPortDLRM
    d, meta = load_golden(name)
    init = {k: torch.from_numpy(v) for k, v in params_with_prefix(d, "init").items()}
    m = TorchPortDLRM(init, meta["sigmoid_top"], meta["itself"], meta["loss"], meta["lr"], **model_options(meta))
    for s, (X, lS_o, lS_i, T) in enumerate(golden_batches(d, meta)):
        loss, Z = m.train_step(torch.from_numpy(X), [torch.from_numpy(o) for o in lS_o],
                               [torch.from_numpy(i) for i in lS_i], torch.from_numpy(T))
        assert np.array_equal(Z.numpy(), d[f"s{s}.Z"])
        assert loss == (float(d["losses"][s]) if meta["loss"] == "wbce" else float(np.float32(d["losses"][s])))   # wbce: float64 mean
    for k, v in params_with_prefix(d, "final").items():
        assert np.array_equal(m.p[k].detach().numpy(), v), k


def test_binary_metrics_oracle_matches_scikit_learn():
    """oracle.binary_metrics (numpy restatement) against the scikit-learn numbers the reference's inference() reports
    (fixture generated by oracle/make_golden.py metrics)."""
    d, meta = load_golden("metrics_sklearn")
    for case in meta["cases"]:
        m = O.binary_metrics(d[case["tag"] + ".scores"], d[case["tag"] + ".targets"])
        for k in ("recall", "precision", "f1", "ap", "roc_auc", "accuracy"):
            assert abs(m[k] - case[k]) <= 1e-12 + 1e-10 * abs(case[k]), (case["tag"], k, m[k], case[k])
        assert m["round_matches"] == case["round_matches"]


def test_datagen_transformation_matches_reference_generator():
    """oracle.bags_from_uniforms replays the uniforms the reference's generate_dist_input_batch consumed
    (dlrm_data_pytorch.py:899-960) and must reproduce its offsets / indices bit for bit."""
    d, meta = load_golden("datagen_uniform")
    for case in meta["cases"]:
        u = d[case["tag"] + ".uniforms"]
        pos = [0]

        def draw(k):
            out = u[pos[0]:pos[0] + k]
            pos[0] += k
            return out
        for k, rows in enumerate(case["ln_emb"]):
            off, idx = O.bags_from_uniforms(draw, rows, case["n"], case["P"], case["fixed"])
            assert np.array_equal(off, d[f"{case['tag']}.off{k}"]) and np.array_equal(idx, d[f"{case['tag']}.idx{k}"]), (case, k)
        assert pos[0] == u.size


def test_philox_known_answers():
    """Philox4x32-10 known-answer vectors (Random123 kat_vectors: zero counter/key, and all-ones)."""
    z = O.philox4x32(0, 0, 0, 0, 0)
    assert [int(x) for x in z] == [0x6627e8d5, 0xe169c58d, 0xbc57ac4c, 0x9b00dbd8]
    o = O.philox4x32(0xffffffff, 0xffffffff, 0xffffffff, 0xffffffff, 0xffffffffffffffff)
    assert [int(x) for x in o] == [0x408f276d, 0x41c83b0e, 0xa20bc7c6, 0x6d5451fd]
    # statistical sanity of the derived streams
    x = O.philox_dense(200001, 7)
    assert x.dtype == np.float32 and 0.0 <= x.min() and x.max() <= 1.0 and abs(x.mean() - 0.5) < 5e-3
    off, idx = O.philox_bags(2, 1000, 500, 10, False, 99)
    lens = np.diff(np.r_[off, idx.size])
    assert lens.min() >= 1 and lens.max() <= 10 and idx.min() >= 0 and idx.max() <= 999
    for b in range(500):
        seg = idx[off[b]:off[b] + lens[b]]
        assert np.all(np.diff(seg) > 0)          # sorted, unique


def test_criteo_bin_transform_matches_reference_dataset():
    """oracle.criteo_bin_transform + the batch byte-range arithmetic of dlrm_amd.criteo_bin against the reference's own
    CriteoBinDataset on a synthetic binary file (data_loader_terabyte.py:197-251): ids and offsets bit-exact, log(x+1)
    to the last bit or one ulp (numpy vs torch log)."""
    from dlrm_amd.criteo_bin import batch_byte_range, num_batches
    d, meta = load_golden("criteo_bin")
    raw = d["raw"]
    nbytes = raw.size * 4
    for case in meta["cases"]:
        bs, mir = case["batch_size"], case["max_ind_range"]
        assert num_batches(nbytes, bs) == case["batches"]
        for i in range(case["batches"]):
            s, e = batch_byte_range(nbytes, bs, i)
            rows = raw.reshape(-1)[s // 4:e // 4].reshape(-1, 40)
            X, lS_o, lS_i, T = O.criteo_bin_transform(rows, mir)
            tag = f"m{mir}.b{i}"
            assert np.array_equal(lS_i, d[tag + ".lS_i"]) and np.array_equal(lS_o, d[tag + ".lS_o"])
            assert np.array_equal(T, d[tag + ".T"])
            np.testing.assert_allclose(X, d[tag + ".X"], rtol=2e-7, atol=0)
    assert batch_byte_range(nbytes, 300, 3) == (3 * 300 * 160, nbytes)      # the short last batch


def test_terabyte_full_batch_fixture_regenerates_and_oracle_reproduces_it():
    """BASELINE.json configs[2] at the FULL batch (B = 65536, 26 tables, D = 128, towers 13-512-256-128 /
    479-1024-1024-512-256-1, rows capped at 2000): the fixture the headline bench number is pinned to.
    (1) tests/golden_tb.py regenerates initial parameters and input batches from numpy's legacy stream and checks
        the SHA-256 digest of every array against what the live reference used (load() raises on a mismatch);
    (2) the oracle (oracle/torch_port.py, the reference's own CPU operator calls) reproduces the golden losses,
        predictions and final parameters of all 3 training steps."""
    import torch
    import golden_tb
    from oracle.torch_port import TorchPortDLRM
    fx = golden_tb.load("terabyte_b65536")
    meta, d = fx.meta, fx.d
    assert meta["B"] == 65536 and len(meta["ln_emb"]) == 26 and meta["m_spa"] == 128 and meta["ln_top"][0] == 479
    m = TorchPortDLRM({k: torch.from_numpy(v) for k, v in fx.init.items()}, meta["sigmoid_top"], meta["itself"],
                      meta["loss"], meta["lr"])
    for s, (X, off, idx, tgt) in enumerate(fx.batches):
        loss, Z = m.train_step(torch.from_numpy(X), [torch.from_numpy(o) for o in off],
                               [torch.from_numpy(i) for i in idx], torch.from_numpy(tgt))
        assert abs(loss - fx.losses[s]) <= 1e-6 * abs(fx.losses[s]), (s, loss, fx.losses[s])
        np.testing.assert_allclose(Z.numpy(), d[f"s{s}.Z"], rtol=2e-6, atol=1e-7)
    for k, v in params_with_prefix(d, "final").items():
        np.testing.assert_allclose(m.p[k].detach().numpy(), v, rtol=1e-5, atol=1e-7, err_msg=k)
    for k, v in params_with_prefix(d, "final_head").items():
        np.testing.assert_allclose(m.p[k].detach().numpy()[:48], v, rtol=1e-5, atol=1e-7, err_msg=k)
        np.testing.assert_allclose(m.p[k].detach().numpy()[-48:], d["final_tail." + k], rtol=1e-5, atol=1e-7, err_msg=k)


def test_multihot_restatement_matches_reference_class():
    """oracle.multihot_tables / multihot_expand against the reference's own Multihot class (torchrec_dlrm/multi_hot.py:
    80-159; fixture multihot_tables.npz): seed-0 lookup tables for "uniform" and "pareto", expanded values (int32,
    table-major) and cumulative offsets (int64) for the constructor's batch size and for a different one."""
    d, meta = load_golden("multihot_tables")
    for c in meta["cases"]:
        tabs = O.multihot_tables(c["sizes"], c["n_emb"], c["dist"])
        for k, t in enumerate(tabs):
            assert t.dtype == np.int32 and np.array_equal(t, d[f"{c['tag']}.table{k}"]), (c["tag"], k)
            assert np.array_equal(t[:, 0], np.arange(c["n_emb"][k]))               # column 0 is the 1-hot id itself
        for b in c["batches"]:
            v, o = O.multihot_expand(d[f"{c['tag']}.b{b}.ids"], tabs)
            assert v.dtype == np.int32 and np.array_equal(v, d[f"{c['tag']}.b{b}.values"])
            assert o.dtype == np.int64 and np.array_equal(o, d[f"{c['tag']}.b{b}.offsets"])
            assert o[-1] == b * sum(c["sizes"])


def test_torchrec_variant_restatement_against_torch_ops():
    """BASELINE configs[4] model semantics (torchrec/models/dlrm.py — third-party, absent: parity UNPINNED): the oracle's
    restatement (triu pair order, logits out of a bare last Linear, BCEWithLogitsLoss) against an independent composition of
    the torch operators torchrec's published model is made of (bmm + torch.triu_indices + cat, F.linear/relu,
    F.binary_cross_entropy_with_logits, autograd, SGD)."""
    import torch
    import torch.nn.functional as Fn
    rng = np.random.default_rng(4)
    D, rows, B = 8, [11, 5, 40], 24
    ln_bot, ln_top = [6, 16, D], [D + 6, 12, 1]
    p = {}
    for k, n in enumerate(rows):
        p[f"emb_l.{k}.weight"] = rng.standard_normal((n, D)).astype(np.float32) * 0.3
    for name, ln in (("bot_l", ln_bot), ("top_l", ln_top)):
        for i in range(len(ln) - 1):
            p[f"{name}.{2 * i}.weight"] = (rng.standard_normal((ln[i + 1], ln[i])) / np.sqrt(ln[i])).astype(np.float32)
            p[f"{name}.{2 * i}.bias"] = rng.standard_normal(ln[i + 1]).astype(np.float32) * 0.1
    X = rng.random((B, 6)).astype(np.float32)
    hot = [3, 1, 2]
    idx = [rng.integers(0, n, size=B * h).astype(np.int64) for n, h in zip(rows, hot)]
    off = [np.arange(B, dtype=np.int64) * h for h in hot]
    T = rng.integers(0, 2, size=(B, 1)).astype(np.float32)
    m = O.OracleDLRM(p, pair_order="triu", final_top_act_none=True, loss="bce_logits")
    loss, Z = m.train_step(X, off, idx, T, 0.5)
    tp = {k: torch.tensor(v, requires_grad=True) for k, v in p.items()}
    x = torch.tensor(X)
    for i in range(2):
        x = torch.relu(Fn.linear(x, tp[f"bot_l.{2 * i}.weight"], tp[f"bot_l.{2 * i}.bias"]))
    ly = [Fn.embedding_bag(torch.tensor(idx[k]), tp[f"emb_l.{k}.weight"], torch.tensor(off[k]), mode="sum") for k in range(3)]
    comb = torch.stack([x] + ly, dim=1)
    inter = torch.bmm(comb, comb.transpose(1, 2))
    iu = torch.triu_indices(4, 4, offset=1)
    z = torch.cat([x, inter[:, iu[0], iu[1]]], dim=1)
    z = torch.relu(Fn.linear(z, tp["top_l.0.weight"], tp["top_l.0.bias"]))
    logits = Fn.linear(z, tp["top_l.2.weight"], tp["top_l.2.bias"])
    E = Fn.binary_cross_entropy_with_logits(logits, torch.tensor(T))
    E.backward()
    np.testing.assert_allclose(Z, logits.detach().numpy(), rtol=1e-5, atol=1e-6)
    assert abs(loss - float(E)) <= 1e-6 * abs(float(E))
    for k, v in tp.items():
        np.testing.assert_allclose(m.p[k], (v - 0.5 * v.grad).detach().numpy(), rtol=1e-4, atol=2e-6, err_msg=k)


def test_crossnet_restatement_against_torch_autograd():
    """oracle.crossnet_fwd / _bwd (DCN-v2 low-rank cross network, torchrec's published forward; parity UNPINNED) against the same
    formula composed of torch operators and differentiated by autograd."""
    import torch
    rng = np.random.default_rng(8)
    B, n, r, L = 9, 24, 5, 3
    x0 = rng.standard_normal((B, n))
    Vs = [rng.standard_normal((r, n)) * 0.3 for _ in range(L)]
    Ws = [rng.standard_normal((n, r)) * 0.3 for _ in range(L)]
    bs = [rng.standard_normal(n) * 0.1 for _ in range(L)]
    g = rng.standard_normal((B, n))
    out, cache = O.crossnet_fwd(x0, Vs, Ws, bs)
    dx0, dVs, dWs, dbs = O.crossnet_bwd(g, Vs, Ws, cache)
    t = lambda a: torch.tensor(a, dtype=torch.float64, requires_grad=True)
    tx0, tV, tW, tb = t(x0), [t(v) for v in Vs], [t(w) for w in Ws], [t(b) for b in bs]
    xl = tx0
    for l in range(L):
        xl = tx0 * (torch.nn.functional.linear(torch.nn.functional.linear(xl, tV[l]), tW[l]) + tb[l]) + xl
    xl.backward(torch.tensor(g))
    np.testing.assert_allclose(out, xl.detach().numpy(), rtol=1e-12, atol=1e-12)
    np.testing.assert_allclose(dx0, tx0.grad.numpy(), rtol=1e-10, atol=1e-12)
    for l in range(L):
        np.testing.assert_allclose(dVs[l], tV[l].grad.numpy(), rtol=1e-10, atol=1e-12)
        np.testing.assert_allclose(dWs[l], tW[l].grad.numpy(), rtol=1e-10, atol=1e-12)
        np.testing.assert_allclose(dbs[l], tb[l].grad.numpy(), rtol=1e-10, atol=1e-12)


def test_reference_baseline_leg_runs_the_compiled_reference_and_agrees_with_the_port():
    """bench.py's baseline leg (oracle/ref_baseline.py): the REAL reference imported from oracle/_ref (`make -C oracle ref`), given
    the GPU run's weights and batch, run through its own loop body — and the port used when oracle/_ref is absent computes the
    same step (losses over 3 iterations equal to 1e-6): the two `cpu_baseline.kind`s time the same arithmetic."""
    import torch
    from oracle import ref_baseline
    from oracle.torch_port import TorchPortDLRM
    if ref_baseline.load_reference() is None:
        pytest.skip("oracle/_ref not built (no reference checkout in this environment)")
    rng = np.random.default_rng(4)
    rows, D, B = [50, 3, 400], 8, 64
    ln_bot, ln_top = [13, 16, D], [D + 6, 12, 1]
    tables = [torch.from_numpy(rng.uniform(-0.3, 0.3, (n, D)).astype(np.float32)) for n in rows]
    mlp = {}
    for name, ln in (("bot_l", ln_bot), ("top_l", ln_top)):
        for i in range(len(ln) - 1):
            mlp[f"{name}.{2 * i}.weight"] = torch.from_numpy(rng.normal(0, 0.3, (ln[i + 1], ln[i])).astype(np.float32))
            mlp[f"{name}.{2 * i}.bias"] = torch.from_numpy(rng.normal(0, 0.3, ln[i + 1]).astype(np.float32))
    X = torch.from_numpy(rng.random((B, 13)).astype(np.float32))
    idx = torch.stack([torch.from_numpy(rng.integers(0, n, B)) for n in rows])
    off = torch.arange(B).repeat(len(rows), 1)
    T = torch.from_numpy(np.round(rng.random((B, 1))).astype(np.float32))
    state = {"m_spa": D, "ln_bot": ln_bot, "ln_top": ln_top, "tables": [t.clone() for t in tables], "mlp": {k: v.clone() for k, v in mlp.items()},
             "batch": (X, off, idx, T), "row_cap": 400}
    cpu, stock = ref_baseline.run(state, lr=0.5, cpu_warmup=1, cpu_steps=2, gpu_device=None)
    assert cpu["kind"] == "reference" and stock is None and cpu["value"] > 0 and len(cpu["ms_per_step_min_max"]) == 2
    params = {f"emb_l.{k}.weight": t.clone() for k, t in enumerate(tables)}
    params.update({k: v.clone() for k, v in mlp.items()})
    port = TorchPortDLRM(params, sigmoid_top=len(ln_top) - 2, loss="bce", lr=0.5)
    for _ in range(cpu["iterations_run"]):
        loss, _ = port.train_step(X, list(off), list(idx), T)
    assert abs(loss - cpu["final_loss"]) <= 1e-6 * abs(loss), (loss, cpu["final_loss"])


def test_mlperf_v2_fixture_agrees_with_the_numpy_oracle_at_step_0():
    """tests/golden/mlperf_v2_dot_b65536.npz was produced by a torch-operator composition of the torchrec model (oracle/make_golden_v2.py);
    the repo's other restatement of the same published model — oracle.OracleDLRM (numpy + the C oracle, pair_order="triu", logits,
    BCEWithLogits) — must give the same step-0 logits on the regenerated inputs and initial parameters: two independent
    restatements of the (unpinnable, third-party) torchrec semantics agree at the benchmark's own scale, 214 lookups per sample."""
    import hashlib
    import sys
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    import make_golden_v2 as G
    from dlrm_amd.torchrec_variant import DLRM
    z = np.load(os.path.join(GOLDEN, "mlperf_v2_dot_b65536.npz"))
    meta = json.loads(bytes(z["meta"]).decode())
    rows, hot, B = meta["rows"], meta["hot"], meta["B"]
    (X, ids, values, off_l, lab), = G.make_inputs(rows, hot, B, 1)
    for tag, a in (("X", X), ("ids", ids), ("values", values), ("off", off_l), ("labels", lab)):
        assert hashlib.sha256(np.ascontiguousarray(a).tobytes()).hexdigest()[:32] == meta["digests"][f"s0.{tag}"], tag
    np.random.seed(meta["seed_init"])
    shell = DLRM(rows, meta["D"], meta["bot"][0], meta["bot"][1:], meta["top"])
    init = {k: v.detach().numpy().copy() for k, v in shell.state_dict().items()}
    ref = O.OracleDLRM(init, pair_order="triu", final_top_act_none=True, loss="bce_logits")
    # the forward is per sample: the first 4096 samples are enough (and keep this test at seconds; the GPU test covers all 65536)
    n = 4096
    off = [off_l[t][:n].astype(np.int64) for t in range(len(rows))]
    idx, o = [], 0
    for h in hot:
        idx.append(values[o:o + n * h].astype(np.int64))
        o += B * h
    logits = ref.forward(X[:n], off, idx)
    np.testing.assert_allclose(logits.reshape(-1), z["s0.logits"][:n], rtol=2e-5, atol=2e-6)


def test_bf16x6_split_is_exact_and_the_dropped_terms_are_one_fp32_rounding():
    """The arithmetic claim behind `--mlp-arith bf16x6` (include/dlrm_hip.h): an fp32 value IS the sum of its three truncation planes, each
    plane is a bfloat16 value, and the six products the kernels keep miss the exact product by at most 2^-21 of its magnitude (m.l + l.m:
    2 x 2^-7 x 2^-15) and by 2^-24 rms — the rms of one fp32 rounding of the product — over 30 binary orders of magnitude, both signs, zero."""
    rng = np.random.default_rng(7)
    a = (rng.standard_normal(200000) * np.exp(rng.standard_normal(200000) * 8)).astype(np.float32)
    b = (rng.standard_normal(200000) * np.exp(rng.standard_normal(200000) * 8)).astype(np.float32)
    a[:3] = [0.0, -1.00390625, 3.0e-30]
    for x in (a, b):
        h, m, l = O.split_bf16x3(x)
        assert np.array_equal((h.astype(np.float64) + m.astype(np.float64)) + l.astype(np.float64), x.astype(np.float64))
        assert np.array_equal((h + m) + l, x)                                   # also in fp32, in this order
        for p in (h, m, l):
            assert not (p.view(np.uint32) & np.uint32(0xffff)).any()            # bfloat16 values
    kept, dropped = O.product_bf16x6(a, b)
    exact = a.astype(np.float64) * b.astype(np.float64)
    assert np.array_equal(kept + dropped, exact)
    nz = exact != 0
    ratio = np.abs(dropped[nz]) / np.abs(exact[nz])
    assert ratio.max() <= 2.0 ** -21 * 1.01
    assert np.sqrt(np.mean(ratio ** 2)) <= 2.0 ** -23.5
