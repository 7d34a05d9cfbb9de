"""DLRM_Net (the drop-in module, HIP kernels underneath) against the golden vectors of the reference and
the CPU oracle: forward outputs, losses and updated parameters over consecutive training steps."""
import os

import numpy as np
import pytest
import torch

from conftest import golden_batches, load_golden, params_with_prefix
from oracle import oracle as O

pytestmark = pytest.mark.gpu

# kaggle_b2048 = BASELINE.json configs[1] shapes (26 tables, D = 16, bot 13-512-256-64-16, top 512-256-1, batch 2048; rows capped)
# cat_wbce_clamp: "cat" interaction + --loss-threshold clamp + --loss-function=wbce, all as HIP kernels
# learned_pooling: --weighted-pooling=learned — pooling weights gathered on the device, their dense gradient from dlrm_emb_psw_grad
FIXTURES = ["config1_b128", "cli_default_mse", "self_interact_d12", "multihot_hotrows", "kaggle_b2048", "cat_wbce_clamp",
            "learned_pooling"]


def build_model(meta, init, device, deterministic=True, mode=None):
    import dlrm_amd
    np.random.seed(0)
    m = dlrm_amd.DLRM_Net(meta["m_spa"], np.asarray(meta["ln_emb"]), np.asarray(meta["ln_bot"]), np.asarray(meta["ln_top"]),
                          arch_interaction_op=meta.get("interaction", "dot"), arch_interaction_itself=meta["itself"], sigmoid_bot=-1,
                          sigmoid_top=meta["sigmoid_top"], loss_function=meta["loss"], loss_threshold=meta.get("loss_threshold", 0.0),
                          weighted_pooling=meta.get("weighted_pooling"))
    if meta["loss"] == "wbce":
        m.loss_ws = torch.tensor(meta["loss_ws"], dtype=torch.float64)      # the reference takes it from the CLI global (:391)
    with torch.no_grad():
        sd = m.state_dict()
        assert set(sd.keys()) == set(init.keys())
        for k, v in init.items():
            sd[k].copy_(torch.from_numpy(v))
    m = m.to(device)
    if mode is not None:
        m.emb_update_mode = mode
    elif deterministic:
        m.emb_update_mode = dlrm_amd.ops.UPD_DETERMINISTIC
    return m


@pytest.mark.parametrize("name", FIXTURES)
@pytest.mark.parametrize("mode,arith", [(1, "f32"), (2, "f32"), (0, "f32"), (2, "bf16x6"), (1, "f32-per-layer"), (1, "f32-towers"), (1, "f32-towers-fwd")],
                         ids=["deterministic", "sorted", "atomic", "sorted-bf16x6", "deterministic-per-layer-gemms", "deterministic-towers",
                              "deterministic-towers-forward-too"])
def test_training_matches_reference_golden(name, mode, arith, monkeypatch):
    # (the fixtures are small batches: "f32-towers" runs the backward pass of fp32 towers on the whole-tower kernels of csrc/tower.hip,
    # "f32-towers-fwd" the forward pass too, "f32-per-layer" everything on the per-layer GEMMs, whatever the defaults are — all held to the
    # same reference values)
    if arith in ("f32-per-layer", "f32-towers", "f32-towers-fwd"):
        from dlrm_amd import functional
        monkeypatch.setattr(functional, "TOWER_ROWS", 0 if arith == "f32-per-layer" else 4096)
        monkeypatch.setattr(functional, "TOWER_FWD", arith == "f32-towers-fwd")
        arith = "f32"
    d, meta = load_golden(name)
    device = torch.device("cuda:0")
    model = build_model(meta, params_with_prefix(d, "init"), device, mode=mode)
    model.set_mlp_arith(arith)
    opt = torch.optim.SGD(model.parameters(), lr=meta["lr"])
    for s, (X, lS_o, lS_i, T) in enumerate(golden_batches(d, meta)):
        Xd = torch.from_numpy(X).to(device)
        lS_od = [torch.from_numpy(o).to(device) for o in lS_o]
        lS_id = [torch.from_numpy(i).to(device) for i in lS_i]
        Z = model(Xd, lS_od, lS_id)
        Td = torch.from_numpy(T).to(device)
        E = loss_like_reference(model, Z, Td)
        if meta["loss"] == "wbce":          # the fully fused weighted BCE (one kernel) equals the script's composition
            assert abs(float(model.weighted_bce(Z.detach(), Td)) - float(E.detach())) <= 2e-6 * abs(float(E.detach()))
        np.testing.assert_allclose(Z.detach().cpu().numpy(), d[f"s{s}.Z"], rtol=2e-5, atol=1e-6)
        # the north-star bar: fp32 loss within 1e-5 relative of the reference CPU path
        assert abs(float(E.detach()) - d["losses"][s]) <= 1e-5 * abs(d["losses"][s])
        opt.zero_grad()
        E.backward()
        for e in model.emb_l:
            assert e.weight.grad is None          # fused update: no sparse gradient is materialised
        opt.step()
        if s == 0:
            for k, v in params_with_prefix(d, "after1").items():
                np.testing.assert_allclose(model.state_dict()[k].cpu().numpy(), v, rtol=1e-4, atol=2e-6, err_msg=k)
    for k, v in params_with_prefix(d, "final").items():
        np.testing.assert_allclose(model.state_dict()[k].cpu().numpy(), v, rtol=1e-4, atol=5e-6, err_msg=k)


def test_rwsadagrad_training_matches_reference_golden():
    """3 training steps with row-wise sparse Adagrad (K4) against the live reference + optim/rwsadagrad.py: losses,
    parameters and the row-wise optimizer state.  The reference's own RWSAdagrad class is not importable on the GPU box,
    FusedRWSAdagrad mirrors its hyper-parameters / state layout and DLRM_Net's step pre-hook drives the fused kernel."""
    from dlrm_amd.optim import FusedRWSAdagrad
    d, meta = load_golden("rwsadagrad_tiny")
    meta.setdefault("itself", False)
    device = torch.device("cuda:0")
    init = {k: v for k, v in params_with_prefix(d, "init").items()}
    model = build_model(meta, init, device)
    opt = FusedRWSAdagrad(model.parameters(), lr=meta["lr"], eps=meta["eps"])
    for s, (X, lS_o, lS_i, T) in enumerate(golden_batches(d, meta)):
        Z = model(torch.from_numpy(X).to(device), [torch.from_numpy(o).to(device) for o in lS_o],
                  [torch.from_numpy(i).to(device) for i in lS_i])
        E = model.loss_fn(Z, torch.from_numpy(T).to(device))
        assert abs(float(E.detach()) - d["losses"][s]) <= 1e-5 * abs(d["losses"][s]), (s, float(E.detach()), d["losses"][s])
        opt.zero_grad()
        E.backward()
        opt.step()
        if s == 0:
            for k, v in params_with_prefix(d, "after1").items():
                if k.startswith("mom"):
                    got = opt.state[model.emb_l[int(k[3:])].weight]["momentum"]
                else:
                    got = model.state_dict()[k]
                np.testing.assert_allclose(got.cpu().numpy(), v, rtol=1e-4, atol=2e-6, err_msg=k)
    for k, v in params_with_prefix(d, "final").items():
        got = opt.state[model.emb_l[int(k[3:])].weight]["momentum"] if k.startswith("mom") else model.state_dict()[k]
        np.testing.assert_allclose(got.cpu().numpy(), v, rtol=2e-4, atol=5e-6, err_msg=k)
    assert opt.state[model.emb_l[0].weight]["step"] == 3


@pytest.mark.parametrize("update", ["deterministic", "sorted"])
def test_graphed_step_equals_eager_step(update, monkeypatch):
    """The whole-step HIP graph (dlrm_amd.graph) replays exactly the kernels of the eager step: same losses, same
    parameters, bit for bit (deterministic embedding update), over more steps than the warm-up + capture.
    "sorted": the sort-based update stays in the captured step (the library's own segmented sorter replays; rocPRIM's did not) —
    equal up to the order of the few atomic adds of runs that cross a 64-entry group."""
    import dlrm_amd
    from dlrm_amd.graph import GraphedTrainStep
    from dlrm_amd.optim import FusedSGD
    monkeypatch.setenv("DLRM_GRAPH_SORTED", "1")      # (default "auto" takes the atomic update for a batch this small)
    d, meta = load_golden("config1_b128")
    device = torch.device("cuda:0")
    batches = [(torch.from_numpy(X).to(device), [torch.from_numpy(o).to(device) for o in lS_o],
                [torch.from_numpy(i).to(device) for i in lS_i], torch.from_numpy(T).to(device))
               for X, lS_o, lS_i, T in golden_batches(d, meta)]
    # fixed shapes are required for replay: rebuild the sparse inputs as one lookup per bag
    B = batches[0][0].size(0)
    fixed = []
    for X, lS_o, lS_i, T in batches:
        idx = [i[o.clamp(max=max(i.numel() - 1, 0))] if i.numel() else i for o, i in zip(lS_o, lS_i)]
        off = [torch.arange(B, device=device)] * len(idx)
        fixed.append((X, off, idx, T))
    seq = [fixed[i % len(fixed)] for i in range(7)]
    results = []
    for use_graph in (False, True):
        model = build_model(meta, params_with_prefix(d, "init"), device)
        if update == "sorted":
            model.emb_update_mode = dlrm_amd.ops.UPD_SORTED
        opt = FusedSGD(model.parameters(), lr=meta["lr"])
        losses = []
        if use_graph:
            # one ordinary eager step first (the usual situation: a model that has already trained); its loss is
            # reduced to a float, so no autograd graph of the default stream outlives the step (graph.py, constraints)
            X, off, idx, T = seq[0]
            E = model.loss_fn(model(X, off, idx), T)
            opt.zero_grad()
            E.backward()
            opt.step()
            losses.append(float(E.detach()))
            del E
            step = GraphedTrainStep(model, opt, warmup=2)
            for X, off, idx, T in seq[1:]:
                losses.append(float(step(X, off, idx, T)))
            assert step.captures == 1
            if update == "sorted":
                assert model.emb_update_mode == dlrm_amd.ops.UPD_SORTED, "the graph path fell back to the atomic update"
        else:
            for X, off, idx, T in seq:
                E = model.loss_fn(model(X, off, idx), T)
                opt.zero_grad()
                E.backward()
                opt.step()
                losses.append(float(E.detach()))
        results.append((losses, {k: v.clone() for k, v in model.state_dict().items()}))
    if update == "deterministic":
        assert results[0][0] == results[1][0], (results[0][0], results[1][0])
        for k in results[0][1]:
            assert torch.equal(results[0][1][k], results[1][1][k]), k
    else:
        np.testing.assert_allclose(results[0][0], results[1][0], rtol=1e-6)
        for k in results[0][1]:
            np.testing.assert_allclose(results[0][1][k].cpu().numpy(), results[1][1][k].cpu().numpy(), rtol=1e-5, atol=1e-6, err_msg=k)


def _run_schedule(name, device, mode, graphed, monkeypatch=None, raw="1", ragged_at=None):
    """the fixture's steps with the lr of every step set as the reference's scheduler set it (d["lrs"]); returns (losses, state_dict, step)"""
    import dlrm_amd
    from dlrm_amd.graph import GraphedTrainStep
    from dlrm_amd.optim import FusedSGD
    d, meta = load_golden(name)
    model = build_model(meta, params_with_prefix(d, "init"), device, mode=mode)
    opt = FusedSGD(model.parameters(), lr=meta["lr"])
    if monkeypatch is not None:
        monkeypatch.setenv("DLRM_GTS_RAW", raw)
        monkeypatch.setenv("DLRM_GRAPH_SORTED", "1")
    step = GraphedTrainStep(model, opt, warmup=2) if graphed else None
    losses = []
    for s, (X, lS_o, lS_i, T) in enumerate(golden_batches(d, meta)):
        for g in opt.param_groups:
            g["lr"] = float(d["lrs"][s])
        Xd, Td = torch.from_numpy(X).to(device), torch.from_numpy(T).to(device)
        lS_od = [torch.from_numpy(o).to(device) for o in lS_o]
        lS_id = [torch.from_numpy(i).to(device) for i in lS_i]
        if ragged_at is not None and s == ragged_at:
            # still B lookups per table, but bag 0 is empty and bag 1 holds two of them: NOT one lookup per bag
            for o in lS_od:
                o[1] = 0
        if graphed:
            losses.append(float(step(Xd, lS_od, lS_id, Td)))
        else:
            E = model.loss_fn(model(Xd, lS_od, lS_id), Td)
            opt.zero_grad()
            E.backward()
            opt.step()
            losses.append(float(E.detach()))
            del E
    torch.cuda.synchronize()
    return d, losses, {k: v.detach().clone() for k, v in model.state_dict().items()}, step


@pytest.mark.parametrize("name,mode,graphed,raw", [
    ("lr_schedule_tiny", 1, False, "1"), ("lr_schedule_tiny", 2, False, "1"), ("lr_schedule_tiny", 0, False, "1"),
    ("lr_schedule_onehot_d128", 1, False, "1"), ("lr_schedule_onehot_d128", 2, False, "1"),
    ("lr_schedule_onehot_d128", 1, True, "1"), ("lr_schedule_onehot_d128", 2, True, "1"), ("lr_schedule_onehot_d128", 0, True, "1"),
    ("lr_schedule_onehot_d128", 1, True, "0")],
    ids=["multihot-deterministic", "multihot-sorted", "multihot-atomic", "onehot-deterministic", "onehot-sorted",
         "onehot-graph-deterministic", "onehot-graph-sorted", "onehot-graph-atomic", "onehot-graph-torch-replay"])
def test_lr_schedule_matches_reference_golden(name, mode, graphed, raw, monkeypatch):
    """SURVEY a-17 / VERDICT r5 #2: a learning rate that CHANGES between steps, against the live reference run with its own
    LRPolicyScheduler (oracle/make_golden.py `lr_schedule`: warm-up, plateau, quadratic decay, frozen tail; dlrm_s_pytorch.py:169-203,
    :1621).  Every step's loss within the 1e-5 bar and the final parameters — for the eager step in all three update modes (multi-hot bags
    with hot rows; one lookup per bag on the fused lookup + interaction path) and for the whole-step HIP GRAPH, where the captured update
    kernels read their step size from a device scalar: ONE capture for the whole schedule, one scalar write per change."""
    device = torch.device("cuda:0")
    d, losses, sd, step = _run_schedule(name, device, mode, graphed, monkeypatch, raw)
    assert len(set(np.round(d["lrs"], 9))) >= 4                                   # the schedule really moves
    for s_, (a, b) in enumerate(zip(losses, d["losses"])):
        assert abs(a - b) <= 1e-5 * abs(b), (s_, a, b, float(d["lrs"][s_]))
    for k, v in params_with_prefix(d, "final").items():
        np.testing.assert_allclose(sd[k].cpu().numpy(), v, rtol=1e-4, atol=5e-6, err_msg=k)
    if graphed:
        assert step.captures == 1, "a changing learning rate re-captured the step"
        lrs = [float(x) for x in d["lrs"]]
        # replays start at call 2 (two warm-up calls); the capture bakes nothing in, every later change is one scalar write
        changes = sum(1 for i in range(3, len(lrs)) if lrs[i] != lrs[i - 1])
        assert step.lr_writes == changes, (step.lr_writes, changes, lrs)
        assert (step._exec is not None) == (raw == "1")


def test_graphed_lr_schedule_is_bit_identical_to_eager(monkeypatch):
    """the same schedule, graph against eager, deterministic update: not one bit apart (the device scalar holds the very fp32 value the
    eager launch passes by value)"""
    device = torch.device("cuda:0")
    _, le, sde, _ = _run_schedule("lr_schedule_onehot_d128", device, 1, False)
    _, lg, sdg, step = _run_schedule("lr_schedule_onehot_d128", device, 1, True, monkeypatch)
    assert le == lg, (le, lg)
    for k in sde:
        assert torch.equal(sde[k], sdg[k]), k


@pytest.mark.parametrize("raw", ["1", "0"])
def test_ragged_batch_during_replay_recaptures_and_equals_eager(raw, monkeypatch):
    """a batch that is NOT one lookup per bag (bag 0 empty, bag 1 double) arrives while the graph of the fused lookup +
    interaction path is being replayed.  The proof in front of the replay must catch it, the captured step (and its raw executable handle)
    must be dropped only after the replay in flight has finished, and the re-captured two-kernel step must give what the eager step gives."""
    device = torch.device("cuda:0")
    _, le, sde, _ = _run_schedule("lr_schedule_onehot_d128", device, 1, False, ragged_at=5)
    _, lg, sdg, step = _run_schedule("lr_schedule_onehot_d128", device, 1, True, monkeypatch, raw=raw, ragged_at=5)
    assert step.captures == 2 and step.model.fuse_emb_interact is False
    np.testing.assert_allclose(le, lg, rtol=1e-6)          # (fused and two-kernel forward differ in the last bits before the switch)
    for k in sde:
        np.testing.assert_allclose(sde[k].cpu().numpy(), sdg[k].cpu().numpy(), rtol=1e-5, atol=1e-6, err_msg=k)


def test_inference_metrics_on_device():
    """dlrm_amd.evaluate.inference (forward + device-side metrics) against the oracle forward + numpy metrics."""
    from dlrm_amd.evaluate import inference
    d, meta = load_golden("config1_b128")
    device = torch.device("cuda:0")
    model = build_model(meta, params_with_prefix(d, "init"), device)
    gb = golden_batches(d, meta)
    batches = [(torch.from_numpy(X), [torch.from_numpy(o) for o in lS_o], [torch.from_numpy(i) for i in lS_i],
                torch.from_numpy(T)) for X, lS_o, lS_i, T in gb]
    m = inference(model, batches, device)
    S = d["s0.Z"]                                  # golden predictions of the untrained model exist for batch 0 only
    ref = O.OracleDLRM(params_with_prefix(d, "init"), sigmoid_top=meta["sigmoid_top"])
    Zs = [ref.forward(X, lS_o, lS_i) for X, lS_o, lS_i, T in gb]
    np.testing.assert_allclose(Zs[0], S, rtol=2e-5, atol=1e-6)
    o = O.binary_metrics(np.concatenate(Zs), np.concatenate([T for *_, T in gb]))
    assert m["n"] == o["n"] and m["positives"] == o["positives"]
    # scores differ from the oracle's in the last bits: the rank statistics agree to ~1e-4, the counts almost always exactly
    assert abs(m["roc_auc"] - o["roc_auc"]) < 2e-3 and abs(m["ap"] - o["ap"]) < 2e-3
    assert abs(m["accuracy"] - o["accuracy"]) <= 2.0 / o["n"]


def test_reference_shaped_helpers():
    """apply_emb / interact_features / apply_mlp called one by one like tools/visualize.py does"""
    d, meta = load_golden("config1_b128")
    device = torch.device("cuda:0")
    model = build_model(meta, params_with_prefix(d, "init"), device)
    X, lS_o, lS_i, T = golden_batches(d, meta)[0]
    with torch.no_grad():
        x = model.apply_mlp(torch.from_numpy(X).to(device), model.bot_l)
        ly = model.apply_emb(torch.stack([torch.from_numpy(o) for o in lS_o]).to(device),
                             [torch.from_numpy(i).to(device) for i in lS_i], model.emb_l, model.v_W_l)
        assert len(ly) == 3 and ly[0].shape == (128, 16)
        z = model.interact_features(x, ly)
        p = model.apply_mlp(z, model.top_l)
    np.testing.assert_allclose(p.cpu().numpy(), d["s0.Z"], rtol=2e-5, atol=1e-6)
    # embedding rows are bit-exact against the oracle
    for k in range(3):
        assert np.array_equal(ly[k].cpu().numpy(), O.emb_fwd(d[f"init.emb_l.{k}.weight"], lS_i[k], lS_o[k]))


def test_full_batch_properties_criteo_shape():
    """B = 65536, T = 26, D = 128 (BASELINE configs[2] shapes, table rows capped to keep the test light):
    size-independent properties instead of an oracle run."""
    from dlrm_amd import ops
    device = torch.device("cuda:0")
    g = torch.Generator(device="cpu").manual_seed(1)
    B, T, D = 65536, 26, 128
    rows = [min(n, 200000) for n in [39884406, 39043, 17289, 7420, 20263, 3, 7120, 1543, 63, 38532951, 2953546, 403346,
                                     10, 2208, 11938, 155, 4, 976, 14, 39979771, 25641295, 39664984, 585935, 12972, 108, 36]]
    Ws = [torch.randn(n, D, generator=g).to(device) for n in rows]
    idx = [torch.randint(0, n, (B,), generator=g).to(device) for n in rows]
    off = [torch.arange(B, device=device) for _ in rows]
    bags = ops.BagBatch(off, idx)
    feat = torch.empty((B, (T + 1) * D), device=device)
    ops.emb_fwd(Ws, bags, feat[:, D:])
    # one-hot pooling is a pure gather: must equal index_select exactly, every table, every sample
    for t in range(T):
        assert torch.equal(feat[:, D * (t + 1):D * (t + 2)], Ws[t].index_select(0, idx[t])), t
    # SGD update is linear: applying +g then -g (deterministic order not needed) restores the tables to rounding
    dV = torch.randn(B, T * D, generator=g).to(device) * 0.01
    before = [w.clone() for w in Ws]
    ops.emb_bwd_sgd(Ws, bags, dV, 0.1, ops.UPD_SORTED)
    changed = sum(int(not torch.equal(a, b)) for a, b in zip(before, Ws))
    assert changed == T
    # checksum: total mass moved equals -lr * sum of gradients (sum over all rows of a table, fp64)
    for t in (0, 5, 12, 25):
        moved = (Ws[t].double().sum(0) - before[t].double().sum(0))
        want = -0.1 * dV[:, t * D:(t + 1) * D].double().sum(0)
        assert torch.allclose(moved, want, rtol=1e-3, atol=1e-3), t
    ops.emb_bwd_sgd(Ws, bags, -dV, 0.1, ops.UPD_SORTED)
    for t in range(T):
        assert torch.allclose(Ws[t], before[t], rtol=0, atol=2e-5), t
    # interaction: symmetric in the order of two embedding features up to a permutation of output columns
    x = torch.randn(B, D, generator=g).to(device)
    feat[:, :D] = x
    R = torch.empty((B, 480), device=device)
    ops.interact_fwd([feat[:, :D], feat[:, D:]], D, False, R)
    ref = torch.bmm(feat.view(B, T + 1, D)[:4096], feat.view(B, T + 1, D)[:4096].transpose(1, 2))
    li, lj = O.pair_order(T + 1, False)
    assert torch.allclose(R[:4096, D:D + 351], ref[:, torch.from_numpy(li), torch.from_numpy(lj)], rtol=1e-4, atol=1e-3)
    assert torch.equal(R[:, :D], x) and torch.all(R[:, 479] == 0)


def loss_like_reference(model, Z, T):
    """loss_fn_wrap of the reference (dlrm_s_pytorch.py:148-156): for wbce the per-sample loss comes from model.loss_fn
    (our elementwise kernel) and the class weighting + mean are the SCRIPT's torch ops."""
    if getattr(model, "loss_function", "bce") != "wbce":
        return model.loss_fn(Z, T)
    loss_ws_ = model.loss_ws[T.detach().view(-1).long().cpu()].view_as(T).to(Z.device)
    return (loss_ws_ * model.loss_fn(Z, T)).mean()


@pytest.mark.parametrize("arith,overlap,fuse,uib", [("f32", False, False, False), ("f32", True, False, False), ("bf16x6", True, False, False),
                                                    ("f32", True, True, False), ("f32", False, True, True), ("f32", True, True, True)],
                         ids=["f32-single-stream", "f32-bench-default", "bf16x6-bench-default", "f32-fused-gather-interaction",
                              "f32-update-in-backward", "f32-update-in-backward-two-streams"])
def test_terabyte_full_batch_matches_reference_golden(arith, overlap, fuse, uib):
    """BASELINE.json configs[2] — the configuration the headline samples/s is quoted on — against 3 training steps of the
    live reference at the full batch (B = 65536, 26 tables, D = 128, towers 13-512-256-128 / 479-1024-1024-512-256-1, lr 1.0,
    rows capped at 2000): loss within 1e-5 relative at every step (north_star), predictions rtol 2e-5, parameters rtol 1e-4.
    "bench-default" = bench.py's default 2-stream schedule ("fused-gather-interaction" = DLRM_Net.fuse_emb_interact, the default since round 3; the other cases run the two kernels): embedding lookups / fused update on a side HIP stream beside the
    bottom-MLP GEMMs (from the second step on the update is launched during backward) — same kernels, same results."""
    import golden_tb
    rel = golden_tb.run_on_gpu(torch.device("cuda:0"), arith=arith, overlap=overlap, fuse=fuse, update_in_backward=uib)
    assert len(rel) == 3 and max(rel) <= 1e-5, rel


@pytest.mark.parametrize("fuse,uib", [(True, False), (False, False), (True, True)], ids=["fused-lookups", "two-kernels", "update-in-backward"])
def test_terabyte_full_batch_hbm_resident_tables_match_reference_golden(fuse, uib):
    """the same 3 reference training steps at B = 65536 with the seven big tables capped at 4 M rows instead of
    2000 (fixture terabyte_b65536_cap4m: 14.5 GB of tables, 22-bit row keys, rows looked up 0-3 times per batch) — the sorted
    update's long-key / few-duplicates regime and the lookups' HBM-resident regime pinned to the live reference: losses 1e-5,
    predictions, MLP parameters, head / tail / TOUCHED rows and fp64 column sums of every table."""
    import golden_tb
    import psutil
    if psutil.virtual_memory().available < 24e9:
        pytest.skip("needs ~16 GB of host RAM to regenerate the reference's initial tables")
    # (update-in-backward: steps 2 and 3 take the SGD step of single-lookup rows inside the fused backward — ABI 17 — in exactly the regime
    # it is for: 4 M-row tables whose rows are looked up 0-3 times per batch)
    rel = golden_tb.run_on_gpu(torch.device("cuda:0"), name="terabyte_b65536_cap4m", fuse=fuse, update_in_backward=uib)      # fuse = the product default (bench.py)
    assert max(rel) <= 1e-5, rel


@pytest.mark.parametrize("route", ["flag", "auto"])
def test_coo_escape_hatch_runs_any_torch_optimizer(route):
    """--fused-emb-update=0 / optimizers the fused kernels do not implement (the reference's --optimizer=adagrad is
    torch.optim.Adagrad on sparse gradients, dlrm_s_pytorch.py:1343-1369): the embedding backward materialises the
    reference's uncoalesced sparse COO gradient (dlrm_emb_bwd_coo) and the optimizer's own step consumes it.
    route "flag": model.fused_emb_update = False, .grad exists right after backward();  route "auto": the optimizer-step
    pre-hook recognises an optimizer it has no fused kernel for.  Checked against the CPU oracle (the reference's torch
    operator calls) driven by the same torch.optim.Adagrad, and for SGD against the golden vectors of the reference."""
    from oracle.torch_port import TorchPortDLRM
    d, meta = load_golden("config1_b128")
    device = torch.device("cuda:0")
    init = params_with_prefix(d, "init")
    # (1) SGD through the COO route == the reference's golden run (flag route only: SGD is fused otherwise)
    if route == "flag":
        model = build_model(meta, init, device)
        model.fused_emb_update = False
        opt = torch.optim.SGD(model.parameters(), lr=meta["lr"])
        for s, (X, lS_o, lS_i, T) in enumerate(golden_batches(d, meta)):
            Z = model(torch.from_numpy(X).to(device), [torch.from_numpy(o).to(device) for o in lS_o],
                      [torch.from_numpy(i).to(device) for i in lS_i])
            E = model.loss_fn(Z, torch.from_numpy(T).to(device))
            assert abs(float(E.detach()) - d["losses"][s]) <= 1e-5 * abs(d["losses"][s])
            opt.zero_grad()
            E.backward()
            g = model.emb_l[0].weight.grad
            assert g is not None and g.is_sparse and not g.is_coalesced()
            if s == 0:
                assert np.array_equal(g._indices().cpu().numpy(), d["s0.emb0_grad_indices"])
                np.testing.assert_allclose(g._values().cpu().numpy(), d["s0.emb0_grad_values"], rtol=1e-4, atol=1e-7)
            opt.step()
        for k, v in params_with_prefix(d, "final").items():
            np.testing.assert_allclose(model.state_dict()[k].cpu().numpy(), v, rtol=1e-4, atol=5e-6, err_msg=k)
    # (2) torch.optim.Adagrad
    model = build_model(meta, init, device)
    model.fused_emb_update = (route == "auto")
    # initial_accumulator_value > 0: from a zero accumulator the first Adagrad step is lr * sign(g), which turns the
    # rounding noise of a near-zero gradient element into a full-size difference — not a property of the gradient route
    opt = torch.optim.Adagrad(model.parameters(), lr=0.05, initial_accumulator_value=0.1)
    ref = TorchPortDLRM({k: torch.from_numpy(v) for k, v in init.items()}, meta["sigmoid_top"], meta["itself"], meta["loss"], 0.05)
    ref.opt = torch.optim.Adagrad(list(ref.p.values()), lr=0.05, initial_accumulator_value=0.1)
    for s, (X, lS_o, lS_i, T) in enumerate(golden_batches(d, meta)):
        Z = model(torch.from_numpy(X).to(device), [torch.from_numpy(o).to(device) for o in lS_o],
                  [torch.from_numpy(i).to(device) for i in lS_i])
        E = model.loss_fn(Z, torch.from_numpy(T).to(device))
        loss_ref, _ = ref.train_step(torch.from_numpy(X), [torch.from_numpy(o) for o in lS_o],
                                     [torch.from_numpy(i) for i in lS_i], torch.from_numpy(T))
        assert abs(float(E.detach()) - loss_ref) <= 1e-5 * abs(loss_ref), (s, float(E.detach()), loss_ref)
        opt.zero_grad()
        E.backward()
        opt.step()
    sd = model.state_dict()
    for k, v in ref.p.items():
        np.testing.assert_allclose(sd[k].cpu().numpy(), v.detach().numpy(), rtol=2e-4, atol=5e-6, err_msg=k)


@pytest.mark.parametrize("verdict", ["device_predicate", "host_proof"])
def test_ragged_batch_with_one_lookup_per_bag_on_average_takes_the_two_kernels(verdict, monkeypatch):
    """every table has nnz == B, but table 1 has an EMPTY bag next to a TWO-lookup bag — legal EmbeddingBag
    input the reference computes correctly (dlrm_s_pytorch.py:453-457).  The fused lookup + interaction path (on by default, D = 128)
    must not RUN for it, and the step matches the oracle.  host_proof (rounds 4-6, DLRM_DEVICE_PREDICATE=0): ops.offsets_are_iota proves
    offsets == arange(B) per offsets tensor before anything is launched.  device_predicate (default since ABI 16): the verdict stays on the
    device — the fused launch and the two kernels are BOTH enqueued behind the launch predicate, the one that must not run returns at
    once (nothing reaches the error block), and a tensor object that comes back is known by then (no second device pass)."""
    import dlrm_amd
    from dlrm_amd import dlrm_net as _net, ops
    monkeypatch.setattr(_net, "DEVICE_PREDICATE", verdict == "device_predicate")
    dp = verdict == "device_predicate"
    device = torch.device("cuda:0")
    rng = np.random.default_rng(5)
    D, rows, B = 128, [50, 300, 7], 96
    F = len(rows) + 1
    ln_bot, ln_top = np.asarray([13, 32, D]), np.asarray([D + F * (F - 1) // 2, 24, 1])
    np.random.seed(2)
    model = dlrm_amd.DLRM_Net(D, np.asarray(rows), ln_bot, ln_top, "dot", sigmoid_top=1, loss_function="bce")
    init = {k: v.numpy().copy() for k, v in model.state_dict().items()}
    model = model.to(device)
    model.emb_update_mode = ops.UPD_DETERMINISTIC
    assert model.fuse_emb_interact and ops.gather_ok(F, D)
    X = rng.random((B, 13)).astype(np.float32)
    T = np.round(rng.random((B, 1))).astype(np.float32)
    lS_i = [rng.integers(0, n, size=B).astype(np.int64) for n in rows]
    lS_o = [np.arange(B, dtype=np.int64) for _ in rows]
    lS_o[1] = lS_o[1].copy()
    lS_o[1][41] = 42                 # bag 40 = lookups 40, 41; bag 41 is empty; total still B
    ref = O.OracleDLRM(init, sigmoid_top=1)
    opt = torch.optim.SGD(model.parameters(), lr=0.1)
    seen = dict(ops.IOTA_STATS)
    calls = {"gather": 0}
    orig = ops.interact_fwd_gather
    ops.interact_fwd_gather = lambda *a, **k: (calls.__setitem__("gather", calls["gather"] + 1), orig(*a, **k))[1]
    try:
        for ragged in (True, False):
            off = lS_o if ragged else [np.arange(B, dtype=np.int64) for _ in rows]
            od = [torch.from_numpy(o).to(device) for o in off]
            g0 = calls["gather"]
            Z = model(torch.from_numpy(X).to(device), od, [torch.from_numpy(i).to(device) for i in lS_i])
            E = model.loss_fn(Z, torch.from_numpy(T).to(device))
            opt.zero_grad(); E.backward(); opt.step()
            ops.check_index_errors(sync=True)            # nothing was reported: the ragged batch never reached the gather kernels
            loss, Zr = ref.train_step(X, off, lS_i, T, 0.1)
            assert abs(float(E) - loss) <= 1e-5 * abs(loss), (ragged, float(E), loss)
            np.testing.assert_allclose(Z.detach().cpu().numpy(), Zr, rtol=2e-5, atol=1e-6)
            # (device predicate: the fused launch is ENQUEUED for the ragged batch too — and returns at once on the device)
            assert calls["gather"] - g0 == ((1 if dp else 0) if ragged else 1)
            # the same tensor objects again: the verdict is cached (no second device pass), and still right
            Z2 = model(torch.from_numpy(X).to(device), od, [torch.from_numpy(i).to(device) for i in lS_i])
            assert calls["gather"] - g0 == ((1 if dp else 0) if ragged else 2)
            np.testing.assert_allclose(Z2.detach().cpu().numpy(), ref.forward(X, off, lS_i), rtol=2e-5, atol=1e-6)   # (after the update)
            del Z2
            model._pending_emb.clear()
    finally:
        ops.interact_fwd_gather = orig
    if dp:
        assert ops.IOTA_STATS["device_predicates"] == seen["device_predicates"] + 2 and ops.IOTA_STATS["checked"] == seen["checked"]
        assert ops.IOTA_STATS["cached"] == seen["cached"] + 2
    else:
        assert ops.IOTA_STATS["checked"] == seen["checked"] + 2 and ops.IOTA_STATS["cached"] == seen["cached"] + 2
    for k, v in ref.p.items():
        np.testing.assert_allclose(model.state_dict()[k].cpu().numpy(), v, rtol=1e-4, atol=2e-6, err_msg=k)
    # an in-place edit of a proven tensor invalidates its verdict (version counter)
    o = torch.arange(B, device=device).repeat(len(rows), 1)
    assert ops.offsets_are_iota(o) is True
    o[1, 41] = 42
    assert ops.offsets_are_iota(o) is False


def test_update_in_backward_follows_the_reference_loop_through_ragged_and_tagged_batches():
    """ABI 17 at the module level (DLRM_Net.update_in_backward; EmbeddingBagBackward + SGD.step of the reference, dlrm_s_pytorch.py:1613,1620,
    for single-lookup rows inside the fused backward): six steps of the reference loop against the oracle — step 0 binds the optimizer (step-time
    update), then fresh untagged offsets (the fused backward + its update run under the DEVICE predicate), a RAGGED batch with nnz == B (the
    predicate does not hold: the two-kernel backward wrote every gradient row and the presorted update applies every lookup), producer-tagged
    offsets (no predicate at all), and a learning rate changed between steps (both halves of a step use the lr the optimizer holds at backward).
    Tables mix rows looked up once, several times and never; losses to 1e-5, predictions, every parameter; and the calls really took the path."""
    import dlrm_amd
    from dlrm_amd import ops
    device = torch.device("cuda:0")
    rng = np.random.default_rng(11)
    D, rows, B = 128, [50, 3000, 7, 40000], 192
    F = len(rows) + 1
    ln_bot, ln_top = np.asarray([13, 32, D]), np.asarray([D + F * (F - 1) // 2, 24, 1])
    np.random.seed(3)
    model = dlrm_amd.DLRM_Net(D, np.asarray(rows), ln_bot, ln_top, "dot", sigmoid_top=1, loss_function="bce")
    init = {k: v.numpy().copy() for k, v in model.state_dict().items()}
    model = model.to(device)
    model.emb_update_mode = ops.UPD_SORTED
    model.update_in_backward = True
    ref = O.OracleDLRM(init, sigmoid_top=1)
    opt = torch.optim.SGD(model.parameters(), lr=0.1)
    counts = {"presort": 0, "fused_sgd": 0, "presorted": 0, "plain": 0}
    orig = (ops.emb_presort, ops.interact_bwd_gather, ops.emb_bwd_sgd_presorted, ops.emb_bwd_sgd)

    def w_presort(*a, **k):
        counts["presort"] += 1
        return orig[0](*a, **k)

    def w_bwd(*a, **k):
        counts["fused_sgd"] += 1 if k.get("presorted") is not None else 0
        return orig[1](*a, **k)

    def w_presorted(*a, **k):
        counts["presorted"] += 1
        return orig[2](*a, **k)

    def w_plain(*a, **k):
        counts["plain"] += 1
        return orig[3](*a, **k)

    ops.emb_presort, ops.interact_bwd_gather, ops.emb_bwd_sgd_presorted, ops.emb_bwd_sgd = w_presort, w_bwd, w_presorted, w_plain
    try:
        for s, kind in enumerate(["bind", "fresh", "ragged", "tagged", "fresh-new-lr", "fresh"]):
            X = rng.random((B, 13)).astype(np.float32)
            T = np.round(rng.random((B, 1))).astype(np.float32)
            lS_i = [rng.integers(0, n, size=B).astype(np.int64) for n in rows]
            lS_o = [np.arange(B, dtype=np.int64) for _ in rows]
            if kind == "ragged":
                lS_o[1] = lS_o[1].copy(); lS_o[1][41] = 42          # bag 40 = two lookups, bag 41 empty, nnz still B
            if kind == "fresh-new-lr":
                for g in opt.param_groups:
                    g["lr"] = 0.03
            lr = opt.param_groups[0]["lr"]
            od = [torch.from_numpy(o).to(device) for o in lS_o]
            idd = [torch.from_numpy(i).to(device) for i in lS_i]
            if kind == "tagged":
                od, idd = torch.stack(od), torch.stack(idd)
                ops.mark_one_lookup_per_bag(od)
            Z = model(torch.from_numpy(X).to(device), od, idd)
            E = model.loss_fn(Z, torch.from_numpy(T).to(device))
            opt.zero_grad(); E.backward(); opt.step()
            ops.check_index_errors(sync=True)
            loss, Zr = ref.train_step(X, lS_o, lS_i, T, lr)
            assert abs(float(E) - loss) <= 1e-5 * abs(loss), (s, kind, float(E), loss)
            np.testing.assert_allclose(Z.detach().cpu().numpy(), Zr, rtol=2e-5, atol=1e-6, err_msg=kind)
            for k, v in ref.p.items():
                np.testing.assert_allclose(model.state_dict()[k].cpu().numpy(), v, rtol=1e-4, atol=2e-6, err_msg="%s after step %d (%s)" % (k, s, kind))
    finally:
        ops.emb_presort, ops.interact_bwd_gather, ops.emb_bwd_sgd_presorted, ops.emb_bwd_sgd = orig
    # step 0: the step-time update (optimizer unknown until its first step); every later step: presort + fused backward + presorted update
    assert counts == {"presort": 5, "fused_sgd": 5, "presorted": 5, "plain": 1}, counts
    assert not model._pending_emb


def test_producer_tags_and_the_proof_stream():
    """Round 5: (a) tensors whose PRODUCER wrote 0..B-1 carry its proof (dlrm_amd.datagen with one fixed lookup per bag — list and stacked
    forms —, Multihot over all-ones hot sizes): ops.offsets_are_iota answers without a device pass; a versioned in-place write voids the
    tag, a copy carries none; a generator with variable bag lengths tags nothing.  (b) the two-phase proof (offsets_are_iota_start /
    _finish: the check kernel on its own stream, the host waits for ITS event) — work enqueued on the caller's stream between the two
    halves does not disturb it, the verdict is cached per tensor object afterwards, and a ragged tensor is refused."""
    from dlrm_amd import ops
    from dlrm_amd.datagen import UniformBatchGenerator
    from dlrm_amd.multihot import Multihot
    device = torch.device("cuda:0")
    rows, B = [50, 300, 7, 100000], 512
    gen = UniformBatchGenerator(13, rows, 1, True, seed=3, device=device)
    s0 = dict(ops.IOTA_STATS)
    _, lS_o, _, _ = gen.batch(B, 0)
    assert ops.offsets_are_iota(lS_o) is True
    _, so, si, _ = gen.batch(B, 1, stacked=True)
    assert so.shape == (len(rows), B) and si.shape == (len(rows), B) and ops.offsets_are_iota(so) is True
    assert ops.IOTA_STATS["tagged"] == s0["tagged"] + 2 and ops.IOTA_STATS["checked"] == s0["checked"]      # no device pass so far
    assert torch.equal(so, torch.arange(B, device=device).repeat(len(rows), 1))                               # ... and the tag tells the truth
    import copy
    dc = copy.deepcopy(lS_o[0])                      # Python attributes travel with a deep copy: the tag must NOT (it names object + address)
    assert getattr(dc, ops._IOTA_TAG, None) is not None and not ops._iota_tagged(dc) and ops._iota_tagged(lS_o[0])
    c = so.clone()                                   # a copy is a new object: it takes the device proof
    assert ops.offsets_are_iota(c) is True and ops.IOTA_STATS["checked"] == s0["checked"] + 1
    so[2, 7] = 9                                     # a versioned write voids the tag; the device pass then sees the ragged bags
    assert ops.offsets_are_iota(so) is False and ops.IOTA_STATS["checked"] == s0["checked"] + 2
    ragged_gen = UniformBatchGenerator(13, rows, 3, False, seed=3, device=device)
    _, ro, _, _ = ragged_gen.batch(B, 0)
    assert not any(getattr(o, ops._IOTA_TAG, None) is not None for o in ro)
    mh = Multihot([1, 1, 1], [50, 300, 7], B, device=device)
    _, _, off_l = mh.expand(torch.randint(0, 7, (3, B), device=device, dtype=torch.int32), want_global_offsets=False)
    assert ops.offsets_are_iota(off_l) is True and ops.IOTA_STATS["checked"] == s0["checked"] + 2            # tagged: no pass
    mh2 = Multihot([2, 1, 1], [50, 300, 7], B, device=device)
    _, _, off2 = mh2.expand(torch.randint(0, 7, (3, B), device=device, dtype=torch.int32), want_global_offsets=False)
    assert getattr(off2, ops._IOTA_TAG, None) is None
    # (b) two-phase proof with caller-stream work in between
    fresh = torch.arange(B, device=device).repeat(len(rows), 1)
    h = ops.offsets_are_iota_start(fresh)
    assert not isinstance(h, bool) and h is not None
    a = torch.randn(2048, 2048, device=device)
    for _ in range(8):
        a = a @ a * 1e-3                             # the caller's stream is busy while the proof runs on its own
    assert ops.offsets_are_iota_finish(h) is True
    assert ops.offsets_are_iota_start(fresh) is True                     # cached per object now
    bad = fresh.clone(); bad[0, 3] = 2
    assert ops.offsets_are_iota_finish(ops.offsets_are_iota_start(bad)) is False
    assert ops.offsets_are_iota_finish(True) is True and ops.offsets_are_iota_finish(False) is False and ops.offsets_are_iota_finish(None) is None
    torch.cuda.synchronize()


@pytest.mark.parametrize("towers", [False, True], ids=["per-layer", "towers"])
@pytest.mark.parametrize("D,fuse", [(128, True), (128, False), (16, False)], ids=["fused-lookups", "two-kernels-D128", "generic-D16"])
def test_bottom_tower_relu_derivative_inside_the_interaction_backward(D, fuse, towers, monkeypatch):
    """The interaction backward applies the derivative of the bottom tower's last ReLU to dx (ops.INTERACT_RELU_X) and the tower's backward
    skips its act_bwd pass (MLP_CONSUMER_APPLIES_LAST_ACT): two SGD steps give the SAME bits for every parameter, embedding table and
    prediction as with DLRM_FUSE_ACT_BWD=0, with one C-ABI call per step fewer (dlrm_s_pytorch.py:238-241, 483-504).  A bottom tower
    that ends in a sigmoid is left alone."""
    import dlrm_amd
    from dlrm_amd import functional, ops
    device = torch.device("cuda:0")
    rows, B = [50, 7, 3000, 11, 400], 384
    F = len(rows) + 1
    ln_bot = np.asarray([13, 64, D])
    ln_top = np.asarray([D + F * (F - 1) // 2, 96, 1])

    # (towers: the small-batch path, csrc/tower.hip, where the derivative is a flag of the one backward launch instead of a pass of its own)
    monkeypatch.setattr(functional, "TOWER_ROWS", 4096 if towers else 0)

    def run(flag, sigmoid_bot=-1):
        monkeypatch.setattr(functional, "FUSE_ACT_BWD", flag)
        np.random.seed(11)
        model = dlrm_amd.DLRM_Net(D, np.asarray(rows), ln_bot, ln_top, "dot", sigmoid_bot=sigmoid_bot, sigmoid_top=ln_top.size - 2,
                                  loss_function="bce").to(device)
        model.emb_update_mode = ops.UPD_DETERMINISTIC
        model.fuse_emb_interact = fuse
        opt = torch.optim.SGD(model.parameters(), lr=0.3)
        g = torch.Generator().manual_seed(5)
        out = []
        calls0 = ops.CALL_COUNT[0]
        for _ in range(2):
            X = torch.rand((B, 13), generator=g).to(device)
            idx = torch.stack([torch.randint(0, n, (B,), generator=g) for n in rows]).to(device)
            off = torch.arange(B).repeat(len(rows), 1).to(device)
            T = torch.randint(0, 2, (B, 1), generator=g).float().to(device)
            Z = model(X, off, idx)
            E = model.loss_fn(Z, T)
            opt.zero_grad()
            E.backward()
            opt.step()
            out.append(Z.detach().clone())
        calls = ops.CALL_COUNT[0] - calls0
        out += [p.detach().clone() for p in model.parameters()]
        torch.cuda.synchronize()
        ops.check_index_errors(sync=True)
        return out, calls

    ref, n_ref = run(False)
    got, n_got = run(True)
    assert len(ref) == len(got) and all(torch.equal(a, b) for a, b in zip(ref, got))
    assert n_got == n_ref - (0 if towers else 2)                    # per-layer path: one act_bwd call per step gone
    # a sigmoid at the end of the bottom tower: nothing to fuse (same calls, same bits) — and a different model
    ref_s, n_ref_s = run(False, sigmoid_bot=ln_bot.size - 2)
    got_s, n_got_s = run(True, sigmoid_bot=ln_bot.size - 2)
    assert n_ref_s == n_got_s and all(torch.equal(a, b) for a, b in zip(ref_s, got_s))
    assert not all(torch.equal(a, b) for a, b in zip(ref, ref_s))


def test_coo_escape_hatch_refuses_row_wise_shards_and_checks_indices_synchronously():
    """ADVICE r2 fixes that had no test: (a) DLRM_Net._materialize_coo_grads exits with the reference-style ERROR when the bags
    belong to a row-wise shard (ignore_oob: out-of-range ids are other ranks' rows, a COO gradient would scatter them);
    (b) an out-of-range index raises IndexError BEFORE a sparse gradient is handed to an optimizer's scatter."""
    import dlrm_amd
    from dlrm_amd import ops
    device = torch.device("cuda:0")
    D, B = 16, 8
    W = [torch.zeros((10, D), device=device, requires_grad=True)]
    off = torch.arange(B, device=device).reshape(1, B)
    idx = torch.arange(B, device=device).reshape(1, B) % 10
    dout = torch.ones((B, D), device=device)
    bags = ops.BagBatch(off, idx)
    bags.ignore_oob = True
    with pytest.raises(SystemExit, match="row-wise table shards"):
        dlrm_amd.DLRM_Net._materialize_coo_grads(W, bags, dout)
    assert W[0].grad is None
    bad = idx.clone(); bad[0, 3] = 10
    ops.emb_fwd([w.detach() for w in W], ops.BagBatch(off, bad), torch.empty((B, D), device=device))    # forward reports it (asynchronously)
    with pytest.raises(IndexError, match="out of range"):
        dlrm_amd.DLRM_Net._materialize_coo_grads(W, ops.BagBatch(off, bad), dout)
    assert W[0].grad is None
    dlrm_amd.DLRM_Net._materialize_coo_grads(W, ops.BagBatch(off, idx), dout)
    assert W[0].grad.is_sparse and W[0].grad._values().shape == (B, D)


def test_side_stream_keeps_its_operands_alive_until_the_join():
    """ADVICE r2 fix that had no test (dlrm_net.py `_side_keep`): with overlap_streams the fused update is launched on the side
    stream from backward; the gradient buffer and the index / offset tensors it reads must stay referenced until the streams join,
    even when the caller drops its batch right after backward()."""
    import dlrm_amd
    from dlrm_amd import ops
    d, meta = load_golden("config1_b128")
    device = torch.device("cuda:0")
    model = build_model(meta, params_with_prefix(d, "init"), device, mode=ops.UPD_SORTED)
    model.overlap_streams = True
    opt = torch.optim.SGD(model.parameters(), lr=meta["lr"])
    import weakref
    for s, (X, lS_o, lS_i, T) in enumerate(golden_batches(d, meta)):
        od = [torch.from_numpy(o).to(device) for o in lS_o]
        idd = [torch.from_numpy(i).to(device) for i in lS_i]
        Z = model(torch.from_numpy(X).to(device), od, idd)
        E = model.loss_fn(Z, torch.from_numpy(T).to(device))
        assert abs(float(E.detach()) - d["losses"][s]) <= 1e-5 * abs(d["losses"][s])
        opt.zero_grad()
        E.backward()
        probe = weakref.ref(idd[0])
        if s >= 1:
            # from the second step on the update was launched during backward on the side stream: the model holds the operands
            assert any(t is idd[0] for t in model._side_keep), "side-stream update does not keep the index tensor alive"
        del od, idd, Z, E
        if s >= 1:
            assert probe() is not None
        opt.step()                  # joins the side stream and releases the operands
        assert model._side_keep == []
    for k, v in params_with_prefix(d, "final").items():
        np.testing.assert_allclose(model.state_dict()[k].cpu().numpy(), v, rtol=1e-4, atol=5e-6, err_msg=k)


def _reference_dir() -> str:
    """$DLRM_REFERENCE (a checkout), else oracle/_ref: the reference compiled where it lay by `make -C oracle ref`
    (__graft_entry__.build() runs it in the build container; the directory travels to the GPU box with the tree)."""
    env = os.environ.get("DLRM_REFERENCE", "")
    if env and os.path.isfile(os.path.join(env, "dlrm_s_pytorch.py")):
        return env
    from oracle.build_ref import ref_dir
    return ref_dir() or ""


_REF = _reference_dir()


@pytest.mark.skipif(not _REF, reason="no reference: neither $DLRM_REFERENCE nor a usable oracle/_ref (run `make -C oracle ref` where a "
                                     "checkout exists)")
def test_launcher_trains_under_the_unmodified_reference_run(tmp_path):
    """`python -m dlrm_amd.launch` — the reference's own run() (CLI, data generation, training loop, timing,
    printing, LR scheduler) with OUR DLRM_Net / ext_dist swapped in — trains on the GPU end to end: losses are printed by the
    reference loop, finite, and equal to the reference's own CPU run of the same command line within the 1e-5 bar at the
    first iteration (identical seeds => identical initial parameters and data) and at every later one — under a learning-rate
    SCHEDULE (round 6, SURVEY a-17): a step that used the wrong lr shows up in the next printed loss."""
    import re
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, PYTHONPATH=root, PYTHONDONTWRITEBYTECODE="1")
    cli = ["--arch-sparse-feature-size=16", "--arch-embedding-size=1000-1000-1000", "--arch-mlp-bot=13-512-16",
           "--arch-mlp-top=22-256-1", "--mini-batch-size=128", "--data-generation=random", "--num-batches=6", "--nepochs=1",
           "--print-freq=1", "--print-time", "--numpy-rand-seed=123", "--learning-rate=0.1",
           # the reference's LRPolicyScheduler (:169-203, stepped every iteration at :1621): warm-up over 2 steps, quadratic decay from step 3
           # over 3 steps — lr differs on almost every one of the 6 iterations, for the dense step AND the fused embedding update
           "--lr-num-warmup-steps=2", "--lr-decay-start-step=3", "--lr-num-decay-steps=3"]
    ours = subprocess.run([sys.executable, "-m", "dlrm_amd.launch", "--reference", _REF, "--"] + cli + ["--use-gpu"],
                          cwd=tmp_path, env=env, capture_output=True, text=True, timeout=600)
    assert ours.returncode == 0, ours.stderr[-3000:]
    stub = ("import sys, types; tb = types.ModuleType('torch.utils.tensorboard'); "
            "tb.SummaryWriter = type('S', (), {'__init__': lambda s, *a, **k: None, 'add_scalar': lambda s, *a, **k: None, 'close': lambda s: None}); "
            "import torch.utils; sys.modules['torch.utils.tensorboard'] = tb; sys.path.insert(0, %r); sys.argv = ['dlrm_s_pytorch.py'] + %r; "
            "import dlrm_s_pytorch as r; r.run()" % (_REF, cli))
    ref = subprocess.run([sys.executable, "-c", stub], cwd=tmp_path, env=env, capture_output=True, text=True, timeout=900)
    assert ref.returncode == 0, ref.stderr[-3000:]
    pat = re.compile(r"Finished training it (\d+)/\d+ of epoch 0, [\d.]+ ms/it, loss ([\d.]+)")
    lo, lr_ = pat.findall(ours.stdout), pat.findall(ref.stdout)
    assert len(lo) >= 6 and len(lo) == len(lr_), (ours.stdout[-1500:], ref.stdout[-1500:])
    for (i, a), (j, b) in zip(lo, lr_):
        assert i == j and abs(float(a) - float(b)) <= 2e-6 + 1e-5 * float(b), (i, a, b)     # printed with 6 decimals


@pytest.mark.parametrize("D,hot,rows", [(128, [3, 1, 7, 2, 1], [50, 7, 3000, 11, 400]), (16, [2, 1, 5], [30, 9, 100])])
def test_torchrec_variant_with_multihot_inputs_matches_oracle(D, hot, rows):
    """BASELINE.json configs[4] semantics end to end on the GPU: dlrm_amd.torchrec_variant.DLRM + DLRMTrain (triu interaction
    order, logits, BCEWithLogitsLoss) fed by dlrm_amd.multihot.Multihot (int32 multi-hot bags expanded on the device), against
    the oracle's restatement of torchrec's published model, over 2 SGD steps — plus one step with the fused row-wise Adagrad
    (the optimizer the benchmark uses) against the C oracle's row-wise Adagrad."""
    from dlrm_amd.multihot import Multihot
    from dlrm_amd.optim import FusedRWSAdagrad, FusedSGD
    from dlrm_amd.torchrec_variant import DLRM, DLRMTrain
    device = torch.device("cuda:0")
    rng = np.random.default_rng(D)
    B, dense_in = 64, 13
    np.random.seed(5)
    model = DLRM(rows, D, dense_in, [32, D], [48, 24, 1]).to(device)
    assert not isinstance(list(model.top_l.children())[-1], (torch.nn.ReLU, torch.nn.Sigmoid))
    init = {k: v.detach().cpu().numpy().copy() for k, v in model.state_dict().items()}
    train = DLRMTrain(model)
    mh = Multihot(hot, rows, B, device=device, seed=3)
    tabs = [t.cpu().numpy() for t in mh.multi_hot_tables_l]
    ref = O.OracleDLRM(init, pair_order="triu", final_top_act_none=True, loss="bce_logits")
    opt = FusedSGD(model.parameters(), lr=0.3)
    for s in range(2):
        ids = np.stack([rng.integers(0, n, size=B) for n in rows])
        X = rng.random((B, dense_in)).astype(np.float32)
        labels = rng.integers(0, 2, size=B)
        lS_o, lS_i = mh.to_model_inputs(torch.from_numpy(ids).to(device).int())
        loss, (loss_d, logits, _) = train(torch.from_numpy(X).to(device), lS_o, lS_i, torch.from_numpy(labels).to(device))
        v, o = O.multihot_expand(ids, tabs)
        off = [(o[t * B:(t + 1) * B] - o[t * B]).astype(np.int64) for t in range(len(rows))]
        idx = [v[o[t * B]:o[(t + 1) * B]].astype(np.int64) for t in range(len(rows))]
        want_loss, want_logits = ref.train_step(X, off, idx, labels.reshape(B, 1).astype(np.float32), 0.3)
        np.testing.assert_allclose(logits.cpu().numpy(), want_logits, rtol=2e-5, atol=2e-6)
        assert abs(float(loss) - want_loss) <= 1e-5 * abs(want_loss), (s, float(loss), want_loss)
        opt.zero_grad()
        loss.backward()
        opt.step()
    sd = model.state_dict()
    for k, v in ref.p.items():
        np.testing.assert_allclose(sd[k].cpu().numpy(), v, rtol=1e-4, atol=5e-6, err_msg=k)
    # the benchmark's optimizer: fused row-wise Adagrad, lr 0.005, eps 1e-8 — one step must move only the touched rows
    opt2 = FusedRWSAdagrad(model.parameters(), lr=0.005, eps=1e-8)
    before = [e.weight.detach().clone() for e in model.emb_l]
    loss, _ = train(torch.from_numpy(X).to(device), lS_o, lS_i, torch.from_numpy(labels).to(device))
    opt2.zero_grad()
    loss.backward()
    opt2.step()
    for t, (e, w0) in enumerate(zip(model.emb_l, before)):
        touched = torch.zeros(rows[t], dtype=torch.bool, device=device)
        touched[lS_i[t].long()] = True
        changed = (e.weight != w0).any(dim=1)
        assert bool((changed & ~touched).sum() == 0) and bool(changed.sum() > 0), t


@pytest.mark.parametrize("interaction,arith", [("dot", "f32"), ("dot", "bf16"), ("dcn", "f32"), ("dcn", "bf16")])
def test_mlperf_v2_bench_configuration_matches_golden(interaction, arith):
    """VERDICT r2 missing-3 / weak-2: bench.py's `--workload mlperf_v2_multihot` configuration at its own scale — B = 65536, 214 int32
    lookups per sample expanded on the device, torchrec model semantics (triu dot or DCN-v2), fused row-wise Adagrad + dense Adagrad
    lr 0.005 eps 1e-8 — for 3 steps against tests/golden/mlperf_v2_*_b65536.npz (the reference's own RWSAdagrad class driving a
    torch-operator restatement of the model; inputs and initial parameters regenerated by the product and SHA-checked).  f32: 1e-5;
    bf16 (the benchmark's arithmetic): the measured tolerance stated in tests/golden_v2.py.  Same check as bench.py's parity_check."""
    import golden_v2
    if not golden_v2.available(interaction):
        pytest.skip("fixture not generated")
    r = golden_v2.run_on_gpu(torch.device("cuda:0"), interaction, arith)
    assert set(r) == {"bench", "conditioned"}


def test_graphed_step_survives_host_syncs_at_full_batch():
    """profiles/r02/graph_probe.md (d): at B = 65536 a run "a few replays, host synchronisation, more replays" ended in a GPU
    memory fault in round 1.  GraphedTrainStep now synchronises its stream after every replay; this replays 36 steps at the
    Criteo-Terabyte shapes (26 tables, D = 128, full towers, rows capped) with host syncs and a D2H read in the middle and
    requires losses bit-identical to the eager step (deterministic embedding update)."""
    import golden_tb
    from dlrm_amd.graph import GraphedTrainStep
    from dlrm_amd.optim import FusedSGD
    fx = golden_tb.load("terabyte_b65536")
    meta = fx.meta
    device = torch.device("cuda:0")
    batches = [(torch.from_numpy(X).to(device), torch.from_numpy(off).to(device), torch.from_numpy(idx).to(device),
                torch.from_numpy(t).to(device)) for X, off, idx, t in fx.batches]
    meta2 = dict(meta, itself=False)
    runs = []
    for use_graph in (False, True):
        model = build_model(meta2, fx.init, device)        # deterministic embedding update
        opt = FusedSGD(model.parameters(), lr=0.05)
        step = GraphedTrainStep(model, opt, warmup=2) if use_graph else None
        losses = []
        for i in range(36):
            X, off, idx, t = batches[i % len(batches)]
            if use_graph:
                losses.append(step(X, off, idx, t))
            else:
                E = model.loss_fn(model(X, off, idx), t)
                opt.zero_grad()
                E.backward()
                opt.step()
                losses.append(E.detach())
            if i in (6, 7, 15, 29):
                torch.cuda.synchronize()                   # the pattern that faulted
                _ = float(losses[-1])
            losses[-1] = losses[-1].clone()
        torch.cuda.synchronize()
        runs.append([float(x) for x in losses])
        if use_graph:
            assert step.captures == 1
        del model, opt, step
    assert runs[0] == runs[1], [(a, b) for a, b in zip(*runs) if a != b][:4]


@pytest.mark.parametrize("B,n,r,L,arith", [(300, 64, 16, 3, "f32"), (4096, 3456, 512, 3, "f32"), (4096, 3456, 512, 2, "bf16x6"), (70, 20, 4, 1, "f32")])
def test_dcn_v2_cross_network_matches_oracle(B, n, r, L, arith):
    """LowRankCrossNetFunction (DCN-v2, the MLPerf-v2 interaction: GEMM kernels + dlrm_cross_fwd / _bwd, hand-written backward)
    against the float64 oracle: output, gradient of x_0 and of every V / W / bias — at the benchmark's 27 x 128 -> rank 512 shape too."""
    from dlrm_amd import ops
    from dlrm_amd.functional import LowRankCrossNetFunction
    device = torch.device("cuda:0")
    rng = np.random.default_rng(B + n)
    x0 = rng.standard_normal((B, n)).astype(np.float32)
    Vs = [(rng.standard_normal((r, n)) / np.sqrt(n)).astype(np.float32) for _ in range(L)]
    Ws = [(rng.standard_normal((n, r)) / np.sqrt(r)).astype(np.float32) for _ in range(L)]
    bs = [(rng.standard_normal(n) * 0.1).astype(np.float32) for _ in range(L)]
    g = rng.standard_normal((B, n)).astype(np.float32)
    want, cache = O.crossnet_fwd(x0, Vs, Ws, bs)
    dx0, dVs, dWs, dbs = O.crossnet_bwd(g, Vs, Ws, cache)
    tx0 = torch.from_numpy(x0).to(device).requires_grad_(True)
    ps = [torch.from_numpy(a).to(device).requires_grad_(True) for l in range(L) for a in (Vs[l], Ws[l], bs[l])]
    out = LowRankCrossNetFunction.apply(ops.arith_code(arith), tx0, *ps)
    out.backward(torch.from_numpy(g).to(device))
    scale = lambda a: max(1.0, float(np.abs(a).max()))
    np.testing.assert_allclose(out.detach().cpu().numpy(), want, rtol=2e-5, atol=2e-5 * scale(want))
    np.testing.assert_allclose(tx0.grad.cpu().numpy(), dx0, rtol=1e-4, atol=2e-5 * scale(dx0))
    for l in range(L):
        np.testing.assert_allclose(ps[3 * l].grad.cpu().numpy(), dVs[l], rtol=1e-4, atol=3e-5 * scale(dVs[l]))
        np.testing.assert_allclose(ps[3 * l + 1].grad.cpu().numpy(), dWs[l], rtol=1e-4, atol=3e-5 * scale(dWs[l]))
        np.testing.assert_allclose(ps[3 * l + 2].grad.cpu().numpy(), dbs[l], rtol=1e-4, atol=3e-5 * scale(dbs[l]))


def test_dcn_v2_cross_network_bf16_storage_equals_in_loop_rounding():
    """arith "bf16" at the benchmark's shape (27 x 128 features -> rank 512, 3 layers): the bf16-STORAGE form of the cross network — x_l and
    du kept as bf16 copies written by the elementwise kernels, v and dv as bf16-only GEMM outputs, weight gradients from the stored
    operands (ds_read_b64_tr_b16), g + dv.V summed in the GEMM epilogue — against the in-loop rounding form (fp32 storage, every GEMM rounds
    its operands to bf16 in the k-loop).  Same products: the output and dx_0 agree to fp32 round-off of the same bf16 products except where
    the storage form rounds v / dv to bf16 a second time as OPERANDS of the next product (the in-loop form does the same rounding when it
    reads them) — i.e. everywhere; parameter gradients differ by the summation order only."""
    from dlrm_amd import functional, ops
    from dlrm_amd.functional import LowRankCrossNetFunction
    device = torch.device("cuda:0")
    rng = np.random.default_rng(11)
    B, n, r, L = 8192, 3456, 512, 3
    x0 = torch.from_numpy(rng.standard_normal((B, n)).astype(np.float32)).to(device)
    g = torch.from_numpy(rng.standard_normal((B, n)).astype(np.float32)).to(device)
    base = []
    for l in range(L):
        base += [torch.from_numpy((rng.standard_normal((r, n)) / np.sqrt(n)).astype(np.float32)).to(device),
                 torch.from_numpy((rng.standard_normal((n, r)) / np.sqrt(r)).astype(np.float32)).to(device),
                 torch.from_numpy((rng.standard_normal(n) * 0.1).astype(np.float32)).to(device)]
    res = []
    saved = functional.BF16_STORAGE, functional.CROSS_FUSE
    try:
        for storage, fuse in ((False, False), (True, False), (True, True)):
            functional.BF16_STORAGE, functional.CROSS_FUSE = storage, fuse
            tx0 = x0.clone().requires_grad_(True)
            ps = [p.clone().requires_grad_(True) for p in base]
            out = LowRankCrossNetFunction.apply(ops.arith_code("bf16"), tx0, *ps)
            out.backward(g)
            torch.cuda.synchronize()
            res.append((out.detach(), tx0.grad, [p.grad for p in ps]))
    finally:
        functional.BF16_STORAGE, functional.CROSS_FUSE = saved
    (o0, d0, g0), (o1, d1, g1), (o2, d2, g2) = res
    sc = lambda a: float(a.abs().max())      # noqa: E731
    assert float((o0 - o1).abs().max()) <= 1e-5 * sc(o0)
    assert float((d0 - d1).abs().max()) <= 2e-5 * sc(d0)
    for k, (a, b) in enumerate(zip(g0, g1)):
        # (bias gradients: column sums of bf16-ROUNDED du — 8192 terms with an independent relative rounding of up to 2^-9 each — against
        # sums of the fp32 du: a random walk of ~1e-3 of the sum's own scale, a few times that at the worst of 3456 columns)
        assert float((a - b).abs().max()) <= (6e-3 if k % 3 == 2 else 1e-4) * sc(a), k
    # the FUSED form (round 5, dlrm_gemm_bf16_cross: x_{l+1} = fma(x0, u, xl) in the epilogue of the second product, u kept as bf16 only): the
    # forward is the same operations in the same order -> BIT-identical output; the backward reads the bf16 copy of u for dx0 += g * u, one
    # more bf16 rounding (2^-9 relative per term) on a factor that is itself the result of bf16-operand products -> dx0 within 4e-3 of its
    # scale, and through g_l = g + dv.V + ... every parameter gradient of the layers below within the bf16 class
    assert torch.equal(o1, o2), float((o1 - o2).abs().max())
    assert float((d1 - d2).abs().max()) <= 4e-3 * sc(d1)
    for k, (a, b) in enumerate(zip(g1, g2)):
        assert float((a - b).abs().max()) <= (6e-3 if k % 3 == 2 else 4e-3) * sc(a), k


def test_dlrm_dcn_model_trains_like_a_torch_composition():
    """torchrec_variant.DLRM_DCN (dense arch + pooled embeddings -> DCN-v2 cross network -> over arch -> logits, BCEWithLogits)
    for 2 SGD steps against the same model composed of torch CPU operators with autograd (torchrec itself is absent: UNPINNED)."""
    import torch.nn.functional as Fn
    from dlrm_amd.optim import FusedSGD
    from dlrm_amd.torchrec_variant import DLRM_DCN
    device = torch.device("cuda:0")
    rng = np.random.default_rng(12)
    D, rows, B, hot = 16, [40, 9, 300], 48, [2, 1, 4]
    np.random.seed(2)
    model = DLRM_DCN(rows, D, 13, [24, D], [32, 1], dcn_num_layers=2, dcn_low_rank_dim=8)
    with torch.no_grad():
        for b_ in model.crossnet.bias:
            b_.copy_(torch.from_numpy((rng.standard_normal(b_.shape) * 0.1).astype(np.float32)))
    ref = {k: v.detach().clone().requires_grad_(True) for k, v in model.state_dict().items()}
    model = model.to(device)
    opt = FusedSGD(model.parameters(), lr=0.2)
    ropt = torch.optim.SGD(list(ref.values()), lr=0.2)
    for s in range(2):
        X = torch.from_numpy(rng.random((B, 13)).astype(np.float32))
        idx = [torch.from_numpy(rng.integers(0, n, size=B * h)) for n, h in zip(rows, hot)]
        off = [torch.arange(B) * h for h in hot]
        T = torch.from_numpy(rng.integers(0, 2, size=(B, 1)).astype(np.float32))
        logits = model(X.to(device), [o.to(device) for o in off], [i.to(device) for i in idx])
        E = model.loss_fn(logits, T.to(device))
        x = X
        for i in range(2):
            x = torch.relu(Fn.linear(x, ref[f"bot_l.{2 * i}.weight"], ref[f"bot_l.{2 * i}.bias"]))
        ly = [Fn.embedding_bag(idx[k], ref[f"emb_l.{k}.weight"], off[k], mode="sum", sparse=False) for k in range(3)]
        x0 = torch.cat([x] + ly, dim=1)
        xl = x0
        for l in range(2):
            xl = x0 * (Fn.linear(Fn.linear(xl, ref[f"crossnet.V_kernels.{l}"]), ref[f"crossnet.W_kernels.{l}"]) + ref[f"crossnet.bias.{l}"]) + xl
        z = torch.relu(Fn.linear(xl, ref["top_l.0.weight"], ref["top_l.0.bias"]))
        rl = Fn.linear(z, ref["top_l.2.weight"], ref["top_l.2.bias"])
        RE = Fn.binary_cross_entropy_with_logits(rl, T)
        np.testing.assert_allclose(logits.detach().cpu().numpy(), rl.detach().numpy(), rtol=5e-5, atol=5e-6)
        assert abs(float(E.detach()) - float(RE)) <= 1e-5 * abs(float(RE))
        opt.zero_grad(); E.backward(); opt.step()
        ropt.zero_grad(); RE.backward(); ropt.step()
    sd = model.state_dict()
    for k, v in ref.items():
        np.testing.assert_allclose(sd[k].cpu().numpy(), v.detach().numpy(), rtol=2e-4, atol=1e-5, err_msg=k)


@pytest.mark.parametrize("towers", [False, True], ids=["per-layer", "towers"])
def test_kernel_timers_runs_partition_the_step(towers, monkeypatch):
    """bench.py's per-kernel timers (ops.KernelTimers): one HIP event per change of launch category.  The category runs of an
    instrumented step must account for every launch (call counts) and add up to the step's GPU time (they share their boundary events).
    Small batches: one call per tower and direction (csrc/tower.hip) instead of one per layer."""
    import dlrm_amd
    from dlrm_amd import functional, ops
    from dlrm_amd.optim import FusedSGD
    monkeypatch.setattr(functional, "TOWER_ROWS", 4096 if towers else 0)
    dev = torch.device("cuda:0")
    np.random.seed(1)
    m = dlrm_amd.DLRM_Net(16, np.asarray([50, 60, 70]), np.asarray([13, 32, 16]), np.asarray([22, 32, 1]), "dot", sigmoid_top=1,
                          loss_function="bce").to(dev)
    opt = FusedSGD(m.parameters(), lr=0.1)
    B = 512
    X = torch.rand(B, 13, device=dev)
    off = torch.arange(B, device=dev).repeat(3, 1)
    idx = torch.randint(0, 50, (3, B), device=dev)
    T = torch.rand(B, 1, device=dev).round()

    def step():
        E = m.loss_fn(m(X, off, idx), T)
        opt.zero_grad()
        E.backward()
        opt.step()

    for _ in range(3):
        step()
    torch.cuda.synchronize()
    ops.timers = ops.KernelTimers()
    try:
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        ops.timers.enabled = True
        step()
        ops.timers.enabled = False          # closes the last run
        b.record()
        step()                              # not instrumented: nothing may be added
        s = ops.timers.summary()
    finally:
        ops.timers = None
    wall = a.elapsed_time(b)
    if towers:          # (the forward stays per layer by default: functional.TOWER_FWD)
        assert s["linear_fwd"]["calls"] == 4 and s["linear_bwd_weight"]["calls"] == 2 and s["linear_bwd_data"]["calls"] == 2
    else:
        # (the 32 -> 1 head's backward is ONE call — dlrm_linear_head_bwd, counted with the weight gradients — instead of act_bwd + weight + data gradient)
        assert s["linear_fwd"]["calls"] == 4 and s["linear_bwd_weight"]["calls"] == 4 and s["linear_bwd_data"]["calls"] == 2
        assert "act_bwd" not in s or s["act_bwd"]["calls"] <= 1
    assert s["emb_fwd"]["calls"] == 1 and s["emb_bwd_sgd"]["calls"] == 1 and s["interact_fwd"]["calls"] == 1
    total = sum(v["total_ms"] for v in s.values())
    assert 0.5 * wall <= total <= 1.05 * wall, (total, wall, s)
