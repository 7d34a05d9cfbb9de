"""Loader of the full-batch golden fixtures tests/golden/terabyte_b65536*.npz (BASELINE.json configs[2] shapes):
  terabyte_b65536        rows capped at 2000  (cache-resident tables, ~33 lookups per row: the duplicate-heavy regime)
  terabyte_b65536_cap4m  rows capped at 4 M   (seven 2 GB tables, 22-bit row keys, a row is looked up 0-3 times per batch: the
                                               HBM-resident regime of the benchmark; regenerating + hashing its 14.5 GB of initial
                                               tables takes ~2 min of host time and 16 GB of host RAM)

TEST / MEASUREMENT INFRASTRUCTURE (imported by tests/ and by bench.py's parity check); numpy only, no oracle import.

The fixture was produced by the live reference (oracle/make_golden.py capture_terabyte).  To keep it small it does not
store the initial parameters or the input batches: both are pure functions of numpy's legacy global RandomState, whose
stream is frozen by numpy's compatibility policy.  This module regenerates them with a vectorised restatement of
  * DLRM_Net.create_emb / create_mlp draws        dlrm_s_pytorch.py:222-228, 280-284 (tables, bottom tower, top tower)
  * generate_dist_input_batch, one fixed lookup   dlrm_data_pytorch.py:899-960 (X = rand(n, m_den); per table, per
    sample: r = random(1), index = round(r * (size - 1)))
  * generate_random_output_batch                   dlrm_data_pytorch.py:835-846 (round(rand(n, 1)))
and checks the SHA-256 digest of EVERY regenerated array against the digest of the array the reference actually used —
a mismatch raises, so a test can never silently run on different data than the golden losses belong to.
"""
from __future__ import annotations

import hashlib
import json
import os
from types import SimpleNamespace

import numpy as np

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def _sha(a) -> str:
    return hashlib.sha256(np.ascontiguousarray(a).tobytes()).hexdigest()[:32]


def regen_init(meta) -> dict:
    """Initial parameters (state_dict names) from numpy's global-RNG stream seeded like the reference run."""
    rs = np.random.RandomState(meta["seed"])
    m = meta["m_spa"]
    p = {}
    for k, n in enumerate(meta["ln_emb"]):
        b = np.sqrt(1 / n)
        p[f"emb_l.{k}.weight"] = rs.uniform(low=-b, high=b, size=(n, m)).astype(np.float32)
    for name, ln in (("bot_l", meta["ln_bot"]), ("top_l", meta["ln_top"])):
        for i in range(len(ln) - 1):
            n_in, n_out = ln[i], ln[i + 1]
            p[f"{name}.{2 * i}.weight"] = rs.normal(0.0, np.sqrt(2 / (n_out + n_in)), size=(n_out, n_in)).astype(np.float32)
            p[f"{name}.{2 * i}.bias"] = rs.normal(0.0, np.sqrt(1 / n_out), size=n_out).astype(np.float32)
    return p, rs


def regen_batches(meta, rs) -> list:
    """[(X [B,13] f32, off [T,B] i64, idx [T,B] i64, target [B,1] f32)] * steps — one lookup per bag."""
    assert meta["num_idx"] == 1 and meta["fixed"]
    B, T = meta["B"], len(meta["ln_emb"])
    out = []
    for _ in range(meta["steps"]):
        X = rs.rand(B, meta["ln_bot"][0]).astype(np.float32)
        idx = np.empty((T, B), dtype=np.int64)
        for t, size in enumerate(meta["ln_emb"]):
            idx[t] = np.round(rs.random_sample(B) * (size - 1)).astype(np.int64)
        off = np.tile(np.arange(B, dtype=np.int64), (T, 1))
        tgt = np.round(rs.rand(B, 1).astype(np.float32)).astype(np.float32)
        out.append((X, off, idx, tgt))
    return out


def load(name: str = "terabyte_b65536", verify: bool = True):
    z = np.load(os.path.join(GOLDEN, name + ".npz"))
    d = {k: z[k] for k in z.files}
    meta = json.loads(bytes(d.pop("meta")).decode())
    init, rs = regen_init(meta)
    # the generator state right after the reference built its model (also stored: guards the restatement above)
    st = rs.get_state()
    if verify:
        assert np.array_equal(np.asarray(st[1], dtype=np.uint32), d["rng_after_init.keys"]) and \
            [st[2], st[3]] == d["rng_after_init.pos_gauss"].tolist(), "golden_tb: RNG state after init differs from the reference's"
    batches = regen_batches(meta, rs)
    if verify:
        dig = meta["digests"]
        for k, v in init.items():
            if _sha(v) != dig[f"init.{k}"]:
                raise AssertionError(f"golden_tb: regenerated init.{k} differs from the reference's array")
        for s, (X, off, idx, tgt) in enumerate(batches):
            for tag, a in (("X", X), ("T", tgt), ("idx", idx), ("off", off)):
                if _sha(a) != dig[f"s{s}.{tag}"]:
                    raise AssertionError(f"golden_tb: regenerated s{s}.{tag} differs from the reference's array")
    return SimpleNamespace(meta=meta, d=d, init=init, batches=batches, losses=d["losses"])


def run_on_gpu(device, arith="f32", mode=None, steps=None, check=True, name="terabyte_b65536", overlap=False, fuse=False, update_in_backward=False):
    """Train dlrm_amd.DLRM_Net in bench.py's configuration — stacked [T, B] int64 inputs, UPD_SORTED fused update,
    FusedSGD, and (selected by the shapes) 256-row GEMM tiles and the DMA interaction kernels — on the full-batch golden
    fixture of the live reference.  Returns the per-step relative loss errors; with check=True also asserts predictions
    (rtol 2e-5), three step-0 gradients, final MLP parameters and final table rows / column sums (rtol 1e-4).
    Used by tests/test_gpu_model.py and by bench.py's `parity_check` (no oracle import: fixture + product path only)."""
    import torch
    import dlrm_amd
    from dlrm_amd import ops
    from dlrm_amd.optim import FusedSGD
    fx = load(name)
    meta, d = fx.meta, fx.d
    np.random.seed(0)
    # tables are allocated on the device (no numpy draw of values that are overwritten right below: that alone would cost
    # minutes for the 4 M-row fixture), then every parameter is set to the reference's initial value
    dlrm_amd.set_embedding_init(device)
    try:
        model = dlrm_amd.DLRM_Net(meta["m_spa"], np.asarray(meta["ln_emb"]), np.asarray(meta["ln_bot"]), np.asarray(meta["ln_top"]),
                                  arch_interaction_op="dot", arch_interaction_itself=meta["itself"], sigmoid_bot=-1,
                                  sigmoid_top=meta["sigmoid_top"], loss_function=meta["loss"])
    finally:
        dlrm_amd.set_embedding_init(None)
    model = model.to(device)
    with torch.no_grad():
        sd = model.state_dict()
        assert set(sd.keys()) == set(fx.init.keys())
        for k in list(fx.init.keys()):
            sd[k].copy_(torch.from_numpy(fx.init.pop(k)))          # (popped: the host copy of a 2 GB table is released at once)
    model.emb_update_mode = ops.UPD_SORTED if mode is None else mode
    model.set_mlp_arith(arith)
    model.fuse_emb_interact = bool(fuse)      # lookups fetched by the interaction kernels (the product default) instead of two kernels
    model.overlap_streams = bool(overlap)     # embedding kernels on a side stream beside the bottom-MLP GEMMs (bench default)
    # from the second step on (the optimizer is known after its first step) the fused backward takes the SGD step of single-lookup rows (ABI 17)
    model.update_in_backward = bool(update_in_backward)
    opt = FusedSGD(model.parameters(), lr=meta["lr"])
    rel = []
    close = np.testing.assert_allclose
    for s, (X, off, idx, tgt) in enumerate(fx.batches[:steps]):
        Z = model(torch.from_numpy(X).to(device), torch.from_numpy(off).to(device), torch.from_numpy(idx).to(device))
        E = model.loss_fn(Z, torch.from_numpy(tgt).to(device))
        rel.append(abs(float(E.detach()) - fx.losses[s]) / abs(fx.losses[s]))
        if check:
            close(Z.detach().cpu().numpy(), d[f"s{s}.Z"], rtol=2e-5, atol=1e-6)
        opt.zero_grad()
        E.backward()
        if s == 0 and check:
            # batch-summed gradients of ~1e-6 magnitude: 65536 signed terms cancel, and a ReLU output within rounding of
            # zero may fall on either side of the threshold (one whole term appears / disappears) — compared on the scale
            # of the gradient tensor, not element by element
            for g, ref in ((model.bot_l[0].bias.grad, d["s0.bot0_bias_grad"]), (model.top_l[8].weight.grad, d["s0.top8_weight_grad"]),
                           (model.top_l[0].bias.grad, d["s0.top0_bias_grad"])):
                close(g.cpu().numpy(), ref, rtol=2e-4, atol=1e-2 * float(np.abs(ref).max()))
        opt.step()
    if check and (steps is None or steps >= meta["steps"]):
        sd = model.state_dict()
        for k, v in d.items():
            if k.startswith("final."):
                close(sd[k[6:]].cpu().numpy(), v, rtol=1e-4, atol=5e-6, err_msg=k)
            elif k.startswith("final_head."):
                n = k[len("final_head."):]
                close(sd[n][:48].cpu().numpy(), v, rtol=1e-4, atol=5e-6, err_msg=k)
                close(sd[n][-48:].cpu().numpy(), d["final_tail." + n], rtol=1e-4, atol=5e-6, err_msg=k)
                # every row of the table, through its fp64 column sums (a lost or doubled update anywhere shows up)
                close(sd[n].double().sum(0).cpu().numpy(), d["final_colsum." + n], rtol=1e-4, atol=1e-4, err_msg=k)
                if "final_touched." + n in d:
                    # rows that WERE updated: the ones the first 64 samples of step 0 looked up (head / tail rows of a 4 M-row
                    # table are almost never touched)
                    t = int(n.split(".")[1])
                    rows_ = torch.from_numpy(fx.batches[0][2][t][:64]).to(device)
                    close(sd[n][rows_].cpu().numpy(), d["final_touched." + n], rtol=1e-4, atol=5e-6, err_msg="touched " + k)
    ops.check_index_errors(sync=True)
    del model, opt
    return rel
