"""Loader / runner of the golden fixtures of bench.py's `--workload mlperf_v2_multihot` configuration (BASELINE.json configs[4]):
tests/golden/mlperf_v2_{dot,dcn}_b65536.npz, produced by oracle/make_golden_v2.py.

TEST / MEASUREMENT INFRASTRUCTURE (imported by tests/ and by bench.py's parity check); numpy + the fixture + the PRODUCT path only, no
oracle import.  The fixture stores no inputs and no initial parameters: `run_on_gpu` builds them exactly as bench.py does — the model
through the product's constructor under the fixture's numpy seed, the batches through dlrm_amd.datagen.UniformBatchGenerator (Philox,
seed 727, int32 ids) + dlrm_amd.multihot.Multihot (seed-0 uniform lookup tables, 214 lookups per sample) — and compares the SHA-256 of
every array with the digest of what the oracle side used; a mismatch raises, so losses are never compared on different data.

What the fixture is pinned to: the optimizer is the live reference's `optim/rwsadagrad.py`; the model is a torch-operator restatement
of torchrec's published DLRM / DLRM_DCN (third-party, absent from the reference tree: UNPINNED — see the generator's header).
"""
from __future__ import annotations

import hashlib
import json
import os

import numpy as np

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")

# Tolerances.  f32: north_star's 1e-5 relative on the loss.  bf16 (the benchmark's arithmetic: bf16 MLP operands, fp32 accumulation
# and fp32 master weights) has no reference number — torchrec's bf16 path is not in the tree — so its bar is MEASURED against this
# fp32 fixture on the MI355X and stated here; the test fails if the bf16 path drifts outside it.
LOSS_RTOL = {"f32": 1e-5, "bf16x6": 1e-5, "bf16": 2e-3}
LOGIT_ATOL = {"f32": 5e-5, "bf16x6": 5e-5, "bf16": 5e-2}


def _sha(a) -> str:
    return hashlib.sha256(np.ascontiguousarray(a).tobytes()).hexdigest()[:32]


def available(interaction: str) -> bool:
    return os.path.isfile(os.path.join(GOLDEN, f"mlperf_v2_{interaction}_b65536.npz"))


def load(interaction: str):
    z = np.load(os.path.join(GOLDEN, f"mlperf_v2_{interaction}_b65536.npz"))
    d = {k: z[k] for k in z.files}
    meta = json.loads(bytes(d.pop("meta")).decode())
    return meta, d


def summary(t) -> np.ndarray:
    """the generator's per-tensor summary ([sum, sum|.|, n] + strided samples), computed on the device in fp64"""
    import torch
    f = t.detach().reshape(-1).double()
    step = max(f.numel() // 4096, 1)
    head = torch.stack([f.sum(), f.abs().sum(), torch.tensor(float(f.numel()), dtype=torch.float64, device=f.device)])
    return torch.cat([head, f[::step][:4096]]).cpu().numpy()


def run_on_gpu(device, interaction: str = "dot", arith: str = "f32", check_params: bool = True, variants=("bench", "conditioned")):
    """3 training steps of bench.py's mlperf_v2_multihot configuration (row-capped tables) on the GPU against the fixture, for
      "bench"        the benchmark's hyper-parameters (row-wise / dense Adagrad lr 0.005, eps 1e-8, ZERO initial accumulator).  Step 0 must
                     hold the arithmetic's bar; steps 1-2 are a sign descent out of the zero accumulator, where the fixture's own fp32 and
                     fp64 runs differ by `cond_rel` (5.9e-5 / 7.1e-6 for the dot model): bar = max(arith bar, 4 x cond_rel[step]);
      "conditioned"  initial_accumulator_value = 1.0, everything else equal: all 3 steps hold the arithmetic's bar, and the final
                     parameters / optimizer state are compared through their per-tensor summaries.
    Returns a dict per variant; raises AssertionError when a bar is missed."""
    import torch
    from dlrm_amd.datagen import UniformBatchGenerator
    from dlrm_amd.multihot import Multihot
    from dlrm_amd.optim import FusedRWSAdagrad
    from dlrm_amd.torchrec_variant import DLRM, DLRM_DCN
    from dlrm_amd import ops
    meta, d = load(interaction)
    rows, hot, D, B = meta["rows"], meta["hot"], meta["D"], meta["B"]
    dig = meta["digests"]
    gen = UniformBatchGenerator(13, rows, 1, True, round_targets=True, seed=meta["seed_data"], device=device, index_dtype=torch.int32)
    mh = Multihot(hot, rows, B, dist_type="uniform", device=device, seed=meta["seed_multihot"])
    batches = []
    for s in range(meta["steps"]):
        X, lS_o, lS_i, T = gen.batch(B, batch_no=s)
        ids = torch.stack(lS_i)
        off, idx = mh.to_model_inputs(ids)
        values = torch.cat([v.reshape(-1) for v in idx])
        for tag, a in (("X", X), ("ids", ids), ("values", values), ("off", torch.stack(off)), ("labels", T)):
            if _sha(a.cpu().numpy()) != dig[f"s{s}.{tag}"]:
                raise AssertionError(f"golden_v2: step {s} input `{tag}` generated on the device differs from the fixture's")
        batches.append((X, off, idx, T))
    result = {}
    for variant in variants:
        pre = "" if variant == "bench" else "cond."
        acc0 = 0.0 if variant == "bench" else float(meta["conditioned_initial_accumulator_value"])
        np.random.seed(meta["seed_init"])
        import dlrm_amd.dlrm_net as _net
        saved_init = _net._EMB_INIT_DEVICE
        _net.set_embedding_init(None)            # the fixture's tables are numpy draws (bench.py allocates its 104 GB of tables on the device)
        try:
            if interaction == "dcn":
                model = DLRM_DCN(rows, D, meta["bot"][0], meta["bot"][1:], meta["top"], dcn_num_layers=3, dcn_low_rank_dim=512)
            else:
                model = DLRM(rows, D, meta["bot"][0], meta["bot"][1:], meta["top"])
        finally:
            _net.set_embedding_init(saved_init)
        for k, v in model.state_dict().items():
            if _sha(v.cpu().numpy()) != dig[f"init.{k}"]:
                raise AssertionError(f"golden_v2: initial {k} differs from the fixture's")
        model = model.to(device)
        model.set_mlp_arith(arith)
        opt = FusedRWSAdagrad(model.parameters(), lr=meta["lr"], eps=meta["eps"], initial_accumulator_value=acc0)
        rel, logit_err, bars = [], [], []
        for s, (X, off, idx, T) in enumerate(batches):
            logits = model(X, off, idx)
            E = model.loss_fn(logits, T)
            want = float(d[pre + "losses"][s])
            rel.append(abs(float(E.detach()) - want) / abs(want))
            logit_err.append(float(np.abs(logits.detach().cpu().numpy().reshape(-1) - d[f"{pre}s{s}.logits"]).max()))
            bars.append(max(LOSS_RTOL[arith], 4.0 * float(d["cond_rel"][s])) if variant == "bench" else LOSS_RTOL[arith])
            opt.zero_grad()
            E.backward()
            opt.step()
        ops.check_index_errors(sync=True)
        out = {"rel_err_per_step": rel, "loss_bar_per_step": bars, "max_logit_abs_err_per_step": logit_err,
               "reference_losses": [float(v) for v in d[pre + "losses"]]}
        for s in range(len(rel)):
            assert rel[s] <= bars[s], (variant, "loss", arith, s, rel, bars)
        if variant == "conditioned":
            assert max(logit_err) <= LOGIT_ATOL[arith], ("logits", arith, logit_err)
        else:
            assert logit_err[0] <= LOGIT_ATOL[arith], ("logits of step 0", arith, logit_err)
        if variant == "conditioned" and check_params and arith != "bf16":
            sd = model.state_dict()
            worst = 0.0
            for k, v in d.items():
                if not k.startswith("final."):
                    continue
                got = summary(sd[k[6:]])
                # sums: relative to the tensor's absolute sum (a signed sum can cancel to ~0); samples: element-wise
                scale = max(abs(v[1]), 1e-30)
                worst = max(worst, abs(got[0] - v[0]) / scale, abs(got[1] - v[1]) / scale)
                assert abs(got[0] - v[0]) <= 2e-4 * scale and abs(got[1] - v[1]) <= 2e-4 * scale, (k, got[:3], v[:3])
                np.testing.assert_allclose(got[3:], v[3:], rtol=2e-3, atol=2e-5, err_msg=k)
            named = dict(model.named_parameters())
            for k, v in d.items():
                if k.startswith("state_rowwise."):
                    got = summary(opt.state[named[k[len("state_rowwise."):]]]["momentum"])
                    assert abs(got[1] - v[1]) <= 1e-3 * max(abs(v[1]), 1e-30), (k, got[:3], v[:3])
            out["worst_param_sum_rel_err"] = worst
        result[variant] = out
        del model, opt
        torch.cuda.empty_cache()
    return result
