#!/usr/bin/env python3
"""Golden vectors for bench.py's `--workload mlperf_v2_multihot` configuration (BASELINE.json configs[4]).  TEST INFRASTRUCTURE.

    PYTHONDONTWRITEBYTECODE=1 python oracle/make_golden_v2.py [dot|dcn|all]      ->  tests/golden/mlperf_v2_{dot,dcn}_b65536.npz

What is pinned to what:
  * the OPTIMIZER is the live reference: `optim/rwsadagrad.py` RWSAdagrad imported from $DLRM_REFERENCE (row-wise sparse Adagrad
    on the tables' sparse gradients, dense Adagrad on the towers), lr 0.005, eps 1e-8 (torchrec_dlrm/README.MD:177-194);
  * the MODEL is a composition of torch CPU operators with autograd restating torchrec's published DLRM / DLRM_DCN (torchrec is a
    third-party dependency absent from the reference tree: parity of the model semantics stays UNPINNED) — EmbeddingBag(sum,
    sparse) tables, ReLU towers, triu dot interaction or the DCN-v2 low-rank cross network (3 layers, rank 512), bare last Linear,
    BCEWithLogitsLoss (torchrec_dlrm/dlrm_main.py:598-653);
  * the INPUTS are the benchmark's: 214 int32 lookups per sample expanded through seed-0 uniform lookup tables (oracle.
    philox_multihot_table + multihot_expand, the restatement of torchrec_dlrm/multi_hot.py pinned in multihot_tables.npz) from
    1-hot ids / dense features / labels of the device generator's Philox stream (oracle.philox_*), seed 727 like bench.py;
  * sizes: B = 65536, 26 tables of the MLPerf-v2 row counts capped at ROW_CAP, D = 128, towers 13-512-256-128 / 1024-1024-512-256-1.

Kept small: inputs and initial parameters are NOT stored (SHA-256 digests are; tests/golden_v2.py regenerates them through the
product's own generators / constructors and refuses to compare on a mismatch).  Stored: the 3 losses, the logits of every step,
and per-tensor summaries of the final parameters and optimizer state (fp64 sum, |sum|, 4096 strided samples).
"""
from __future__ import annotations

import hashlib
import json
import os
import sys
import time

os.environ.setdefault("PYTHONDONTWRITEBYTECODE", "1")
sys.dont_write_bytecode = True

import numpy as np
import torch
import torch.nn.functional as Fn

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
REF = os.environ.get("DLRM_REFERENCE", "/root/reference")
OUT = os.path.join(ROOT, "tests", "golden")

from oracle import oracle as O  # noqa: E402

MLPERF_V2_ROWS = [40000000, 39060, 17295, 7424, 20265, 3, 7122, 1543, 63, 40000000, 3067956, 405282, 10, 2209, 11938, 155, 4,
                  976, 14, 40000000, 40000000, 40000000, 590152, 12973, 108, 36]             # torchrec_dlrm/README.MD:45
MLPERF_V2_HOT = [3, 2, 1, 2, 6, 1, 1, 1, 1, 7, 3, 8, 1, 6, 9, 5, 1, 1, 1, 12, 100, 27, 10, 3, 1, 1]   # README.MD:159
ROW_CAP = 200000
SEED_DATA, SEED_INIT, SEED_MULTIHOT = 727, 123, 0


def sha(a) -> str:
    return hashlib.sha256(np.ascontiguousarray(a).tobytes()).hexdigest()[:32]


def gen_seed(seed: int, batch_no: int, stream_id: int) -> int:
    """dlrm_amd/datagen.py UniformBatchGenerator._seed (pure arithmetic; the class itself needs a GPU)."""
    z = (seed * 0x9E3779B97F4A7C15 + batch_no * 0xBF58476D1CE4E5B9 + stream_id * 0x94D049BB133111EB) & (2 ** 64 - 1)
    z ^= z >> 31
    return z & (2 ** 64 - 1)


def summary(a: np.ndarray) -> np.ndarray:
    """[sum, sum|.|, n] in fp64 followed by up to 4096 strided samples — enough to catch a wrong update anywhere in a tensor that is
    too large to store (the DCN kernels alone are 42 MB)."""
    f = np.asarray(a, dtype=np.float64).reshape(-1)
    step = max(f.size // 4096, 1)
    return np.concatenate([[f.sum(), np.abs(f).sum(), f.size], f[::step][:4096]])


def make_inputs(rows, hot, B, steps):
    """[(X [B,13] f32, ids [T,B] i32, values i32 [B*sum(hot)], off_local [T,B] i32, labels [B,1] f32)] per step, exactly what
    bench.make_batches builds on the device: UniformBatchGenerator(13, rows, 1, True, round_targets=True, seed=727, int32).batch(B, k)
    then Multihot(hot, rows, B, "uniform", seed=0).to_model_inputs(ids)."""
    tabs = [O.philox_multihot_table(t, n, h, 0, SEED_MULTIHOT) for t, (n, h) in enumerate(zip(rows, hot))]
    out = []
    for k in range(steps):
        X = O.philox_dense(B * 13, gen_seed(SEED_DATA, k, 1)).reshape(B, 13)
        lab = O.philox_dense(B, gen_seed(SEED_DATA, k, 2), round_values=True).reshape(B, 1)
        ids = np.stack([O.philox_onehot(t, n, B, gen_seed(SEED_DATA, k, 16)) for t, n in enumerate(rows)]).astype(np.int32)
        values, _ = O.multihot_expand(ids, tabs)
        off_l = np.stack([np.arange(B, dtype=np.int32) * h for h in hot])
        out.append((X, ids, values, off_l, lab))
    return out


def train(interaction, init, batches, rows, hot, lr, eps, acc0, dtype, rwsadagrad, D=128, nbot=3, ntop=5):
    """3 steps of the torch-operator model + the reference's RWSAdagrad; returns (losses, logits per step, final params, optimizer)."""
    B, T = batches[0][0].shape[0], len(rows)
    p = {k: torch.from_numpy(v.copy()).to(dtype).requires_grad_(True) for k, v in init.items()}
    opt = rwsadagrad.RWSAdagrad(list(p.values()), lr=lr, eps=eps, initial_accumulator_value=acc0)
    iu = torch.triu_indices(T + 1, T + 1, offset=1)
    losses, logits = [], []
    for s, (X, ids, values, off_l, lab) in enumerate(batches):
        t0 = time.time()
        x = torch.from_numpy(X).to(dtype)
        for i in range(nbot):
            x = torch.relu(Fn.linear(x, p[f"bot_l.{2 * i}.weight"], p[f"bot_l.{2 * i}.bias"]))
        ly, o = [], 0
        for t, h in enumerate(hot):
            idx = torch.from_numpy(values[o:o + B * h].astype(np.int64))
            o += B * h
            ly.append(Fn.embedding_bag(idx, p[f"emb_l.{t}.weight"], torch.from_numpy(off_l[t].astype(np.int64)), mode="sum", sparse=True))
        if interaction == "dcn":
            x0 = torch.cat([x] + ly, dim=1)
            z = x0
            for l in range(3):
                z = x0 * (Fn.linear(Fn.linear(z, p[f"crossnet.V_kernels.{l}"]), p[f"crossnet.W_kernels.{l}"]) + p[f"crossnet.bias.{l}"]) + z
        else:
            feat = torch.cat([x] + ly, dim=1).view(B, T + 1, D)
            Z = torch.bmm(feat, feat.transpose(1, 2))
            z = torch.cat([x, Z[:, iu[0], iu[1]]], dim=1)
        for i in range(ntop):
            z = Fn.linear(z, p[f"top_l.{2 * i}.weight"], p[f"top_l.{2 * i}.bias"])
            if i + 1 < ntop:
                z = torch.relu(z)
        E = Fn.binary_cross_entropy_with_logits(z, torch.from_numpy(lab).to(dtype))
        logits.append(z.detach().numpy().reshape(-1).astype(np.float32))
        losses.append(float(E.detach()))
        opt.zero_grad()
        E.backward()
        if dtype != torch.float32:                       # the reference keeps its optimizer state in fp32 whatever the parameters are
            for v in p.values():
                v.grad = v.grad.to(dtype)
        opt.step()
        print(f"  step {s}: loss {losses[-1]:.8f}  ({time.time() - t0:.0f} s)")
    return losses, logits, p, opt


def capture(interaction: str, B: int = 65536, steps: int = 3, lr: float = 0.005, eps: float = 1e-8):
    """Three runs on the same inputs and initial parameters:
      bench        lr 0.005, eps 1e-8, zero initial accumulator — the benchmark's hyper-parameters (FBGEMM hard-codes the zero, dlrm_main.py:
                   641-645).  Step 1 of Adagrad from a zero accumulator is a SIGN descent (g / (|g| + eps)): gradients that cancel to
                   rounding noise pick their sign by the arithmetic, every such weight lands 2 * lr apart.  How far two CORRECT arithmetics
                   land from each other is measured here —
      bench, fp64  the same composition in float64 -> `cond_rel`: |loss_fp32 - loss_fp64| / loss per step (what no fp32 implementation
                   can be held below at steps 1-2) —
      conditioned  initial_accumulator_value = 1.0, everything else equal: the update is smooth (~ lr * g), all 3 steps hold the 1e-5 bar."""
    sys.path.insert(0, os.path.join(REF, "optim"))
    import rwsadagrad                                           # the reference's own optimizer class
    from dlrm_amd.torchrec_variant import DLRM, DLRM_DCN        # host-side construction only: the RNG order of the product's initialisation
    rows = [min(n, ROW_CAP) for n in MLPERF_V2_ROWS]
    hot = [min(h, n) for h, n in zip(MLPERF_V2_HOT, rows)]
    D, bot, top = 128, [512, 256, 128], [1024, 1024, 512, 256, 1]
    np.random.seed(SEED_INIT)
    if interaction == "dcn":
        shell = DLRM_DCN(rows, D, 13, bot, top, dcn_num_layers=3, dcn_low_rank_dim=512)
    else:
        shell = DLRM(rows, D, 13, bot, top)
    init = {k: v.detach().numpy().copy() for k, v in shell.state_dict().items()}
    del shell
    digests = {f"init.{k}": sha(v) for k, v in init.items()}
    t0 = time.time()
    batches = make_inputs(rows, hot, B, steps)
    print(f"inputs: {time.time() - t0:.0f} s, {batches[0][2].size / B:.0f} lookups/sample")
    for s, (X, ids, values, off_l, lab) in enumerate(batches):
        for tag, a in (("X", X), ("ids", ids), ("values", values), ("off", off_l), ("labels", lab)):
            digests[f"s{s}.{tag}"] = sha(a)
    out = {}
    print("bench (fp32):")
    losses, logits, _, _ = train(interaction, init, batches, rows, hot, lr, eps, 0.0, torch.float32, rwsadagrad)
    print("bench (fp64):")
    l64, g64, _, _ = train(interaction, init, batches, rows, hot, lr, eps, 0.0, torch.float64, rwsadagrad)
    print("conditioned (fp32, initial accumulator 1.0):")
    lc, gc, p, opt = train(interaction, init, batches, rows, hot, lr, eps, 1.0, torch.float32, rwsadagrad)
    for s in range(steps):
        out[f"s{s}.logits"] = logits[s]
        out[f"cond.s{s}.logits"] = gc[s]
    out["losses"] = np.asarray(losses, dtype=np.float64)
    out["losses_fp64"] = np.asarray(l64, dtype=np.float64)
    out["cond_rel"] = np.abs(out["losses"] - out["losses_fp64"]) / np.abs(out["losses_fp64"])
    out["cond_logit_abs"] = np.asarray([float(np.abs(logits[s] - g64[s]).max()) for s in range(steps)])
    out["cond.losses"] = np.asarray(lc, dtype=np.float64)
    for k, v in p.items():
        out[f"final.{k}"] = summary(v.detach().numpy())
        st = opt.state[v]
        if "momentum" in st:
            out[f"state_rowwise.{k}"] = summary(st["momentum"].numpy())
        if "sum" in st:
            out[f"state_sum.{k}"] = summary(st["sum"].numpy())
    meta = dict(name=f"mlperf_v2_{interaction}_b{B}", interaction=interaction, rows=rows, hot=hot, D=D, bot=[13] + bot, top=top, B=B,
                steps=steps, lr=lr, eps=eps, row_cap=ROW_CAP, seed_data=SEED_DATA, seed_init=SEED_INIT, seed_multihot=SEED_MULTIHOT,
                conditioned_initial_accumulator_value=1.0, digests=digests, torch=torch.__version__, numpy=np.__version__,
                pinned="optimizer = the reference's optim/rwsadagrad.py RWSAdagrad; model = torch-operator restatement of torchrec's "
                       "published DLRM / DLRM_DCN (third-party, absent: UNPINNED)")
    out["meta"] = np.frombuffer(json.dumps(meta).encode(), dtype=np.uint8)
    os.makedirs(OUT, exist_ok=True)
    np.savez_compressed(os.path.join(OUT, meta["name"] + ".npz"), **out)
    print(meta["name"], "bench", losses, "cond_rel", out["cond_rel"].tolist(), "conditioned", lc)


if __name__ == "__main__":
    which = sys.argv[1] if len(sys.argv) > 1 else "all"
    for kind in ("dot", "dcn"):
        if which in ("all", kind):
            capture(kind)
