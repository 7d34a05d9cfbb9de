#!/usr/bin/env python3
"""Generate tests/golden/*.npz|json by RUNNING THE REFERENCE (facebookresearch/dlrm at /root/reference)
on CPU in the build container.  TEST INFRASTRUCTURE ONLY — not shipped, not imported by dlrm_amd/.

    PYTHONDONTWRITEBYTECODE=1 python oracle/make_golden.py            # all fixtures
    PYTHONDONTWRITEBYTECODE=1 python oracle/make_golden.py dist       # one group

The reference is imported as a library (its `DLRM_Net`, its data generator, its `extend_distributed`
and its `RWSAdagrad`); `torch.utils.tensorboard` (absent here, imported at dlrm_s_pytorch.py:101) is
stubbed.  Nothing is copied from the reference: the fixtures hold only inputs, initial parameters and
the numbers the reference produced (outputs, losses, updated parameters).  /root/reference does not
exist on the GPU box, which is why the vectors are committed.
"""
from __future__ import annotations

import json
import os
import sys
import tempfile
import types

os.environ.setdefault("PYTHONDONTWRITEBYTECODE", "1")
sys.dont_write_bytecode = True

import numpy as np
import torch

REF = os.environ.get("DLRM_REFERENCE", "/root/reference")
OUT = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden")


def import_reference():
    tb = types.ModuleType("torch.utils.tensorboard")

    class SummaryWriter:  # noqa: D401 - stub
        def __init__(self, *a, **k): pass
        def add_scalar(self, *a, **k): pass
        def close(self): pass
    tb.SummaryWriter = SummaryWriter
    sys.modules["torch.utils.tensorboard"] = tb
    if REF not in sys.path:
        sys.path.insert(0, REF)
    cwd = os.getcwd()
    os.chdir(tempfile.mkdtemp(prefix="dlrm_ref_"))
    try:
        import dlrm_s_pytorch as ref
        import dlrm_data_pytorch as dp
        import extend_distributed as ext
    finally:
        os.chdir(cwd)
    return ref, dp, ext


def gen_batch(dp, m_den, ln_emb, n, num_idx, fixed, round_targets=True):
    X, lS_o, lS_i = dp.generate_dist_input_batch(m_den, ln_emb, n, num_idx, fixed, rand_data_dist="uniform",
                                                 rand_data_min=0, rand_data_max=1, rand_data_mu=-1, rand_data_sigma=1)
    T = dp.generate_random_output_batch(n, 1, round_targets)
    return X, lS_o, lS_i, T


def pack_batches(batches):
    d = {}
    for s, (X, lS_o, lS_i, T) in enumerate(batches):
        d[f"s{s}.X"] = X.numpy().copy()
        d[f"s{s}.T"] = T.numpy().copy()
        for k in range(len(lS_i)):
            d[f"s{s}.off{k}"] = lS_o[k].numpy().astype(np.int64).copy()
            d[f"s{s}.idx{k}"] = lS_i[k].numpy().astype(np.int64).copy()
    return d


def sd_np(model, prefix):
    return {f"{prefix}.{k}": v.detach().cpu().numpy().copy() for k, v in model.state_dict().items()}


def capture_training(ref, dp, name, m_spa, ln_emb, ln_bot, top_tail, B, steps, lr, loss, itself=False,
                     num_idx=10, fixed=False, seed=123, round_targets=True, compact=False, interaction="dot",
                     loss_threshold=0.0, loss_weights=None, weighted_pooling=None, lr_schedule=None):
    """lr_schedule = (num_warmup_steps, decay_start_step, num_decay_steps): the reference's LRPolicyScheduler (dlrm_s_pytorch.py:169-203)
    built on the optimizer as run() does (:1370) and stepped after every optimizer step (:1621); the lr each step was applied with is
    recorded as `lrs`."""
    ln_emb = np.asarray(ln_emb)
    ln_bot = np.asarray(ln_bot)
    F = ln_emb.size + 1
    if interaction == "cat":
        num_int = F * ln_bot[-1]
    else:
        num_int = (F * (F + 1)) // 2 + ln_bot[-1] if itself else (F * (F - 1)) // 2 + ln_bot[-1]
    ln_top = np.asarray([num_int] + list(top_tail))
    np.random.seed(seed)
    torch.manual_seed(seed)
    if loss == "wbce":          # the reference reads the CLI global `args.loss_weights` (dlrm_s_pytorch.py:391)
        ref.args = types.SimpleNamespace(loss_weights=loss_weights, loss_function="wbce")
    model = ref.DLRM_Net(m_spa, ln_emb, ln_bot, ln_top, arch_interaction_op=interaction, arch_interaction_itself=itself,
                         sigmoid_bot=-1, sigmoid_top=ln_top.size - 2, loss_function=loss, loss_threshold=loss_threshold,
                         weighted_pooling=weighted_pooling)
    if weighted_pooling == "learned":       # all-ones weights would hide a wrong gather: start from distinct values
        with torch.no_grad():
            for k, w in enumerate(model.v_W_l):
                w.copy_(torch.tensor(np.random.RandomState(50 + k).uniform(0.5, 1.5, size=w.shape).astype(np.float32)))
    ref.dlrm = model            # loss_fn_wrap uses the module globals `dlrm` and `args` (:148-156)
    out = {}
    out.update(sd_np(model, "init"))
    batches = [gen_batch(dp, int(ln_bot[0]), ln_emb, B, num_idx, fixed, round_targets) for _ in range(steps)]
    out.update(pack_batches(batches))
    opt = torch.optim.SGD(model.parameters(), lr=lr)
    sched = ref.LRPolicyScheduler(opt, *lr_schedule) if lr_schedule is not None else None
    losses, lrs = [], []
    for s, (X, lS_o, lS_i, T) in enumerate(batches):
        lrs.append(float(opt.param_groups[0]["lr"]))
        Z = model(X, lS_o, lS_i)
        E = ref.loss_fn_wrap(Z, T, False, "cpu") if loss == "wbce" else model.loss_fn(Z, T)
        out[f"s{s}.Z"] = Z.detach().numpy().copy()
        losses.append(float(E.detach().numpy()))
        opt.zero_grad()
        E.backward()
        if s == 0:
            g = model.emb_l[0].weight.grad
            out["s0.emb0_grad_indices"] = g._indices().numpy().copy()
            out["s0.emb0_grad_values"] = g._values().numpy().copy()
            out["s0.top0_weight_grad"] = model.top_l[0].weight.grad.numpy().copy()
            out["s0.bot0_bias_grad"] = model.bot_l[0].bias.grad.numpy().copy()
        opt.step()
        if sched is not None:
            sched.step()
        if s == 0 and not compact:
            out.update(sd_np(model, "after1"))
    out.update(sd_np(model, "final"))
    out["losses"] = np.asarray(losses, dtype=np.float64)
    if sched is not None:
        out["lrs"] = np.asarray(lrs, dtype=np.float64)
    meta = dict(name=name, m_spa=int(m_spa), ln_emb=ln_emb.tolist(), ln_bot=ln_bot.tolist(), ln_top=ln_top.tolist(),
                B=B, steps=steps, lr=lr, loss=loss, itself=bool(itself), sigmoid_top=int(ln_top.size - 2),
                interaction=interaction, loss_threshold=float(loss_threshold), weighted_pooling=weighted_pooling,
                loss_ws=None if loss_weights is None else [float(x) for x in loss_weights.split("-")],
                lr_schedule=None if lr_schedule is None else [int(x) for x in lr_schedule],
                torch=torch.__version__, reference="facebookresearch/dlrm @ /root/reference (2025-10-03)")
    out["meta"] = np.frombuffer(json.dumps(meta).encode(), dtype=np.uint8)
    np.savez_compressed(os.path.join(OUT, f"{name}.npz"), **out)
    print(f"{name}: losses {losses}")


def capture_adagrad(ref, dp, name="rwsadagrad_tiny"):
    sys.path.insert(0, os.path.join(REF, "optim"))
    import rwsadagrad as RowWiseSparseAdagrad
    ln_emb, ln_bot, m_spa = np.asarray([40, 7, 3]), np.asarray([5, 8, 4]), 4
    F = 4
    ln_top = np.asarray([F * (F - 1) // 2 + 4, 8, 1])
    np.random.seed(7)
    torch.manual_seed(7)
    model = ref.DLRM_Net(m_spa, ln_emb, ln_bot, ln_top, arch_interaction_op="dot", sigmoid_top=1, loss_function="bce")
    out = {}
    out.update(sd_np(model, "init"))
    batches = [gen_batch(dp, 5, ln_emb, 16, 5, False) for _ in range(3)]
    out.update(pack_batches(batches))
    opt = RowWiseSparseAdagrad.RWSAdagrad(model.parameters(), lr=0.05)
    losses = []
    for s, (X, lS_o, lS_i, T) in enumerate(batches):
        Z = model(X, lS_o, lS_i)
        E = model.loss_fn(Z, T)
        losses.append(float(E.detach().numpy()))
        opt.zero_grad()
        E.backward()
        if s == 0:
            # dense dV of every table, recovered from the MLP-side autograd graph is not exposed; store the
            # coalesced sparse grads instead (what RWSAdagrad consumes)
            for k in range(3):
                g = model.emb_l[k].weight.grad.coalesce()
                out[f"s0.emb{k}_cgrad_indices"] = g._indices().numpy().copy()
                out[f"s0.emb{k}_cgrad_values"] = g._values().numpy().copy()
        opt.step()
        if s == 0:
            out.update(sd_np(model, "after1"))
            for k in range(3):
                out[f"after1.mom{k}"] = opt.state[model.emb_l[k].weight]["momentum"].numpy().copy()
    out.update(sd_np(model, "final"))
    for k in range(3):
        out[f"final.mom{k}"] = opt.state[model.emb_l[k].weight]["momentum"].numpy().copy()
    out["losses"] = np.asarray(losses)
    meta = dict(name=name, m_spa=4, ln_emb=ln_emb.tolist(), ln_bot=ln_bot.tolist(), ln_top=ln_top.tolist(), B=16,
                steps=3, lr=0.05, eps=1e-10, loss="bce", sigmoid_top=1)
    out["meta"] = np.frombuffer(json.dumps(meta).encode(), dtype=np.uint8)
    np.savez_compressed(os.path.join(OUT, f"{name}.npz"), **out)
    print(f"{name}: losses {losses}")


def capture_bookkeeping(ref, ext):
    cases = []
    for size in (1, 2, 3, 4, 8):
        for n in (1, 2, 3, 7, 8, 26, 128, 2048, 65536):
            if n < size:
                continue
            per_rank = []
            for rank in range(size):
                ext.my_size, ext.my_rank = size, rank
                sl = ext.get_my_slice(n)
                mine, splits = ext.get_split_lengths(n)
                per_rank.append(dict(slice=[sl.start, sl.stop, sl.step], my_len=mine, splits=splits))
            cases.append(dict(n=n, size=size, ranks=per_rank))
    ext.my_size, ext.my_rank = -1, -1
    # interaction pair order as the reference builds it (dlrm_s_pytorch.py:499-501)
    pairs = {}
    for F in (2, 3, 4, 8, 27):
        for off in (0, 1):
            li = [i for i in range(F) for j in range(i + off)]
            lj = [j for i in range(F) for j in range(i + off)]
            pairs[f"F{F}_self{off}"] = dict(li=li, lj=lj)
    with open(os.path.join(OUT, "bookkeeping.json"), "w") as f:
        json.dump(dict(partition=cases, pairs=pairs), f)
    print("bookkeeping.json:", len(cases), "partition cases")


# ---------------------------------------------------------------------------------------------
# distributed reference run (2 gloo ranks): pins the table-sharded forward, the all-to-all layout
# and the N x embedding-gradient quirk
# ---------------------------------------------------------------------------------------------
def _dist_worker(rank, size, port, full_init, batches, cfg, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(size),
                      LOCAL_RANK=str(rank))
    ref, dp, ext = import_reference()
    ext.init_distributed(rank=rank, local_rank=rank, size=size, use_gpu=False, backend="gloo")
    ln_emb, ln_bot, ln_top = (np.asarray(cfg[k]) for k in ("ln_emb", "ln_bot", "ln_top"))
    np.random.seed(1)
    model = ref.DLRM_Net(cfg["m_spa"], ln_emb, ln_bot, ln_top, arch_interaction_op="dot", sigmoid_top=ln_top.size - 2,
                         loss_function="bce")
    # identical parameters on every rank, local tables taken from the full list
    with torch.no_grad():
        for j, g in enumerate(model.local_emb_indices):
            model.emb_l[j].weight.copy_(torch.tensor(full_init[f"init.emb_l.{g}.weight"]))
        for name, p in list(model.bot_l.named_parameters()) + []:
            p.copy_(torch.tensor(full_init[f"init.bot_l.{name}"]))
        for name, p in model.top_l.named_parameters():
            p.copy_(torch.tensor(full_init[f"init.top_l.{name}"]))
    model.bot_l = ext.DDP(model.bot_l)
    model.top_l = ext.DDP(model.top_l)
    params = [{"params": [p for emb in model.emb_l for p in emb.parameters()], "lr": cfg["lr"]},
              {"params": model.bot_l.parameters(), "lr": cfg["lr"]},
              {"params": model.top_l.parameters(), "lr": cfg["lr"]}]
    opt = torch.optim.SGD(params, lr=cfg["lr"])
    res = {}
    for s, (X, lS_o, lS_i, T) in enumerate(batches):
        Z = model(X, torch.stack(lS_o), lS_i)
        Tl = T[ext.get_my_slice(T.shape[0])]
        E = model.loss_fn(Z, Tl)
        res[f"s{s}.Z"] = Z.detach().numpy().copy()
        res[f"s{s}.loss"] = float(E.detach().numpy())
        opt.zero_grad()
        E.backward()
        opt.step()
    for j, g in enumerate(model.local_emb_indices):
        res[f"final.emb_l.{g}.weight"] = model.emb_l[j].weight.detach().numpy().copy()
    if rank == 0:
        for name, p in model.bot_l.module.named_parameters():
            res[f"final.bot_l.{name}"] = p.detach().numpy().copy()
        for name, p in model.top_l.module.named_parameters():
            res[f"final.top_l.{name}"] = p.detach().numpy().copy()
    res["local_emb_indices"] = list(model.local_emb_indices)
    res["n_emb_per_rank"] = model.n_emb_per_rank
    q.put((rank, res))
    torch.distributed.barrier()
    torch.distributed.destroy_process_group()


def capture_distributed(name="dist2_tiny", size=2, ln_emb=(30, 20, 10), B=8, top_mid=8, port=29631):
    import torch.multiprocessing as mp
    ref, dp, ext = import_reference()
    cfg = dict(m_spa=4, ln_emb=list(ln_emb), ln_bot=[5, 8, 4], lr=0.5, B=B, steps=2)
    F = len(ln_emb) + 1
    cfg["ln_top"] = [F * (F - 1) // 2 + 4, top_mid, 1]
    np.random.seed(11)
    torch.manual_seed(11)
    full = ref.DLRM_Net(cfg["m_spa"], np.asarray(cfg["ln_emb"]), np.asarray(cfg["ln_bot"]), np.asarray(cfg["ln_top"]),
                        arch_interaction_op="dot", sigmoid_top=1, loss_function="bce")
    out = {}
    full_init = sd_np(full, "init")
    out.update(full_init)
    batches = [gen_batch(dp, 5, np.asarray(cfg["ln_emb"]), cfg["B"], 3, False) for _ in range(cfg["steps"])]
    out.update(pack_batches(batches))
    # single-process reference on the same data (for the N x quirk comparison)
    opt = torch.optim.SGD(full.parameters(), lr=cfg["lr"])
    for s, (X, lS_o, lS_i, T) in enumerate(batches):
        Z = full(X, lS_o, lS_i)
        E = full.loss_fn(Z, T)
        out[f"single.s{s}.Z"] = Z.detach().numpy().copy()
        out[f"single.s{s}.loss"] = np.float64(E.detach().numpy())
        opt.zero_grad(); E.backward(); opt.step()
    out.update(sd_np(full, "single.final"))
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_dist_worker, args=(r, size, port, full_init, batches, cfg, q)) for r in range(size)]
    for p in procs:
        p.start()
    results = dict(q.get(timeout=300) for _ in range(size))
    for p in procs:
        p.join(60)
    for r, res in results.items():
        for k, v in res.items():
            out[f"rank{r}.{k}"] = np.asarray(v)
    meta = dict(name=name, size=size, **cfg, loss="bce", sigmoid_top=1)
    out["meta"] = np.frombuffer(json.dumps(meta).encode(), dtype=np.uint8)
    np.savez_compressed(os.path.join(OUT, f"{name}.npz"), **out)
    print(name, "rank losses", [[float(results[r][f"s{s}.loss"]) for s in range(cfg["steps"])] for r in range(size)])


def capture_datagen(dp, name="datagen_uniform"):
    """The reference's own generate_dist_input_batch (uniform) run on a RECORDED stream of uniforms: pins the
    transformation uniforms -> (offsets, indices) that oracle.bags_from_uniforms restates."""
    class Recorder:
        def __init__(self, seed):
            self.rs, self.log = np.random.RandomState(seed), []
        def random(self, k=None):
            r = self.rs.random_sample(k)
            self.log.append(np.atleast_1d(np.asarray(r, dtype=np.float64)).copy())
            return r
        def rand(self, *shape):
            return self.rs.rand(*shape)
    out, cases = {}, []
    for tag, ln_emb, n, P, fixed in (("var_p10", [1000, 3, 40000000], 64, 10, False), ("fixed_p4", [7, 100000], 50, 4, True),
                                     ("onehot", [5, 39884406], 80, 1, True), ("var_small_tables", [1, 2, 3], 40, 6, False)):
        rec = Recorder(1234)
        saved, dp.ra = dp.ra, rec
        try:
            X, lS_o, lS_i = dp.generate_dist_input_batch(3, np.asarray(ln_emb), n, P, fixed, "uniform", 0, 1, -1, 1)
        finally:
            dp.ra = saved
        out[f"{tag}.uniforms"] = np.concatenate(rec.log)
        for k in range(len(ln_emb)):
            out[f"{tag}.off{k}"], out[f"{tag}.idx{k}"] = lS_o[k].numpy(), lS_i[k].numpy()
        cases.append(dict(tag=tag, ln_emb=ln_emb, n=n, P=P, fixed=fixed))
    out["meta"] = np.frombuffer(json.dumps(dict(name=name, cases=cases)).encode(), dtype=np.uint8)
    np.savez_compressed(os.path.join(OUT, f"{name}.npz"), **out)
    print(name, [(c["tag"], int(sum(out[f"{c['tag']}.idx{k}"].size for k in range(len(c["ln_emb"]))))) for c in cases])


def capture_criteo_bin(name="criteo_bin"):
    """A synthetic Criteo binary file read through the reference's own CriteoBinDataset (data_loader_terabyte.py:197-251)."""
    import tempfile
    sys.path.insert(0, REF)
    import data_loader_terabyte as dlt
    rng = np.random.default_rng(3)
    n = 1000                                                     # 3 full batches of 300 + a short one of 100
    raw = np.empty((n, 40), dtype=np.int32)
    raw[:, 0] = rng.integers(0, 2, n)
    raw[:, 1:14] = (rng.pareto(1.5, (n, 13)) * 20).astype(np.int32)
    raw[:5, 1:14] = 0
    raw[:, 14:] = rng.integers(0, 2 ** 31 - 1, (n, 26), dtype=np.int64).astype(np.int32)
    raw[::7, 14:] = rng.integers(0, 50, (len(raw[::7]), 26))
    out = {"raw": raw}
    cases = []
    with tempfile.TemporaryDirectory() as td:
        f, cf = os.path.join(td, "day.bin"), os.path.join(td, "counts.npz")
        raw.tofile(f)
        np.savez(cf, counts=np.full(26, 40000000))
        for mir in (-1, 40000000, 1000):
            ds = dlt.CriteoBinDataset(f, cf, batch_size=300, max_ind_range=mir)
            assert len(ds) == 4
            for i in range(len(ds)):
                X, lS_o, lS_i, T = ds[i]
                tag = f"m{mir}.b{i}"
                out[tag + ".X"], out[tag + ".lS_o"], out[tag + ".lS_i"], out[tag + ".T"] = (X.numpy(), lS_o.numpy(), lS_i.numpy(), T.numpy())
            cases.append(dict(max_ind_range=mir, batch_size=300, batches=len(ds)))
            del ds
    out["meta"] = np.frombuffer(json.dumps(dict(name=name, cases=cases, records=n)).encode(), dtype=np.uint8)
    np.savez_compressed(os.path.join(OUT, f"{name}.npz"), **out)
    print(name, cases)


def capture_metrics(name="metrics_sklearn"):
    """scores/targets -> the scikit-learn numbers inference() reports (dlrm_s_pytorch.py:828-847)."""
    import sklearn.metrics as M
    rng = np.random.default_rng(11)
    out, cases = {}, []
    def add(tag, s, t):
        s = s.astype(np.float32); t = t.astype(np.float32)
        out[f"{tag}.scores"], out[f"{tag}.targets"] = s, t
        r = dict(tag=tag,
                 recall=float(M.recall_score(y_true=t, y_pred=np.round(s), zero_division=0)),
                 precision=float(M.precision_score(y_true=t, y_pred=np.round(s), zero_division=0)),
                 f1=float(M.f1_score(y_true=t, y_pred=np.round(s), zero_division=0)),
                 ap=float(M.average_precision_score(t, s)), roc_auc=float(M.roc_auc_score(t, s)),
                 accuracy=float(M.accuracy_score(y_true=t, y_pred=np.round(s))),
                 round_matches=int(np.sum((np.round(s, 0) == t).astype(np.uint8))))
        cases.append(r)
    n = 5000
    t = np.round(rng.random(n))
    add("informative", 1 / (1 + np.exp(-(2.0 * (t - 0.5) + rng.standard_normal(n)))), t)
    add("random", rng.random(n), t)
    add("heavy_ties", np.round(rng.random(n) * 10) / 10, t)                    # 11 distinct thresholds
    add("saturated", np.clip(np.round(rng.random(n) * 3) / 2 - 0.25, 0, 1), t)   # exact 0.0 / 0.5 / 1.0 values
    add("tiny", np.asarray([0.1, 0.4, 0.35, 0.8, 0.5, 0.5]), np.asarray([0, 0, 1, 1, 1, 0]))
    add("imbalanced", rng.random(3000) ** 3, (rng.random(3000) < 0.03).astype(np.float64))
    meta = dict(name=name, cases=cases, sklearn=__import__("sklearn").__version__)
    out["meta"] = np.frombuffer(json.dumps(meta).encode(), dtype=np.uint8)
    np.savez_compressed(os.path.join(OUT, f"{name}.npz"), **out)
    print(name, [(c["tag"], round(c["roc_auc"], 4), round(c["ap"], 4)) for c in cases])



# ---------------------------------------------------------------------------------------------
# BASELINE.json configs[2] at FULL batch: the configuration bench.py's headline number is quoted on
# ---------------------------------------------------------------------------------------------
CRITEO_TB_ROWS = [39884406, 39043, 17289, 7420, 20263, 3, 7120, 1543, 63, 38532951, 2953546, 403346, 10, 2208, 11938, 155,
                  4, 976, 14, 39979771, 25641295, 39664984, 585935, 12972, 108, 36]     # tools/visualize.py:1195-1223


def _sha(a):
    import hashlib
    a = np.ascontiguousarray(a)
    return hashlib.sha256(a.tobytes()).hexdigest()[:32]


def capture_terabyte(ref, dp, name="terabyte_b65536", row_cap=2000, B=65536, steps=3, lr=1.0, seed=123):
    """MLPerf Criteo-Terabyte shapes (bench/run_and_time.sh:17): 26 tables, D = 128, bot 13-512-256-128, top
    479-1024-1024-512-256-1, one lookup per bag, batch 65536, lr 1.0, table rows capped at `row_cap` (the 96 GB tables do
    not fit the host).  The reference's DLRM_Net, data generator and training-loop body run for `steps` steps.

    Kept small: initial parameters and input batches are NOT stored — both are pure functions of numpy's legacy global
    RandomState (frozen stream), so the fixture holds the seed, the generator state after model construction and SHA-256
    digests of every array the reference actually used; tests/golden_tb.py regenerates them (vectorised restatement of
    dlrm_data_pytorch.py:899-960,835-846) and checks the digests.  Stored: losses, predictions of every step, final MLP
    parameters, the first/last 48 rows of every final table, a few step-0 gradients."""
    import time
    ln_emb = np.asarray([min(n, row_cap) for n in CRITEO_TB_ROWS])
    ln_bot = np.asarray([13, 512, 256, 128])
    m_spa = 128
    F = ln_emb.size + 1
    ln_top = np.asarray([F * (F - 1) // 2 + m_spa, 1024, 1024, 512, 256, 1])
    np.random.seed(seed)
    torch.manual_seed(seed)
    model = ref.DLRM_Net(m_spa, ln_emb, ln_bot, ln_top, arch_interaction_op="dot", arch_interaction_itself=False,
                         sigmoid_bot=-1, sigmoid_top=ln_top.size - 2, loss_function="bce")
    out, digests = {}, {}
    for k, v in model.state_dict().items():
        digests[f"init.{k}"] = _sha(v.detach().numpy())
    st = np.random.get_state()
    assert st[0] == "MT19937"
    out["rng_after_init.keys"] = np.asarray(st[1], dtype=np.uint32)
    out["rng_after_init.pos_gauss"] = np.asarray([st[2], st[3]], dtype=np.int64)
    out["rng_after_init.cached"] = np.asarray([st[4]], dtype=np.float64)
    t0 = time.time()
    batches = [gen_batch(dp, 13, ln_emb, B, 1, True, True) for _ in range(steps)]
    print(f"{name}: reference generator {time.time() - t0:.0f} s")
    for s, (X, lS_o, lS_i, T) in enumerate(batches):
        digests[f"s{s}.X"] = _sha(X.numpy())
        digests[f"s{s}.T"] = _sha(T.numpy())
        digests[f"s{s}.idx"] = _sha(np.stack([i.numpy().astype(np.int64) for i in lS_i]))
        digests[f"s{s}.off"] = _sha(np.stack([o.numpy().astype(np.int64) for o in lS_o]))
    opt = torch.optim.SGD(model.parameters(), lr=lr)
    losses = []
    t0 = time.time()
    for s, (X, lS_o, lS_i, T) in enumerate(batches):
        Z = model(X, lS_o, lS_i)
        E = model.loss_fn(Z, T)
        out[f"s{s}.Z"] = Z.detach().numpy().copy()
        losses.append(float(E.detach().numpy()))
        opt.zero_grad()
        E.backward()
        if s == 0:
            out["s0.bot0_bias_grad"] = model.bot_l[0].bias.grad.numpy().copy()
            out["s0.top8_weight_grad"] = model.top_l[8].weight.grad.numpy().copy()
            out["s0.top0_bias_grad"] = model.top_l[0].bias.grad.numpy().copy()
            g = model.emb_l[5].weight.grad.coalesce()       # the 3-row table: every row hit ~21845 times
            out["s0.emb5_cgrad_values"] = g._values().numpy().copy()
        opt.step()
    print(f"{name}: reference training {time.time() - t0:.0f} s")
    for k, v in model.state_dict().items():
        v = v.detach().numpy()
        if k.startswith("emb_l."):
            out[f"final_head.{k}"] = v[:48].copy()
            out[f"final_tail.{k}"] = v[-48:].copy()
            out[f"final_colsum.{k}"] = v.astype(np.float64).sum(0)
            # rows that WERE updated (head/tail rows of a 4 M-row table almost never are): the rows the first 64 samples of
            # step 0 looked up in this table
            t = int(k.split(".")[1])
            out[f"final_touched.{k}"] = v[batches[0][2][t].numpy().astype(np.int64)[:64]].copy()
        else:
            out[f"final.{k}"] = v.copy()
    out["losses"] = np.asarray(losses, dtype=np.float64)
    meta = dict(name=name, m_spa=m_spa, ln_emb=ln_emb.tolist(), ln_bot=ln_bot.tolist(), ln_top=ln_top.tolist(), B=B,
                steps=steps, lr=lr, loss="bce", itself=False, sigmoid_top=int(ln_top.size - 2), seed=seed, row_cap=row_cap,
                num_idx=1, fixed=True, digests=digests, torch=torch.__version__, numpy=np.__version__,
                reference="facebookresearch/dlrm @ /root/reference (bench/run_and_time.sh:17 shapes)")
    out["meta"] = np.frombuffer(json.dumps(meta).encode(), dtype=np.uint8)
    np.savez_compressed(os.path.join(OUT, f"{name}.npz"), **out)
    print(f"{name}: losses {losses}")
    # the vectorised regeneration must reproduce what the reference used, bit for bit
    sys.path.insert(0, os.path.join(os.path.dirname(OUT)))
    import golden_tb
    fx = golden_tb.load(name)
    for s, (X, lS_o, lS_i, T) in enumerate(batches):
        assert np.array_equal(fx.batches[s][0], X.numpy()) and np.array_equal(fx.batches[s][3], T.numpy())
        assert np.array_equal(fx.batches[s][2], np.stack([i.numpy() for i in lS_i]))
    print(f"{name}: regeneration verified")


# ---------------------------------------------------------------------------------------------
# BASELINE.json configs[3] AT ITS OWN SHAPES: the reference's 8-rank run (gloo, CPU) of the MLPerf Criteo-Terabyte model —
# 26 tables table-wise over 8 ranks ([4,4,3,3,3,3,3,3]), D = 128, towers 13-512-256-128 / 479-1024-1024-512-256-1, GLOBAL batch
# 65536 (8192 per rank), rows capped (dlrm_s_pytorch.py:528-585, extend_distributed.py:541-576, README.md:345-346)
# ---------------------------------------------------------------------------------------------
def _dist_tb_worker(rank, size, port, meta, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(size),
                      LOCAL_RANK=str(rank))
    torch.set_num_threads(1)
    ref, dp, ext = import_reference()
    sys.path.insert(0, os.path.dirname(OUT))
    import golden_tb
    init, rs = golden_tb.regen_init(meta)
    batches = golden_tb.regen_batches(meta, rs)
    ext.init_distributed(rank=rank, local_rank=rank, size=size, use_gpu=False, backend="gloo")
    ln_emb, ln_bot, ln_top = (np.asarray(meta[k]) for k in ("ln_emb", "ln_bot", "ln_top"))
    np.random.seed(1)
    model = ref.DLRM_Net(meta["m_spa"], ln_emb, ln_bot, ln_top, arch_interaction_op="dot", sigmoid_top=ln_top.size - 2,
                         loss_function="bce")
    with torch.no_grad():
        for j, g in enumerate(model.local_emb_indices):
            model.emb_l[j].weight.copy_(torch.from_numpy(init[f"emb_l.{g}.weight"]))
        for name, p in model.bot_l.named_parameters():
            p.copy_(torch.from_numpy(init[f"bot_l.{name}"]))
        for name, p in model.top_l.named_parameters():
            p.copy_(torch.from_numpy(init[f"top_l.{name}"]))
    model.bot_l = ext.DDP(model.bot_l)
    model.top_l = ext.DDP(model.top_l)
    lr = meta["lr"]
    opt = torch.optim.SGD([{"params": [p for emb in model.emb_l for p in emb.parameters()], "lr": lr},
                           {"params": model.bot_l.parameters(), "lr": lr}, {"params": model.top_l.parameters(), "lr": lr}], lr=lr)
    res = {}
    for s, (X, off, idx, T) in enumerate(batches):
        Z = model(torch.from_numpy(X), torch.from_numpy(off), [torch.from_numpy(i) for i in idx])
        Tl = torch.from_numpy(T)[ext.get_my_slice(T.shape[0])]
        E = model.loss_fn(Z, Tl)
        res[f"s{s}.Z"] = Z.detach().numpy().copy()
        res[f"s{s}.loss"] = np.float64(E.detach().numpy())
        opt.zero_grad()
        E.backward()
        if s == 0 and rank == 0:
            res["s0.top8_weight_grad"] = model.top_l.module[8].weight.grad.numpy().copy()      # AFTER the DDP all-reduce (mean)
            res["s0.bot0_bias_grad"] = model.bot_l.module[0].bias.grad.numpy().copy()
        opt.step()
    for j, g in enumerate(model.local_emb_indices):
        v = model.emb_l[j].weight.detach().numpy()
        res[f"final_head.emb_l.{g}.weight"] = v[:48].copy()
        res[f"final_tail.emb_l.{g}.weight"] = v[-48:].copy()
        res[f"final_colsum.emb_l.{g}.weight"] = v.astype(np.float64).sum(0)
        res[f"final_touched.emb_l.{g}.weight"] = v[batches[0][2][g][:64]].copy()
    if rank == 0:
        for tower, mod in (("bot_l", model.bot_l.module), ("top_l", model.top_l.module)):
            for name, p in mod.named_parameters():
                v = p.detach().numpy()
                if v.ndim == 2 and v.size > 70000:
                    # large matrices: every 8th row, plus fp64 row and column sums of the whole matrix (every element is covered)
                    res[f"final_rows8.{tower}.{name}"] = v[::8].copy()
                    res[f"final_colsum.{tower}.{name}"] = v.astype(np.float64).sum(0)
                    res[f"final_rowsum.{tower}.{name}"] = v.astype(np.float64).sum(1)
                else:
                    res[f"final.{tower}.{name}"] = v.copy()
    res["local_emb_indices"] = np.asarray(list(model.local_emb_indices))
    q.put((rank, res))
    torch.distributed.barrier()
    torch.distributed.destroy_process_group()


def capture_distributed_tb(name="dist8_tb", size=8, row_cap=2000, B=65536, steps=2, lr=0.5, seed=321, port=29651):
    """Inputs and initial parameters are NOT stored (as in capture_terabyte): the fixture holds the seed, the generator state and the
    digests; tests/golden_tb.py regenerates and verifies them.  Stored: every rank's predictions and loss per step, rank 0's two
    step-0 gradients after the DDP all-reduce, the final towers (small tensors whole; large matrices as every 8th row + fp64 row/column
    sums) and head / tail / looked-up rows + fp64 column sums of every final table from the rank that owns it."""
    import time
    import torch.multiprocessing as mp
    ref, dp, ext = import_reference()
    ln_emb = np.asarray([min(n, row_cap) for n in CRITEO_TB_ROWS])
    ln_bot = np.asarray([13, 512, 256, 128])
    m_spa = 128
    F = ln_emb.size + 1
    ln_top = np.asarray([F * (F - 1) // 2 + m_spa, 1024, 1024, 512, 256, 1])
    np.random.seed(seed)
    torch.manual_seed(seed)
    model = ref.DLRM_Net(m_spa, ln_emb, ln_bot, ln_top, arch_interaction_op="dot", arch_interaction_itself=False,
                         sigmoid_bot=-1, sigmoid_top=ln_top.size - 2, loss_function="bce")
    out, digests = {}, {}
    for k, v in model.state_dict().items():
        digests[f"init.{k}"] = _sha(v.detach().numpy())
    st = np.random.get_state()
    out["rng_after_init.keys"] = np.asarray(st[1], dtype=np.uint32)
    out["rng_after_init.pos_gauss"] = np.asarray([st[2], st[3]], dtype=np.int64)
    out["rng_after_init.cached"] = np.asarray([st[4]], dtype=np.float64)
    t0 = time.time()
    batches = [gen_batch(dp, 13, ln_emb, B, 1, True, True) for _ in range(steps)]
    print(f"{name}: reference generator {time.time() - t0:.0f} s")
    for s, (X, lS_o, lS_i, T) in enumerate(batches):
        digests[f"s{s}.X"] = _sha(X.numpy())
        digests[f"s{s}.T"] = _sha(T.numpy())
        digests[f"s{s}.idx"] = _sha(np.stack([i.numpy().astype(np.int64) for i in lS_i]))
        digests[f"s{s}.off"] = _sha(np.stack([o.numpy().astype(np.int64) for o in lS_o]))
    meta = dict(name=name, size=size, m_spa=m_spa, ln_emb=ln_emb.tolist(), ln_bot=ln_bot.tolist(), ln_top=ln_top.tolist(), B=B,
                steps=steps, lr=lr, loss="bce", itself=False, sigmoid_top=int(ln_top.size - 2), seed=seed, row_cap=row_cap,
                num_idx=1, fixed=True, digests=digests, torch=torch.__version__, numpy=np.__version__,
                reference="facebookresearch/dlrm @ /root/reference: DLRM_Net.distributed_forward + extend_distributed on 8 gloo ranks "
                          "(README.md:345-346 launch pattern, bench/run_and_time.sh:17 shapes)")
    out["meta"] = np.frombuffer(json.dumps(meta).encode(), dtype=np.uint8)
    out["losses"] = np.zeros(0)
    np.savez_compressed(os.path.join(OUT, f"{name}.npz"), **out)      # (provisional: the workers regenerate through golden_tb.regen_*)
    del model, batches
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    t0 = time.time()
    procs = [ctx.Process(target=_dist_tb_worker, args=(r, size, port, meta, q)) for r in range(size)]
    for p in procs:
        p.start()
    results = dict(q.get(timeout=3600) for _ in range(size))
    for p in procs:
        p.join(120)
    print(f"{name}: reference {size}-rank training {time.time() - t0:.0f} s")
    for r, res in results.items():
        for k, v in res.items():
            out[f"rank{r}.{k}"] = np.asarray(v)
    out["losses"] = np.asarray([[float(results[r][f"s{s}.loss"]) for s in range(steps)] for r in range(size)])
    np.savez_compressed(os.path.join(OUT, f"{name}.npz"), **out)
    print(name, "rank losses", out["losses"].tolist())
    sys.path.insert(0, os.path.dirname(OUT))
    import golden_tb
    golden_tb.load(name)        # digests + RNG state of the vectorised regeneration
    print(f"{name}: regeneration verified")


# ---------------------------------------------------------------------------------------------
# BASELINE.json configs[4] inputs: the reference's Multihot class (torchrec_dlrm/multi_hot.py:27-175)
# ---------------------------------------------------------------------------------------------
def import_multihot():
    """torchrec is not installed: stub the two names multi_hot.py imports (`Batch`, `KeyedJaggedTensor.from_offsets_sync`)
    — containers only, none of the arithmetic under test lives in them."""
    tr = types.ModuleType("torchrec"); ds = types.ModuleType("torchrec.datasets"); ut = types.ModuleType("torchrec.datasets.utils")
    sp = types.ModuleType("torchrec.sparse"); jt = types.ModuleType("torchrec.sparse.jagged_tensor")

    class Batch:
        def __init__(self, dense_features, sparse_features, labels):
            self.dense_features, self.sparse_features, self.labels = dense_features, sparse_features, labels

    class KeyedJaggedTensor:
        def __init__(self, keys, values, offsets):
            self._keys, self._values, self._offsets = keys, values, offsets

        @classmethod
        def from_offsets_sync(cls, keys, values, offsets):
            return cls(keys, values, offsets)
    ut.Batch, jt.KeyedJaggedTensor = Batch, KeyedJaggedTensor
    sys.modules.update({"torchrec": tr, "torchrec.datasets": ds, "torchrec.datasets.utils": ut, "torchrec.sparse": sp,
                        "torchrec.sparse.jagged_tensor": jt})
    sys.path.insert(0, os.path.join(REF, "torchrec_dlrm"))
    import multi_hot
    return multi_hot, Batch, KeyedJaggedTensor


def capture_multihot(name="multihot_tables"):
    """The reference's own Multihot: seed-0 lookup tables (uniform and pareto), 1-hot -> multi-hot expansion and the
    cumulative offsets, on small tables (the MLPerf sizes 3,2,1,...,100,27,... need 24 GB of lookup tables)."""
    multi_hot, Batch, KJT = import_multihot()
    out, cases = {}, []
    for tag, dist, sizes, n_emb, B, B2 in (("uniform", "uniform", [3, 2, 1, 6, 1, 12], [50, 7, 3, 1000, 10, 40000], 32, 20),
                                           ("pareto", "pareto", [4, 1, 9], [100, 5, 3000], 16, 16)):
        mh = multi_hot.Multihot(sizes, n_emb, B, collect_freqs_stats=False, dist_type=dist)
        for k, t in enumerate(mh.multi_hot_tables_l):
            out[f"{tag}.table{k}"] = t.numpy().copy()
        rng = np.random.default_rng(5)
        for b_ in sorted({B, B2}):
            ids = np.stack([rng.integers(0, n, size=b_) for n in n_emb]).astype(np.int64)      # [T, b] 1-hot ids, key-major
            kjt = KJT([f"cat_{k}" for k in range(len(n_emb))], torch.from_numpy(ids.reshape(-1)), None)
            nb = mh.convert_to_multi_hot(Batch(torch.zeros(b_, 13), kjt, torch.zeros(b_)))
            out[f"{tag}.b{b_}.ids"] = ids
            out[f"{tag}.b{b_}.values"] = nb.sparse_features._values.numpy().copy()
            out[f"{tag}.b{b_}.offsets"] = nb.sparse_features._offsets.numpy().copy()
            assert nb.sparse_features._values.dtype == torch.int32
        cases.append(dict(tag=tag, dist=dist, sizes=sizes, n_emb=n_emb, batches=sorted({B, B2})))
    out["meta"] = np.frombuffer(json.dumps(dict(name=name, cases=cases, numpy=np.__version__)).encode(), dtype=np.uint8)
    np.savez_compressed(os.path.join(OUT, f"{name}.npz"), **out)
    print(name, [(c["tag"], [int(out[f"{c['tag']}.b{b}.values"].size) for b in c["batches"]]) for c in cases])


def main(which):
    os.makedirs(OUT, exist_ok=True)
    if which == "metrics":
        return capture_metrics()
    if which == "criteo_bin":
        return capture_criteo_bin()
    if which == "multihot":
        return capture_multihot()
    ref, dp, ext = import_reference()
    if which in ("all", "train"):
        # BASELINE.json configs[0]: 3 tables x 1000 x 16, bot 13-512-16, batch 128 (top tower 128-64-1)
        capture_training(ref, dp, "config1_b128", 16, [1000, 1000, 1000], [13, 512, 16], [128, 64, 1], B=128, steps=3,
                         lr=0.1, loss="bce")
        # the reference CLI's default tiny model (test/dlrm_s_test.sh): m_spa 2, emb 4-3-2, bot 4-3-2, top 4-2-1, mse
        capture_training(ref, dp, "cli_default_mse", 2, [4, 3, 2], [4, 3, 2], [4, 2, 1], B=2, steps=3, lr=0.1, loss="mse",
                         round_targets=False)
        # self-interaction, one-hot fixed lookups, odd embedding dim
        capture_training(ref, dp, "self_interact_d12", 12, [50, 7, 3, 19, 5], [6, 16, 12], [16, 1], B=33, steps=2, lr=0.2,
                         loss="bce", itself=True, num_idx=1, fixed=True)
        # multi-hot with many duplicates across bags (hot rows)
        capture_training(ref, dp, "multihot_hotrows", 8, [3, 4, 100], [9, 8], [8, 1], B=64, steps=2, lr=0.3, loss="bce",
                         num_idx=8)
    if which in ("all", "train", "options"):
        # the remaining --arch-* / loss options in one run: "cat" interaction (dlrm_s_pytorch.py:505-507), --loss-threshold
        # clamp (:607-610), --loss-function=wbce with --loss-weights (:388-391, loss_fn_wrap :150-156), multi-hot bags
        capture_training(ref, dp, "cat_wbce_clamp", 8, [30, 5, 200, 11], [7, 24, 8], [16, 1], B=96, steps=3, lr=0.3, loss="wbce",
                         num_idx=4, interaction="cat", loss_threshold=0.45, loss_weights="0.3-1.7", round_targets=True)
    if which in ("all", "train", "options", "learned"):
        # --weighted-pooling=learned: per-row pooling weights as parameters (dlrm_s_pytorch.py:289-293, 370-375, 425-428), multi-hot bags
        capture_training(ref, dp, "learned_pooling", 8, [20, 3, 150], [6, 12, 8], [10, 1], B=48, steps=3, lr=0.3, loss="bce",
                         num_idx=5, weighted_pooling="learned")
    if which in ("all", "train", "lr_schedule"):
        # the reference's learning-rate schedule (LRPolicyScheduler, :169-203; --lr-num-warmup-steps=3 --lr-decay-start-step=5
        # --lr-num-decay-steps=4): warm-up 0, 1/3, 2/3 of lr, two steps at lr, quadratic decay, then frozen — a different lr on almost every
        # one of the 11 steps; multi-hot bags with hot rows (every update mode sees duplicates)
        capture_training(ref, dp, "lr_schedule_tiny", 8, [40, 6, 300], [7, 16, 8], [12, 1], B=64, steps=11, lr=0.8, loss="bce",
                         num_idx=4, lr_schedule=(3, 5, 4), compact=True)
        # ... and on the fused lookup + interaction path of the Criteo data sets (one lookup per bag, D = 128)
        capture_training(ref, dp, "lr_schedule_onehot_d128", 128, [50, 9, 700, 3], [13, 32, 128], [64, 1], B=96, steps=8, lr=0.5, loss="bce",
                         num_idx=1, fixed=True, lr_schedule=(2, 3, 3), compact=True)
    if which in ("all", "kaggle"):
        # BASELINE.json configs[1]: Criteo-Kaggle shapes — 26 tables, D = 16, bot 13-512-256-64-16, top 512-256-1
        # (bench/dlrm_s_criteo_kaggle.sh:24), batch 2048, one lookup per bag; table rows capped at 600 to keep the fixture small
        kaggle_rows = [1460, 583, 10131227, 2202608, 305, 24, 12517, 633, 3, 93145, 5683, 8351593, 3194, 27, 14992, 5461306,
                       10, 5652, 2173, 4, 7046547, 18, 15, 286181, 105, 142572]
        capture_training(ref, dp, "kaggle_b2048", 16, [min(n, 600) for n in kaggle_rows], [13, 512, 256, 64, 16], [512, 256, 1],
                         B=2048, steps=2, lr=0.1, loss="bce", num_idx=1, fixed=True, compact=True)
    if which in ("all", "terabyte"):
        capture_terabyte(ref, dp)
    if which == "terabyte4m":
        # the HBM-resident regime of the benchmark: the seven >= 25 M-row tables capped at 4 M rows (2 GB each,
        # row keys of 22 bits, a row is looked up at most a handful of times per batch) instead of 2000 (cache-resident, ~33
        # duplicates per row).  SURVEY 8(d) names this cap as what fits the build host (62 GB).  Not part of "all": ~15 GB of RAM.
        capture_terabyte(ref, dp, name="terabyte_b65536_cap4m", row_cap=4_000_000)
    if which in ("all", "adagrad"):
        capture_adagrad(ref, dp)
    if which in ("all", "book"):
        capture_bookkeeping(ref, ext)
    if which in ("all", "datagen"):
        capture_datagen(dp)
    if which in ("all", "dist"):
        capture_distributed()
    if which in ("all", "dist8"):
        # the real Criteo split: 26 tables over 8 ranks -> [4,4,3,3,3,3,3,3] (extend_distributed.py:47-62), B = 64 -> 8 per rank
        capture_distributed("dist8_t26", size=8, ln_emb=[5 + (7 * k) % 23 for k in range(26)], B=64, top_mid=16, port=29641)
    if which == "dist8tb":
        # not part of "all": the reference's 8-rank run at the full Terabyte shapes takes a few minutes of host time
        capture_distributed_tb()
    if which == "all":
        capture_multihot()
        capture_metrics()
        capture_criteo_bin()


if __name__ == "__main__":
    main(sys.argv[1] if len(sys.argv) > 1 else "all")
