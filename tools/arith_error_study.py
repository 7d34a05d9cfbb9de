#!/usr/bin/env python3
"""Device-side element-wise error study of the MLP arithmetics on the REAL operands of the benchmark.

For each of the 23 GEMMs of one Criteo-Terabyte training step (8 forward, 7 data-gradient, 8 weight-gradient; B = 65536, towers
13-512-256-128 / 479-1024-1024-512-256-1) the operands are taken from an actual step on the golden fixture of the live reference
(tests/golden/terabyte_b65536.npz: the reference's initial parameters and its first batch) — not N(0,1) — and the product is computed by
  * the native fp32 MFMA kernels (DLRM_ARITH_F32, the headline arithmetic),
  * the bf16x6 kernels (DLRM_ARITH_BF16X6: exact 3-term bf16 split, six bf16 MFMA products, fp32 accumulation),
  * torch's CPU sgemm on the host (MKL, what the reference itself computes with),
and compared element by element with an fp64 product of the same fp32 operands (torch.mm in float64 on the GPU).
Reported per GEMM: max |err| and RMS err, both relative to the RMS of the exact result, and max |err| relative to the mean of
sum_k |a_k||b_k| (the condition-independent scale).  Writes a markdown table to stdout.

    python tools/arith_error_study.py > profiles/round3/arith_error_study.md          (needs a GPU; ~2 minutes)
"""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import golden_tb  # noqa: E402
import dlrm_amd  # noqa: E402
from dlrm_amd import ops  # noqa: E402
from dlrm_amd.functional import alloc2d  # noqa: E402

dev = torch.device("cuda:0")
fx = golden_tb.load("terabyte_b65536")
meta = fx.meta
np.random.seed(0)
dlrm_amd.set_embedding_init(dev)
try:
    model = dlrm_amd.DLRM_Net(meta["m_spa"], np.asarray(meta["ln_emb"]), np.asarray(meta["ln_bot"]), np.asarray(meta["ln_top"]),
                              arch_interaction_op="dot", arch_interaction_itself=meta["itself"], sigmoid_bot=-1,
                              sigmoid_top=meta["sigmoid_top"], loss_function=meta["loss"])
finally:
    dlrm_amd.set_embedding_init(None)
model = model.to(dev)
with torch.no_grad():
    sd = model.state_dict()
    for k in list(fx.init.keys()):
        sd[k].copy_(torch.from_numpy(fx.init.pop(k)))
model.set_mlp_arith("f32")

# ---- one real training step (forward + backward) of the product in fp32, recording the operands of every MLP GEMM it launches
calls = []
_fwd, _dgrad, _wgrad = ops.linear_fwd, ops.linear_bwd_data, ops.linear_bwd_weight


def rec_fwd(X, W, bias, act, Y, *a, **k):
    calls.append(("fwd", X, W))
    return _fwd(X, W, bias, act, Y, *a, **k)


def rec_dgrad(dY, W, Xact, kind, dX, *a, **k):
    calls.append(("dgrad", dY, W))
    return _dgrad(dY, W, Xact, kind, dX, *a, **k)


def rec_wgrad(dY, X, dW, *a, **k):
    calls.append(("wgrad", dY, X))
    return _wgrad(dY, X, dW, *a, **k)


ops.linear_fwd, ops.linear_bwd_data, ops.linear_bwd_weight = rec_fwd, rec_dgrad, rec_wgrad
X, off, idx, tgt = [torch.from_numpy(a).to(dev) for a in fx.batches[0]]
Z = model(X, off, idx)
E = model.loss_fn(Z, tgt)
E.backward()
torch.cuda.synchronize()
ops.linear_fwd, ops.linear_bwd_data, ops.linear_bwd_weight = _fwd, _dgrad, _wgrad
loss_rel = abs(float(E.detach()) - fx.losses[0]) / abs(fx.losses[0])

# layer names: the forward runs bot.0-2 then top.0-4, the backward top.4-0 then bot.2-0 (a weight gradient before the data gradient of its layer)
fwd_names = ["bot.%d" % i for i in range(3)] + ["top.%d" % i for i in range(5)]
bwd_names = ["top.%d" % i for i in range(4, -1, -1)] + ["bot.%d" % i for i in range(2, -1, -1)]
gemms, nf, nw = [], 0, -1
for kind, a, b in calls:
    if kind == "fwd":
        tag = "fwd   %s %d->%d" % (fwd_names[nf], b.size(1), b.size(0))
        nf += 1
    elif kind == "wgrad":
        nw += 1
        tag = "wgrad %s %d->%d" % (bwd_names[nw], b.size(1), a.size(1))
    else:
        tag = "dgrad %s %d->%d" % (bwd_names[nw], b.size(1), b.size(0))
    gemms.append((tag, kind, a.detach(), b.detach(), None))
assert len(gemms) == 23, [g[0] for g in gemms]


def run_kernel(kind, a, b, arith):
    if kind == "fwd":                                   # Y = X W^T (no bias, no activation)
        y = alloc2d(a.size(0), b.size(0), a)
        ops.linear_fwd(a, b, None, ops.ACT_NONE, y, arith)
        return y
    if kind == "dgrad":                                 # dX = dZ W
        y = alloc2d(a.size(0), b.size(1), a)
        ops.linear_bwd_data(a, b, None, ops.ACT_NONE, y, arith)
        return y
    dW = torch.empty(a.size(1), b.size(1), device=dev)  # dW = dZ^T X
    ops.linear_bwd_weight(a, b, dW, None, arith=arith)
    return dW


def exact64(kind, a, b):
    a64, b64 = a.double(), b.double()
    if kind == "fwd":
        return a64 @ b64.t(), a64.abs() @ b64.abs().t()
    if kind == "dgrad":
        return a64 @ b64, a64.abs() @ b64.abs()
    return a64.t() @ b64, a64.abs().t() @ b64.abs()


def cpu_sgemm(kind, a, b):
    a_, b_ = a.float().cpu(), b.float().cpu()
    if kind == "fwd":
        return a_ @ b_.t()
    if kind == "dgrad":
        return a_ @ b_
    return a_.t() @ b_


print("# Element-wise error of the MLP arithmetics on the operands of a real Criteo-Terabyte step\n")
print("`tools/arith_error_study.py` on one MI355X: operands of all 23 GEMMs of one training step on the golden fixture of the live reference")
print("(`terabyte_b65536`: the reference's initial parameters and first batch, B = 65536; loss of this step vs the reference's: %.1e relative)," % loss_rel)
print("recorded from the product's own forward + backward, each product compared element by element with an")
print("fp64 product of the same fp32 operands.  `max` / `rms`: error relative to the RMS of the exact result; `max/scale`: max error relative to")
print("the mean of sum_k |a_k||b_k| (the scale a rounding error of the accumulation is proportional to).  Torch CPU sgemm = what the reference")
print("itself computes with (MKL, %d threads).\n" % torch.get_num_threads())
print("| GEMM | M x N x K | fp32 MFMA max | rms | max/scale | bf16x6 max | rms | max/scale | torch CPU sgemm max | rms | max/scale |")
print("|---|---|---:|---:|---:|---:|---:|---:|---:|---:|---:|")
worst = {"f32": 0.0, "bf16x6": 0.0, "cpu": 0.0, "bf16": 0.0}
for name, kind, a, b, _ in gemms:
    ex, scale = exact64(kind, a, b)
    rms = float(ex.pow(2).mean().sqrt())
    sc = float(scale.mean())
    cells = []
    for key, res in (("f32", run_kernel(kind, a, b, "f32").double()), ("bf16x6", run_kernel(kind, a, b, "bf16x6").double()),
                     ("cpu", cpu_sgemm(kind, a, b).to(dev).double())):
        e = (res[:ex.size(0), :ex.size(1)] - ex).abs()
        cells += ["%.2e" % (float(e.max()) / rms), "%.2e" % (float(e.pow(2).mean().sqrt()) / rms), "%.2e" % (float(e.max()) / sc)]
        worst[key] = max(worst[key], float(e.max()) / sc)
    e = (run_kernel(kind, a, b, "bf16").double()[:ex.size(0), :ex.size(1)] - ex).abs()          # single-product bf16, summary line only
    worst["bf16"] = max(worst["bf16"], float(e.max()) / sc)
    M_, N_ = ex.shape
    K_ = a.size(1) if kind == "fwd" else (a.size(1) if kind == "dgrad" else a.size(0))
    print("| %s | %d x %d x %d | %s |" % (name, M_, N_, K_, " | ".join(cells)), flush=True)
    del ex, scale
print("\nWorst max/scale over the 23 GEMMs: fp32 MFMA %.2e, bf16x6 %.2e, torch CPU sgemm %.2e; single-product bf16 (the `--arith bf16` lines) %.2e."
      % (worst["f32"], worst["bf16x6"], worst["cpu"], worst["bf16"]))
print("2^-24 = 6.0e-8 is half an fp32 ulp of the scale.  The 13->512 weight gradient (K <= 16 kernel) and the 256->1 layer (matrix-vector kernels)")
print("are fp32 FMA kernels under every `arith`: their bf16x6 columns repeat the fp32 ones by construction.")
