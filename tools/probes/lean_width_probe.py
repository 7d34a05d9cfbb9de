"""Where does the bf16 weight gradient of a 200-wide layer differ from fp64?  (round 5 diagnostic for test_lean_towers_*)"""
import os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from dlrm_amd import ops
from dlrm_amd.functional import MLPFunction
dev = torch.device("cuda:0")
rng = np.random.default_rng(648)
for ln in ([128, 256, 200, 64], [128, 256, 192, 64], [128, 256, 224, 64]):
    B, L = 2048, 3
    params = []
    for i in range(L):
        params += [torch.from_numpy((rng.standard_normal((ln[i + 1], ln[i])) * np.sqrt(2 / (ln[i] + ln[i + 1]))).astype(np.float32)).to(dev).requires_grad_(True),
                   torch.from_numpy((rng.standard_normal(ln[i + 1]) * 0.1).astype(np.float32)).to(dev).requires_grad_(True)]
    x = torch.from_numpy(rng.random((B, ln[0])).astype(np.float32)).to(dev).requires_grad_(True)
    dy = torch.from_numpy(rng.standard_normal((B, ln[-1])).astype(np.float32)).to(dev)
    res = {}
    for arith in ("f32", "bf16"):
        for p in params: p.grad = None
        x.grad = None
        y = MLPFunction.apply(x, tuple([1] * L), None, ops.arith_code(arith), *params)
        y.backward(dy)
        torch.cuda.synchronize()
        res[arith] = [p.grad.double().cpu().numpy() for p in params] + [x.grad.double().cpu().numpy()]
    for k in range(len(res["f32"])):
        a, b = res["f32"][k], res["bf16"][k]
        err = np.linalg.norm(a - b) / np.linalg.norm(a)
        line = "ln %s tensor %d shape %s rel err %.4f" % (ln, k, a.shape, err)
        if a.ndim == 2 and k == 2:
            rows = np.linalg.norm(a - b, axis=1) / (np.linalg.norm(a, axis=1) + 1e-30)
            line += "  per-row-block(32): " + " ".join("%.3f" % rows[i:i + 32].mean() for i in range(0, a.shape[0], 32))
        print(line, flush=True)
