import os, sys, torch
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
from dlrm_amd import ops
dev = torch.device("cuda:0")
M, N, K = 65536, 512, 16
dY = torch.randn(M, N, device=dev); X = torch.randn(M, K, device=dev); X[:, 13:] = 0
dW = torch.empty(N, 13, device=dev); db = torch.empty(N, device=dev)
for _ in range(20): ops.linear_bwd_weight(dY, X, dW, db)
torch.cuda.synchronize()
a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
a.record()
for _ in range(200): ops.linear_bwd_weight(dY, X, dW, db)
b.record(); torch.cuda.synchronize()
ref = (dY.double().t() @ X.double())[:, :13]
print("smallk wgrad %s: %.1f us per call, max rel err %.2e" % (os.environ.get("DLRM_HIP_LIB", "head")[-12:], a.elapsed_time(b) / 200 * 1e3, float(((dW.double() - ref).abs().max() / ref.abs().max()))))
