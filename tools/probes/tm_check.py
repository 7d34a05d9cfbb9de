import sys, os, torch
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
from dlrm_amd import ops
torch.manual_seed(0)
dev = torch.device("cuda:0")
out = []
for (M, N, K) in ((65536, 1024, 480), (4097, 256, 512), (65536, 512, 1024)):
    X = torch.randn(M, K, device=dev); W = torch.randn(N, K, device=dev) * 0.05; b = torch.randn(N, device=dev)
    Y = torch.empty(M, N, device=dev); bits = ops.relu_bits_alloc(M, N, dev)
    ops.linear_fwd(X, W, b, 1, Y, "f32", relu_bits=bits)
    dX = torch.empty(M, K, device=dev)
    ops.linear_bwd_data(Y, W, X, 1, dX, "f32")
    ref = torch.relu(X.double() @ W.double().t() + b.double())
    out.append("%d %d %d maxerr %.3e bits %d dX %.6e" % (M, N, K, (Y.double() - ref).abs().max().item(), int(bits.sum().item() % 1000003), dX.double().abs().sum().item()))
print(" | ".join(out))
