"""Does writing the ReLU sign bits cost the forward GEMMs anything?  (round 5: the in-step 512 -> 256 forward takes 155-190 us, the isolated
microbench 135 us) — linear_fwd with / without relu_bits, and with a cold operand (another 1 GB tensor streamed in between)."""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from dlrm_amd import ops
from tools.microbench import timeit
dev = torch.device("cuda:0")
M = 65536
flush_src = torch.randn(256 << 20, device=dev)          # 1 GiB
flush_dst = torch.empty_like(flush_src)
for N, K in ((512, 16), (256, 512), (128, 256), (1024, 1024)):
    X = torch.randn(M, K, device=dev); W = torch.randn(N, K, device=dev) * 0.03; b = torch.randn(N, device=dev)
    Y = torch.empty(M, N, device=dev); bits = ops.relu_bits_alloc(M, N, dev)
    t0 = timeit(lambda: ops.linear_fwd(X, W, b, 1, Y, "f32"))
    t1 = timeit(lambda: ops.linear_fwd(X, W, b, 1, Y, "f32", relu_bits=bits))
    # cold operands: stream 2 GiB through the caches before every call; the copy's own time is measured and subtracted
    tc = timeit(lambda: flush_dst.copy_(flush_src), iters=5)
    t2 = timeit(lambda: (flush_dst.copy_(flush_src), ops.linear_fwd(X, W, b, 1, Y, "f32", relu_bits=bits)), iters=5) - tc
    print("fwd M=%d N=%d K=%d: no bits %.1f us, with bits %.1f us, with bits + cold caches %.1f us" % (M, N, K, t0 * 1e3, t1 * 1e3, t2 * 1e3), flush=True)
