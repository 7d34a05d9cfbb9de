"""Would the top tower's weight gradients gain from sharing the chip?  Upper-bound probe for a grouped launch: the four wgrad GEMMs of
the Terabyte top tower back to back on one stream vs spread over 2 / 4 streams (the hardware then interleaves their workgroups).  Results of the
concurrent runs are WRONG (the calls share one split-K workspace): timing only."""
import os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from dlrm_amd import ops
dev = torch.device("cuda:0")
B = 65536
shapes = [(256, 512), (512, 1024), (1024, 1024), (1024, 480)]            # (N, K) of the top tower's GEMM layers, in backward order
ops_ = []
for N, K in shapes:
    ops_.append((torch.randn(B, N, device=dev), torch.randn(B, K, device=dev), torch.empty(N, K, device=dev), torch.zeros(N, device=dev)))
def run(streams):
    main = torch.cuda.current_stream()
    ev = main.record_event()
    for i, (dY, X, dW, db) in enumerate(ops_):
        st = streams[i % len(streams)]
        st.wait_event(ev)
        with torch.cuda.stream(st):
            ops.linear_bwd_weight(dY, X, dW, db)
    for st in streams:
        main.wait_stream(st)
def timeit(streams, iters=20):
    t_end = time.perf_counter() + 0.1
    while time.perf_counter() < t_end:
        run(streams)
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(iters):
        run(streams)
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) / iters * 1e3
main = torch.cuda.current_stream()
s = [torch.cuda.Stream() for _ in range(4)]
for rep in range(2):
    print("one stream        %.1f us" % timeit([main]))
    print("two streams       %.1f us" % timeit(s[:2]))
    print("four streams      %.1f us" % timeit(s))
# narrow layer beside ONE wide layer only
ops_ = [ops_[0], ops_[2]]
for rep in range(2):
    print("512->256 + 1024^2: one stream %.1f us, two streams %.1f us" % (timeit([main]), timeit(s[:2])))
