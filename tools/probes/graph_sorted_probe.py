import sys, time
sys.path.insert(0, __import__("os").path.dirname(__import__("os").path.dirname(__import__("os").path.dirname(__import__("os").path.abspath(__file__))))); sys.path.insert(0, __import__("os").path.join(__import__("os").path.dirname(__import__("os").path.dirname(__import__("os").path.dirname(__import__("os").path.abspath(__file__)))), "tests"))
import numpy as np, torch
import golden_tb, dlrm_amd
from dlrm_amd import ops
from dlrm_amd.graph import GraphedTrainStep
from dlrm_amd.optim import FusedSGD
mode = {"sorted": ops.UPD_SORTED, "atomic": ops.UPD_ATOMIC, "det": ops.UPD_DETERMINISTIC}[sys.argv[1]]
fx = golden_tb.load("terabyte_b65536"); meta = fx.meta
dev = torch.device("cuda:0")
np.random.seed(0)
m = dlrm_amd.DLRM_Net(meta["m_spa"], np.asarray(meta["ln_emb"]), np.asarray(meta["ln_bot"]), np.asarray(meta["ln_top"]), "dot", sigmoid_top=meta["sigmoid_top"], loss_function="bce").to(dev)
m.emb_update_mode = mode
opt = FusedSGD(m.parameters(), lr=0.05)
step = GraphedTrainStep(m, opt, warmup=2)
bs = [(torch.from_numpy(X).to(dev), torch.from_numpy(o).to(dev), torch.from_numpy(i).to(dev), torch.from_numpy(t).to(dev)) for X, o, i, t in fx.batches]
t0 = time.time()
for i in range(30):
    X, o, ii, t = bs[i % 3]
    l = step(X, o, ii, t)
    if i in (6, 7, 15):
        torch.cuda.synchronize(); print("sync", i, float(l), flush=True)
torch.cuda.synchronize()
print("done", sys.argv[1], float(l), "captures", step.captures, "%.2fs" % (time.time() - t0), flush=True)
