"""which 64 x 64 tiles of the tower weight gradient disagree with torch, over repeated launches (debug aid)"""
import sys
import numpy as np
import torch
sys.path.insert(0, ".")
from dlrm_amd import ops

M, widths = int(sys.argv[1]) if len(sys.argv) > 1 else 17, [64, 1024, 1024, 64]
dev = torch.device("cuda:0")
g = torch.Generator().manual_seed(1)
L = len(widths) - 1
ins = [torch.randn((M, widths[l]), generator=g).to(dev) for l in range(L)]
dZs = [torch.randn((M, widths[l + 1]), generator=g).to(dev) for l in range(L)]
for rep in range(4):
    dWs = [torch.full((widths[l + 1], widths[l]), 7.0, device=dev) for l in range(L)]
    dbs = [torch.full((widths[l + 1],), 7.0, device=dev) for l in range(L)]
    ops.tower_wgrad(dZs, ins, dWs, dbs)
    torch.cuda.synchronize()
    for l in range(L):
        want = dZs[l].double().t() @ ins[l].double()
        err = (dWs[l].double() - want).abs()
        bad = err > 1e-3 * max(1.0, float(want.abs().max()))
        tiles = bad.view(widths[l + 1] // 64 if widths[l + 1] >= 64 else 1, -1, max(widths[l] // 64, 1), min(64, widths[l])).any(dim=3).any(dim=1) if widths[l + 1] % 64 == 0 and widths[l] % 64 == 0 else bad.any()
        nb = int(tiles.sum()) if hasattr(tiles, "sum") else int(tiles)
        dberr = float((dbs[l].double() - dZs[l].double().sum(0)).abs().max())
        print("rep %d layer %d: bad tiles %d, db err %.2e" % (rep, l, nb, dberr))
        if nb and hasattr(tiles, "nonzero") and tiles.dim() == 2:
            idx = tiles.nonzero().tolist()
            print("   (n-block, k-block):", idx[:40])
            i, j = idx[0]
            blk = (dWs[l][64 * i:64 * i + 64, 64 * j:64 * j + 64]).cpu()
            print("   first bad tile: got[0,:4]", blk[0, :4].tolist(), "want", want[64 * i, 64 * j:64 * j + 4].tolist(), "is7", float((blk == 7.0).float().mean()))
