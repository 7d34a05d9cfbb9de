#!/usr/bin/env python3
"""Fold the per-kernel PMC averages written by tools/gpu_pmc_traffic.sh (pmc_FETCH_SIZE.csv / pmc_WRITE_SIZE.csv, KB per
dispatch) into per-C-ABI-call HBM traffic for bench.py's `roofline.traffic`.  FETCH_SIZE is doubled: on gfx950 rocprofv3
tallies the 128-byte requests of wide (16 B/lane) coalesced reads at 64 B (MI355X_MICROARCH.md, HBM section) — confirmed
here by interact_fwd_dma_kernel, whose only reads are the 65536 x 27 x 512 B feature rows (906 MB) and whose raw
FETCH_SIZE is 453 MB.  WRITE_SIZE is exact in this access pattern (emb_fwd writes 26 x 65536 x 512 B = 851,968 KB:
counter 851,968).   usage: python tools/pmc_to_json.py gpurun_out/pmc01 profiles/r01/pmc_traffic.json"""
import csv
import json
import sys

CATS = {
    "emb_fwd": ["emb_fwd_kernel"],
    "emb_bwd_sgd": ["expand_kernel", "rocprim::", "seg_hist_kernel", "seg_scan_kernel", "seg_colscan_kernel", "seg_binscan_kernel", "seg_scatter_kernel",
                    "sorted_update_kernel"],
    "interact_fwd": ["interact_fwd"],                                  # (the gather instantiations <NI, true> are split off below)
    "interact_bwd": ["interact_bwd"],
    "emb_interact_fwd": ["interact_fwd_dma_kernel<"],
    "emb_interact_bwd": ["interact_bwd_dma_kernel<"],
    "linear_fwd": ["gemm3_kernel<true, true", "gemm_f32_kernel<true, true"],
    "linear_bwd_data": ["gemm3_kernel<true, false", "gemm_f32_kernel<true, false"],
    "linear_bwd_weight": ["gemm3_kernel<false, false", "gemm_f32_kernel<false, false", "splitk_reduce_kernel"],
}
CALLS_PER_STEP = {"emb_fwd": 1, "emb_bwd_sgd": 1, "interact_fwd": 1, "interact_bwd": 1, "emb_interact_fwd": 1, "emb_interact_bwd": 1,
                  "linear_fwd": 8, "linear_bwd_data": 7, "linear_bwd_weight": 8}


def in_cat(cat, pats, kernel):
    """fused lookup + interaction = the <NI, true> instantiations of the LDS-DMA interaction kernels"""
    # interact_fwd_dma_kernel<NI, GATHER> / interact_bwd_dma_kernel<NI, GATHER, UPD> (UPD since ABI 17): the SECOND template argument
    targs = kernel.split("_dma_kernel<", 1)[1].split(">", 1)[0].replace(" ", "").split(",") if "_dma_kernel<" in kernel else []
    gather = len(targs) >= 2 and targs[1] == "true"
    if cat == "emb_fwd" and "emb_fwd_kernel<" in kernel and kernel.split("emb_fwd_kernel<", 1)[1].split(">", 1)[0].replace(" ", "").endswith(",true"):
        return False        # the LOOP = true instantiation is the PREDICATED launch of the fused step (ABI 16): it returns at once there
    if cat.startswith("emb_interact"):
        return gather and any(p in kernel for p in pats)
    if cat.startswith("interact"):
        return not gather and any(p in kernel for p in pats)
    return any(p in kernel for p in pats)


def load(path):
    rows = list(csv.DictReader(open(path)))
    return [(r["kernel"], int(r["calls"]), float(r["avg_value"])) for r in rows]


def main():
    src, dst = sys.argv[1], sys.argv[2]
    fetch_rows = load(f"{src}/pmc_FETCH_SIZE.csv")
    # training steps the profiled process ran = launches of the ONE dense-SGD kernel of a step (warm-up + timed steps + bench.py's comparison
    # legs: counted, not assumed — round 5's tagged-offsets leg doubled the step count and a fixed 6 doubled every figure)
    steps = sum(c for k, c, _ in fetch_rows if "sgd_dense_multi_kernel" in k) or 6
    out = {"source": "rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE (separate passes), bench.py --steps 4 --warmup 2 (%d training steps in the process)" % steps,
           "units": "bytes per C-ABI call (average over the calls of one training step)",
           "fetch_correction": "FETCH_SIZE x 2 (gfx950: 128-B requests counted as 64 B)", "kernels": {}}
    # stamp: SHA-256 of every kernel source the counters were collected on (bench.py refuses a stale file per category) + commit
    import hashlib, os, subprocess
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    csrc = os.path.join(root, "dlrm_amd", "csrc")
    out["sources"] = {f: hashlib.sha256(open(os.path.join(csrc, f), "rb").read()).hexdigest()[:16]
                      for f in sorted(os.listdir(csrc)) if f.endswith((".hip", ".h"))}
    try:
        out["git_head"] = subprocess.run(["git", "-C", root, "rev-parse", "HEAD"], capture_output=True, text=True).stdout.strip()
    except Exception:
        out["git_head"] = None
    fetch, write = fetch_rows, load(f"{src}/pmc_WRITE_SIZE.csv")
    for cat, pats in CATS.items():
        f_kb = sum(c * v for k, c, v in fetch if in_cat(cat, pats, k)) / steps
        w_kb = sum(c * v for k, c, v in write if in_cat(cat, pats, k)) / steps
        n = CALLS_PER_STEP[cat]
        if cat == "emb_fwd":
            # the stand-alone lookup kernel is not part of the fused step: bench.py measures it after the timed region (7 launches); one kernel
            # per call, so its per-call bytes are the launch-weighted mean over its own launches
            calls = sum(c for k, c, _ in fetch if in_cat(cat, pats, k))
            if calls:
                f_kb = sum(c * v for k, c, v in fetch if in_cat(cat, pats, k)) / calls
                w_kb = sum(c * v for k, c, v in write if in_cat(cat, pats, k)) / max(sum(c for k, c, _ in write if in_cat(cat, pats, k)), 1)
        out["kernels"][cat] = {"fetch_raw_bytes": f_kb * 1024 / n, "fetch_bytes": 2 * f_kb * 1024 / n,
                               "write_bytes": w_kb * 1024 / n, "traffic_bytes": (2 * f_kb + w_kb) * 1024 / n,
                               "calls_per_step": n}
    json.dump(out, open(dst, "w"), indent=1)
    for k, v in out["kernels"].items():
        print("%-18s traffic %8.1f MB/call (fetch %8.1f, write %8.1f)" % (k, v["traffic_bytes"] / 1e6, v["fetch_bytes"] / 1e6, v["write_bytes"] / 1e6))


if __name__ == "__main__":
    main()
