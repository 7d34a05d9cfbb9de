#!/usr/bin/env python3
"""Model of the 1/2/4/8-GPU step time of the table-sharded DLRM step from ONE single-GPU bench line (only one
GPU is reachable from the build container; the driver measures the real curve at round end).

    python tools/scaling_model.py profiles/r02/bench_full.json [--link-gbs 153 --link-eff 0.7 --host-ms 1.2]

Per rank at N GPUs (global batch B fixed — strong scaling, extend_distributed.py semantics):
  * MLP towers, interaction, loss: batch-split  -> single-GPU kernel time / N (efficiency loss of smaller GEMMs ignored);
  * embedding gather / update: table-split, WHOLE batch for ceil(T/N) tables -> single-GPU time * ceil(T/N)/T;
  * all-to-all of pooled embeddings, each direction: a rank sends B*T_loc*D*4*(N-1)/N bytes over (N-1) xGMI links
    (one per peer): time = bytes_per_peer / (link_gbs * link_eff); only the bottom MLP overlaps it (today's schedule);
  * gradient all-reduce of 9.5 MB (DDP): 2*(N-1)/N * 9.5 MB / (link bandwidth), overlapped with the backward GEMMs -> ignored
    unless it exceeds them;
  * the host launch path (about `host_ms` per step, measured at launch-bound shapes) bounds the step from below.
"""
import argparse
import json
import math


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("bench_json")
    ap.add_argument("--link-gbs", type=float, default=153.0, help="peak GB/s of one xGMI link, one direction")
    ap.add_argument("--link-eff", type=float, default=0.7)
    ap.add_argument("--host-ms", type=float, default=1.2)
    a = ap.parse_args()
    d = json.load(open(a.bench_json))
    k = {n: v["ms_per_step"] for n, v in d["kernels"].items()}
    cfg = d["config"]
    B, T, D = cfg["global_batch"], cfg["tables"], cfg["emb_dim"]
    dense = k["linear_fwd"] + k["linear_bwd_data"] + k["linear_bwd_weight"] + k["interact_fwd"] + k["interact_bwd"] + \
        k.get("act_bwd", 0) + k.get("bce_loss", 0) + k.get("sgd_dense", 0)
    emb = k["emb_fwd"] + k.get("emb_bwd_sgd", k.get("emb_bwd_adagrad", 0))
    bot_fraction = 0.09          # bottom MLP share of the GEMM time (0.34 of 4.73 MFLOP/sample fwd): what overlaps the exchange
    rows = []
    t1 = None
    for N in (1, 2, 4, 8):
        t_loc = math.ceil(T / N)
        dense_n = dense / N
        emb_n = emb * t_loc / T
        if N == 1:
            a2a = 0.0
        else:
            per_peer = B / N * t_loc * D * 4            # bytes one rank sends to ONE peer (its tables, that peer's batch slice)
            a2a = per_peer / (a.link_gbs * a.link_eff * 1e9) * 1e3
        overlap = min(a2a, bot_fraction * (k["linear_fwd"] / N)) if N > 1 else 0.0
        gpu = dense_n + emb_n + 2 * a2a - 2 * overlap
        step = max(gpu, a.host_ms if N > 1 else 0.0)
        if N == 1:
            step = d["ms_per_step"]
            t1 = step
        rows.append((N, t_loc, dense_n, emb_n, a2a, gpu, step, B / step * 1e3, t1 / step / N))
    print("| N | tables/rank | dense ms | embedding ms | all-to-all ms (each way) | GPU ms | step ms | samples/s | efficiency |")
    print("|---|---:|---:|---:|---:|---:|---:|---:|---:|")
    for r in rows:
        print("| %d | %d | %.2f | %.2f | %.2f | %.2f | %.2f | %.2fM | %.2f |" % (r[0], r[1], r[2], r[3], r[4], r[5], r[6], r[7] / 1e6, r[8]))


if __name__ == "__main__":
    main()
