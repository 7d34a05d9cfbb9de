#!/usr/bin/env python3
"""bench.py — samples/sec of the DLRM training step on MI355X (BASELINE.json metric), one JSON line.

    python bench.py [--gpus 1] [--steps K] [--warmup W]
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P \
        bench.py --gpus N --steps K --warmup W

Workload (config.workload): BASELINE.json configs[2] — MLPerf Criteo-Terabyte shapes: 26 tables with the
row counts of the reference (tools/visualize.py:1195-1223), D = 128, bottom MLP 13-512-256-128, top MLP
479-1024-1024-512-256-1, dot interaction, one-hot lookups, GLOBAL batch 65536, fp32, BCE loss, SGD.
Synthetic data (uniform indices / dense features, rounded targets) resident in HBM before the timed
region.  A "step" = forward + loss + zero_grad + backward + optimizer.step — the region the reference
times between its time_wrap calls (dlrm_s_pytorch.py:1558-1626).  For N > 1 the global batch stays
65536 (strong scaling): tables are sharded table-wise over the ranks, pooled embeddings cross one RCCL
all-to-all per direction, MLP gradients one DDP all-reduce (extend_distributed.py semantics).
"""
from __future__ import annotations

import argparse
import gc
import json
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

# MLPerf Criteo-Terabyte table sizes (max-ind-range 40M), as recorded in the reference, tools/visualize.py:1195-1223
CRITEO_TB_ROWS = [39884406, 39043, 17289, 7420, 20263, 3, 7120, 1543, 63, 38532951, 2953546, 403346, 10, 2208, 11938, 155,
                  4, 976, 14, 39979771, 25641295, 39664984, 585935, 12972, 108, 36]
# Criteo-Kaggle table sizes, tools/visualize.py:1123-1152
CRITEO_KAGGLE_ROWS = [1460, 583, 10131227, 2202608, 305, 24, 12517, 633, 3, 93145, 5683, 8351593, 3194, 27, 14992, 5461306,
                      10, 5652, 2173, 4, 7046547, 18, 15, 286181, 105, 142572]
# MLPerf-v2 (torchrec_dlrm) table sizes and multi-hot sizes, torchrec_dlrm/README.MD:45,159 — 204.18 M rows, 214 lookups/sample
MLPERF_V2_ROWS = [40000000, 39060, 17295, 7424, 20265, 3, 7122, 1543, 63, 40000000, 3067956, 405282, 10, 2209, 11938, 155, 4,
                  976, 14, 40000000, 40000000, 40000000, 590152, 12973, 108, 36]
MLPERF_V2_HOT = [3, 2, 1, 2, 6, 1, 1, 1, 1, 7, 3, 8, 1, 6, 9, 5, 1, 1, 1, 12, 100, 27, 10, 3, 1, 1]
WORKLOADS = {
    "criteo_terabyte": dict(rows=CRITEO_TB_ROWS, D=128, bot=[13, 512, 256, 128], top=[1024, 1024, 512, 256, 1], batch=65536),
    "criteo_kaggle": dict(rows=CRITEO_KAGGLE_ROWS, D=16, bot=[13, 512, 256, 64, 16], top=[512, 256, 1], batch=2048),
    # BASELINE.json configs[4] (first slice: inputs + embedding path + row-wise Adagrad; dot interaction, dlrm_s towers):
    # int32 multi-hot indices expanded on the device by dlrm_amd.multihot.Multihot (torchrec_dlrm/multi_hot.py semantics)
    "mlperf_v2_multihot": dict(rows=MLPERF_V2_ROWS, hot=MLPERF_V2_HOT, D=128, bot=[13, 512, 256, 128],
                               top=[1024, 1024, 512, 256, 1], batch=65536),
}
HBM_PEAK_GBS = 8000.0      # MI355X_MICROARCH.md: HBM3E 8 TB/s
FP32_MFMA_PEAK_TF = 157.3  # dense fp32 MFMA peak (v_mfma_f32_32x32x2_f32)
BF16_MFMA_PEAK_TF = 2500.0  # dense bf16 MFMA peak (v_mfma_f32_32x32x16_bf16; MI355X_MICROARCH.md: ~2.5 PF dense, 2495 TF measured)
MEASURED_BF16_MFMA_TF = 2495.0
# MLP arithmetic -> (matrix-pipe peak the GEMMs are priced against, MFMA FLOPs issued per algorithmic FLOP, measured peak):
# bf16x6 issues six bf16 products per fp32 product, so its ALGORITHMIC rate is bounded by the bf16 peak / 6.  A `frac` above 1 is
# impossible by construction: the denominator is always the pipe the instructions run on.
ARITH_PEAK = {"f32": (FP32_MFMA_PEAK_TF, 1.0, 155.0), "bf16": (BF16_MFMA_PEAK_TF, 1.0, MEASURED_BF16_MFMA_TF),
              "bf16x6": (BF16_MFMA_PEAK_TF / 6.0, 6.0, MEASURED_BF16_MFMA_TF / 6.0)}
# what MI355X_MICROARCH.md MEASURES as achievable on the box (float4 device copy; back-to-back fp32 MFMA): reported beside spec
MEASURED_HBM_GBS = 6290.0
MEASURED_FP32_MFMA_TF = 155.0


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--workload", default="criteo_terabyte", choices=sorted(WORKLOADS))
    ap.add_argument("--batch", type=int, default=0, help="override the global batch (default: the workload's)")
    ap.add_argument("--row-cap", type=int, default=0, help="cap table rows (debug / small-memory runs)")
    ap.add_argument("--lr", type=float, default=0.01)
    ap.add_argument("--interaction", default="", choices=["", "dot", "dcn"],
                    help="mlperf_v2_multihot only: dcn (default there) = DCN-v2 low-rank cross network, 3 layers, rank 512 — the MLPerf-v2 "
                         "model (torchrec_dlrm/dlrm_main.py:608-619); dot = torchrec's triu dot interaction")
    ap.add_argument("--overlap", action="store_true",
                    help="headline on two HIP streams: the HBM-bound embedding kernels (pooled lookups; fused sparse update) beside the "
                         "MFMA-bound bottom-MLP GEMMs they do not depend on (DLRM_Net.overlap_streams).  Default: single stream — the "
                         "gain is 0.2-1.7 %% (profiles/r03/ceilings.md) and overlapped kernels stretch each other's event times, which "
                         "would blur the per-kernel roofline; the 2-stream schedule is measured in the same run as alt_stream_overlap")
    ap.add_argument("--no-overlap", action="store_true", help=argparse.SUPPRESS)
    ap.add_argument("--alts", action="store_true",
                    help="also measure the alternative schedules beside the headline (bf16x6 arithmetic, two-kernel lookups, 2 HIP streams; "
                         "N > 1: the other all-to-all schedule and the other dense-gradient all-reduce).  Off by default: the default run is "
                         "the headline + roofline + cpu_baseline only")
    ap.add_argument("--offsets", default="fresh", choices=["fresh", "tagged", "resident"],
                    help="what the timed steps receive as bag offsets.  fresh (default): a NEW untagged offsets tensor object every step, as the "
                         "reference loop's loader + dlrm_wrap deliver (dlrm_s_pytorch.py:129-145,1541-1548) — the fused lookup path then "
                         "proves one-lookup-per-bag on the device with a stream synchronisation INSIDE the timed region, every step; tagged: "
                         "new tensor objects carrying the producer's proof (dlrm_amd.datagen / CriteoBinBatches tag what their kernels "
                         "wrote: no device pass, no synchronisation); resident: four batches reused (their proof is cached after warm-up)")
    ap.add_argument("--no-full-size-parity", action="store_true", help="skip the 3-step comparison with the unmodified reference module at the FULL table sizes "
                                                                        "(both models resident on the GPU: 2 x 96 GB)")
    ap.add_argument("--calibrate", choices=["last", "first"], default="first",
                    help="where the box calibration (MFMA / copy / gather probes + a rocm-smi subprocess) sits: first (default since the end of round 6) "
                         "= in front of the W warm-up steps, so that the timed region follows its warm-up steps directly; last (rounds 4-6) = between "
                         "the warm-up steps and the timed region, + one untimed step — the second of idle queue it puts there cost the headline "
                         "30-90 us per step (profiles/round6/proof_wait.md)")
    ap.add_argument("--extra-warmup-step", action="store_true", help=argparse.SUPPRESS)
    ap.add_argument("--gc", choices=["default", "freeze", "off"], default="default",
                    help="Python garbage collector around the timed region: default = untouched; freeze = gc.collect() + gc.freeze() before the "
                         "warm-up (objects alive then are never scanned again); off = gc.disable() inside the timed region (A/B of host stalls)")
    ap.add_argument("--no-high-row-check", action="store_true", help="skip the top-eighth-of-every-table check of the embedding kernels after the timed region")
    ap.add_argument("--no-standalone-emb", action="store_true", help="skip the stand-alone dlrm_emb_fwd measurement (profiling runs)")
    ap.add_argument("--no-rccl-selfcheck", action="store_true", help="skip the one-rank RCCL self-check child process (N = 1)")
    ap.add_argument("--no-alt-overlap", action="store_true", help="do not also measure the 2-stream schedule (profiling runs: keeps the "
                                                                    "rocprofv3 / PMC statistics to the single-stream headline steps)")
    ap.add_argument("--no-fuse", dest="fuse", action="store_false", default=True,
                    help="run the embedding lookups and the interaction as two kernels.  Default (N = 1, one lookup per bag, D = 128, as "
                         "in DLRM_Net): the interaction kernels fetch the embedding rows themselves and the pooled-embedding buffer never "
                         "exists (DLRM_Net.fuse_emb_interact; bit-identical results); the two-kernel step is measured in the same run as "
                         "alt_two_kernel_lookup")
    ap.add_argument("--fuse", dest="fuse", action="store_true", help=argparse.SUPPRESS)
    ap.add_argument("--no-alt-fuse", action="store_true", help="do not also measure the two-kernel lookup + interaction step")
    ap.add_argument("--update-in-backward", action="store_true",
                    help="headline WITH DLRM_Net.update_in_backward (opt-in, ABI 17): from the second step on the fused backward takes the sparse SGD "
                         "step of every embedding row that ONE lookup of the batch names; the rest when the optimizer steps.  Default: off — the "
                         "headline is the reference loop's schedule (every row updated at optimizer.step()); the opt-in schedule is measured in "
                         "the same run as alt_update_in_backward")
    ap.add_argument("--no-alt-update-in-backward", action="store_true", help="do not also measure the update-in-backward step")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-parity-check", action="store_true",
                    help="skip the pre-run check of this exact configuration against the golden fixture of the live reference")
    ap.add_argument("--parity-fixture", default="terabyte_b65536", choices=["terabyte_b65536", "terabyte_b65536_cap4m"],
                    help="golden fixture of the pre-run parity check: rows capped at 2000 (default; seconds) or at 4 M (the "
                         "HBM-resident regime; ~2 min of host time to regenerate the reference's 14.5 GB of initial tables)")
    ap.add_argument("--no-alt-arith", action="store_true", help="skip the extra bf16x6 measurement")
    ap.add_argument("--graph", action="store_true",
                    help="run the timed region as HIP-graph replays of the captured step (dlrm_amd.graph; N=1 only, implies "
                         "--no-kernel-timers: events cannot be recorded inside a replay)")
    ap.add_argument("--alt-graph", action="store_true",
                    help="after the timed region, also measure the same step replayed as one HIP graph (reported as alt_hip_graph)")
    ap.add_argument("--dense-sync", choices=["ddp", "flat"], default="ddp",
                    help="N > 1: gradient all-reduce of the data-parallel MLP towers: torch DistributedDataParallel (the reference, "
                         "dlrm_s_pytorch.py:1329-1336; default) or dlrm_amd.ext_dist.FlatDDP (one flat buffer the weight-gradient GEMMs "
                         "write into, one collective per tower); the other one is measured in the same run as alt_dense_sync")
    ap.add_argument("--no-box-calibration", action="store_true", help="skip the in-run MFMA / HBM probes (profiling runs: keeps the probe "
                    "kernels out of the trace); frac_of_measured_peak then falls back to the guide's constants")
    ap.add_argument("--no-kernel-timers", action="store_true", help="no per-kernel HIP events in the timed region (no roofline)")
    ap.add_argument("--timer-every", type=int, default=4,
                    help="per-kernel HIP events are recorded on every n-th step of the timed region (an event is a queue barrier, "
                         "~5 us: one per change of launch category, ~0.1 ms per instrumented step; dlrm_amd.ops.KernelTimers)")
    ap.add_argument("--no-reference-region", action="store_true", help="skip the reference_timed_region leg (the same steps with the reference "
                                                                        "loop's per-step H2D input copies and loss read-back inside the timing)")
    ap.add_argument("--cpu-row-cap", type=int, default=4000000, help="row cap of the baseline legs' tables (SURVEY 8d: 4 M)")
    ap.add_argument("--cpu-steps", type=int, default=10, help="timed iterations of the CPU baseline (median reported; SURVEY 8d: >= 10)")
    ap.add_argument("--cpu-warmup", type=int, default=3, help="warm-up iterations of the CPU baseline (SURVEY 8d: >= 3)")
    ap.add_argument("--cpu-batch", type=int, default=16384,
                    help="rows of the global batch the CPU baseline leg iterates over (a BOUNDED sample of the same workload: the first rows of the GPU "
                         "run's first batch, same tables / weights; 0 = the whole global batch, ~6 s per iteration on this pool's hosts).  samples/s is "
                         "what is compared; 3 + 10 iterations of 16384 samples are ~20-25 s of host time")
    ap.add_argument("--cpu-budget", type=float, default=90.0, help="seconds of host time the CPU baseline leg may take: a safety net only (the timed "
                                                                    "iterations are cut short, never below 3, when a host is slower than expected)")
    ap.add_argument("--emb-update", default="sorted", choices=["sorted", "atomic", "deterministic"])
    ap.add_argument("--a2a-chunks", type=int, default=int(os.environ.get("DLRM_A2A_CHUNKS", "1")),
                    help="N > 1: schedule of the HEADLINE measurement. 1 (default) = the reference schedule: one all-to-all per "
                         "direction, only the bottom MLP overlaps it (dlrm_s_pytorch.py:563-568); C > 1 = the exchange pipelined in "
                         "C batch chunks.  Whatever is chosen here, the other schedule is measured in the same process and reported "
                         "as alt_a2a_pipelined / alt_a2a_reference")
    ap.add_argument("--alt-a2a-chunks", type=int, default=0,
                    help="chunk count of the alternative (pipelined) schedule; 0 = auto: 4 at N=2, 2 at N=4 and N=8")
    ap.add_argument("--hang-timeout", type=int, default=int(os.environ.get("DLRM_BENCH_HANG_TIMEOUT", "240")),
                    help="N > 1: seconds the rendezvous + RCCL capability probe + first step, and later each measurement phase, "
                         "may take before every thread's stack is dumped and the process exits with code 3 (a hung collective "
                         "must fail loudly, not sit until the driver kills the job)")
    ap.add_argument("--optimizer", default="sgd", choices=["sgd", "rwsadagrad"],
                    help="sgd: the reference default (and the headline); rwsadagrad: row-wise sparse Adagrad (K4, optim/rwsadagrad.py)")
    ap.add_argument("--mlp-arith", default=os.environ.get("DLRM_MLP_ARITH", "f32"), choices=["f32", "bf16x6", "bf16"],
                    help="f32: native fp32 MFMA; bf16x6: exact 3-term bf16 split of the fp32 operands, 6 bf16 MFMA products, "
                         "fp32 accumulation (fp32 round-off class)")
    return ap.parse_args()


def measure_box(device, quick=False):
    """In-run calibration of THIS box (boxes of the pool differ by up to 6 %): what its matrix pipes and its HBM sustain
    right now, measured with the library's own probes (dlrm_calib_mfma / dlrm_calib_hbm_copy, ~50 ms each) and HIP events on the
    current stream.  `frac_of_measured_peak` of every kernel category is priced against these numbers, so a slow-box line and a
    fast-box line of the same commit agree; `frac` stays priced against the spec peaks of MI355X_MICROARCH.md."""
    import ctypes as C
    from dlrm_amd import _lib
    lib = _lib.load()
    st = C.c_void_p(torch.cuda.current_stream(device).cuda_stream)
    scratch = torch.zeros(4, dtype=torch.float32, device=device)
    cus = torch.cuda.get_device_properties(device).multi_processor_count
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    out = {"cu_count": cus}
    # (kinds 2 / 3: the same probes on RANDOM operands — a GEMM's bit toggling, which is where GPUs of the pool differ: csrc/calib.hip)
    for kind, key, iters in ((0, "mfma_f32_tflops", 4000 if quick else 15000), (1, "mfma_bf16_tflops", 8000 if quick else 30000),
                             (2, "mfma_f32_random_tflops", 4000 if quick else 15000), (3, "mfma_bf16_random_tflops", 8000 if quick else 30000)):
        flop = C.c_double(0.0)
        _lib.check(lib.dlrm_calib_mfma(kind, 200, C.c_void_p(scratch.data_ptr()), C.byref(flop), st), "dlrm_calib_mfma")     # warm
        torch.cuda.synchronize(device)
        e0.record()
        _lib.check(lib.dlrm_calib_mfma(kind, iters, C.c_void_p(scratch.data_ptr()), C.byref(flop), st), "dlrm_calib_mfma")
        e1.record()
        torch.cuda.synchronize(device)
        out[key] = flop.value / (e0.elapsed_time(e1) * 1e-3) / 1e12
    out["mfma_clock_mhz"] = out["mfma_f32_tflops"] * 1e12 / (cus * 256.0) / 1e6      # 256 fp32 MFMA FLOP per clock per CU
    nbytes = 1 << 30
    a = torch.empty(nbytes, dtype=torch.uint8, device=device)
    b = torch.empty(nbytes, dtype=torch.uint8, device=device)
    a.zero_(); b.zero_()
    reps = 25 if quick else 100
    for r in range(reps + 2):
        if r == 2:
            torch.cuda.synchronize(device)
            e0.record()
        _lib.check(lib.dlrm_calib_hbm_copy(C.c_void_p(a.data_ptr()), C.c_void_p(b.data_ptr()), nbytes, st), "dlrm_calib_hbm_copy")
    e1.record()
    torch.cuda.synchronize(device)
    out["hbm_copy_gbps"] = 2.0 * nbytes * reps / (e0.elapsed_time(e1) * 1e-3) / 1e9
    del a, b
    # 512-byte rows at random places of 16 GiB (the embedding kernels' access pattern: what page-table / channel effects do to a box shows
    # here, not in the streaming probes)
    tbytes = 16 << 30
    try:
        t = torch.empty(tbytes, dtype=torch.uint8, device=device)
    except Exception as e:                                  # noqa: BLE001 — (a configuration that fills the HBM: the other probes stand)
        out["hbm_gather_error"] = repr(e)[:120]
        return out
    t.zero_()
    nb = C.c_double(0.0)
    rows = 1 << 23                                          # 4 GiB per launch
    reps = 4 if quick else 12
    total = 0.0
    for r in range(reps + 1):
        if r == 1:
            torch.cuda.synchronize(device)
            e0.record()
        _lib.check(lib.dlrm_calib_hbm_gather(C.c_void_p(t.data_ptr()), tbytes, rows, 12345 + r, C.c_void_p(scratch.data_ptr()), C.byref(nb), st),
                   "dlrm_calib_hbm_gather")
        if r >= 1:
            total += nb.value
    e1.record()
    torch.cuda.synchronize(device)
    out["hbm_gather_gbps"] = total / (e0.elapsed_time(e1) * 1e-3) / 1e9
    del t
    return out


def measure_box_or_none(device):
    """the calibration must never cost a run its headline: any failure is reported on stderr and the line goes out without `box`"""
    try:
        return measure_box(device)
    except Exception as e:                                   # noqa: BLE001
        sys.stderr.write("bench.py: box calibration failed (%r): continuing without it\n" % (e,))
        try:
            torch.cuda.empty_cache()
        except Exception:                                    # noqa: BLE001
            pass
        return None


def node_state():
    """What the NODE looks like around the run (never fatal): the current shader clock of every GPU the driver exposes under /sys/class/drm
    (the pool's nodes are shared: the other seven GPUs of a box may be running somebody else's job — `cards_at_high_clock` counts them,
    including this process's own GPU once it is busy), and this GPU's identity / power state from rocm-smi.  Recorded because visits of
    the same commit differ by 5 % (memory-bound kernels by 17-22 %) with all three calibration probes nominal (profiles/round4/box_classes.md):
    whatever separates those visits is outside what a probe on this GPU sees, so the line at least says which GPU it ran on and how busy
    its neighbours were."""
    import glob
    import re
    import subprocess
    out = {}
    try:
        clocks = []
        for f in sorted(glob.glob("/sys/class/drm/card*/device/pp_dpm_sclk")):
            cur = None
            for line in open(f):
                if "*" in line:
                    m = re.search(r"(\d+)\s*Mhz", line, re.I)
                    cur = int(m.group(1)) if m else None
            clocks.append(cur)
        out["sclk_mhz_all_cards"] = clocks
        out["cards_at_high_clock"] = sum(1 for c in clocks if c and c >= 1500)
    except Exception as e:                                   # noqa: BLE001 — diagnostics only
        out["sysfs_error"] = repr(e)[:120]
    try:
        r = subprocess.run(["rocm-smi", "--showuniqueid", "--showmaxpower", "--showpower", "--showtemp", "--showclocks", "--json"],
                           capture_output=True, text=True, timeout=20)
        cards = json.loads(r.stdout[r.stdout.index("{"):])
        out["smi"] = {name: {k.strip().rstrip(":"): v for k, v in card.items()} for name, card in cards.items() if isinstance(card, dict)}
    except Exception as e:                                   # noqa: BLE001
        out["smi_error"] = repr(e)[:120]
    return out


def merge_box(b0, b1):
    """mean of the calibration before and after the timed region (+ the two readings, so drift inside a run is visible)"""
    box = {k: (0.5 * (b0[k] + b1[k]) if (isinstance(b0[k], float) and isinstance(b1.get(k), float)) else b0[k]) for k in b0}
    for k in b1:
        box.setdefault(k, b1[k])
    box["before"] = {k: round(v, 2) for k, v in b0.items() if isinstance(v, float)}
    box["after"] = {k: round(v, 2) for k, v in b1.items() if isinstance(v, float)}
    box["note"] = ("measured in this run by dlrm_calib_mfma / dlrm_calib_hbm_copy / dlrm_calib_hbm_gather in front of the warm-up steps and right after the timed "
                   "region (mean); frac_of_measured_peak is priced against the constant-operand MFMA and copy rates, frac against the spec peaks; "
                   "mfma_*_random_tflops = the MFMA probes on random operands (what the chip holds under a GEMM's bit toggling); hbm_gather_gbps = "
                   "512-byte rows at random places of 16 GiB, the embedding kernels' access pattern (reported, not priced against)")
    return box


def make_batches(n, B, rows, device, seed, hot=None, local_rows=None):
    """Synthetic batches in the reference's layout (dlrm_data_pytorch.py:899-960 with one lookup per bag, as the Criteo
    data sets have): indices and offsets as stacked [T, B] int64 tensors (row t = table t), generated on the device by
    dlrm_amd.datagen (same distributions as the reference generator, Philox stream).
    hot = per-table multi-hot sizes (MLPerf-v2): the 1-hot ids are expanded on the device through HBM-resident lookup
    tables (dlrm_amd.multihot.Multihot = torchrec_dlrm/multi_hot.py:80-159) into int32 bags of hot[t] ids; returns the
    per-batch expansion time as well (HIP events).
    local_rows (a slice; sharded model, N > 1): the batch is the SAME global batch on every rank (same seed), but only this rank's
    samples are kept and expanded — (X[slice], values, None, T[slice]) with `values` the key-major ids of those samples (the per-rank
    KJT of the reference's torchrec loader, multi_hot_criteo.py:200-214): inputs are NOT replicated."""
    from dlrm_amd.datagen import UniformBatchGenerator
    gen = UniformBatchGenerator(13, rows, 1, True, round_targets=True, seed=seed, device=device,
                                index_dtype=torch.int32 if hot else torch.int64)
    mh, expand_ms = None, []
    if hot:
        from dlrm_amd.multihot import Multihot
        mh = Multihot(hot, rows, B, dist_type="uniform", device=device, seed=0)
    out = []
    for k in range(n):
        if mh is None:
            out.append(gen.batch(B, batch_no=k, stacked=True))            # [T, B] offsets / indices; offsets tagged by their producer
            continue
        X, lS_o, lS_i, T = gen.batch(B, batch_no=k)
        if mh is None:
            pass
        elif local_rows is not None:
            ids = torch.stack(lS_i)[:, local_rows].contiguous()
            mh.expand(ids, want_global_offsets=False)                     # warm
            a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            a.record()
            values, _, _ = mh.expand(ids, want_global_offsets=False)
            b.record()
            torch.cuda.synchronize()
            expand_ms.append(a.elapsed_time(b))
            out.append((X[local_rows].contiguous(), values, None, T[local_rows].contiguous()))
        else:
            ids = torch.stack(lS_i)
            mh.to_model_inputs(ids)                                       # warm
            a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            a.record()
            off, idx = mh.to_model_inputs(ids)
            b.record()
            torch.cuda.synchronize()
            expand_ms.append(a.elapsed_time(b))
            out.append((X, off, idx, T))
    return (out, expand_ms) if hot else out


def _block_partition_imbalance(plan, N):
    """max / mean per-rank cost of the reference's contiguous block partition (extend_distributed.py:47-62) for the same tables"""
    from dlrm_amd import sharding
    owner = sharding.reference_plan(len(plan.cost), N)
    per = [sum(c for c, o in zip(plan.cost, owner) if o == r) for r in range(N)]
    return max(per) / (sum(per) / N)


def baseline_state(model, batch, wl, ln_top, args):
    """What the baseline legs run on: the GPU run's own MLP weights, the first `cpu_row_cap` rows of its tables and its first
    batch, copied to the host (same weights, same pre-generated inputs)."""
    cap = args.cpu_row_cap
    X, off, idx, T = batch
    tables = [e.weight.detach()[:cap].cpu().contiguous() for e in model.emb_l]
    rows = torch.tensor([t.shape[0] for t in tables], dtype=idx.dtype, device=idx.device).view(-1, 1)
    mlp = {k: v.detach().cpu().clone() for k, v in model.state_dict().items() if not k.startswith("emb_l.")}
    full = (X.cpu(), off.cpu(), (idx % rows).cpu(), T.cpu())
    nb = int(args.cpu_batch) if 0 < int(args.cpu_batch) < X.shape[0] else X.shape[0]
    # the CPU leg's sample: the first nb samples of that batch (one lookup per bag: bag b of every table is lookup b, so the slice is a batch)
    sample = (full[0][:nb].contiguous(), full[1][:, :nb].contiguous(), full[2][:, :nb].contiguous(), full[3][:nb].contiguous())
    return {"m_spa": wl["D"], "ln_bot": list(wl["bot"]), "ln_top": [int(v) for v in ln_top], "tables": tables, "mlp": mlp,
            "batch": full, "cpu_batch": sample, "row_cap": cap}


def baseline_legs(state, args, device):
    """The reference timed beside the GPU path, in the same run on the same box: `cpu_baseline` on the host cores and
    `stock_gpu_baseline` (the unmodified reference with --use-gpu semantics on the same MI355X).  With oracle/_ref present (the
    reference compiled where it lay, `make -C oracle ref`) both run the REAL reference ("kind": "reference"); without it the CPU
    leg falls back to oracle/torch_port.py, the bit-pinned port of its operator calls ("kind": "port")."""
    import contextlib
    from oracle import ref_baseline
    with contextlib.redirect_stdout(sys.stderr):        # the reference prints at import ("Unable to import mlperf_logging"): stdout carries ONE JSON line
        out = ref_baseline.run(state, args.lr, cpu_warmup=args.cpu_warmup, cpu_steps=args.cpu_steps, gpu_device=device,
                               cpu_budget_s=args.cpu_budget)
    if out is not None:
        cpu, stock = out
        return {"cpu_baseline": cpu, "stock_gpu_baseline": stock}
    from oracle.torch_port import TorchPortDLRM
    default_threads = torch.get_num_threads()
    torch.set_num_threads(os.cpu_count())
    params = {f"emb_l.{k}.weight": t for k, t in enumerate(state["tables"])}
    params.update(state["mlp"])
    m = TorchPortDLRM(params, sigmoid_top=len(state["ln_top"]) - 2, loss="bce", lr=args.lr)
    X, off, idx, T = state.get("cpu_batch", state["batch"])
    off, idx = list(off), list(idx)
    times, t_begin = [], time.time()
    for it in range(args.cpu_warmup + args.cpu_steps):
        t0 = time.time()
        m.train_step(X, off, idx, T)
        if it >= args.cpu_warmup:
            times.append((time.time() - t0) * 1e3)
        if len(times) >= 3 and time.time() - t_begin > args.cpu_budget:
            break
    med = float(np.median(times))
    B = X.shape[0]
    torch.set_num_threads(default_threads)
    return {"cpu_baseline": {
        "value": B / (med * 1e-3), "unit": "samples/s", "cores": os.cpu_count(), "kind": "port", "ms_per_step": med,
        "ms_per_step_min_max": [float(min(times)), float(max(times))],
        "sample": f"{args.cpu_warmup} warm-up + {args.cpu_steps} timed steps (median) of global batch {B}, tables capped at "
                  f"{state['row_cap']} rows, the GPU run's own MLP weights / table rows / first batch (oracle/torch_port.py: the "
                  f"reference's torch CPU operator calls)",
        "threads": {"used": os.cpu_count(), "os_cpu_count": os.cpu_count(), "torch_default": default_threads},
        "parallel_info": torch.__config__.parallel_info().strip().splitlines()[:8],
        "deviations": [f"tables capped at {state['row_cap']} rows (the 96 GB of tables do not fit the host)",
                       "oracle/_ref (the compiled reference) is absent on this box: a port of the reference's operator calls "
                       "(bit-pinned by tests/test_oracle_golden.py), with the tril index lists cached"]},
        "stock_gpu_baseline": None}


def parity_check(args, device):
    """Before anything is timed: the bench configuration (stacked [T, B] inputs, sorted fused update, FusedSGD, the GEMM /
    interaction kernels these shapes select, the requested MLP arithmetic) trained for 3 steps on the full-batch golden
    fixture of the live reference (tests/golden/terabyte_b65536.npz: B = 65536, 26 tables, D = 128, rows capped at 2000;
    oracle/make_golden.py capture_terabyte).  north_star bar: fp32 loss within 1e-5 relative at every step."""
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import golden_tb
    mode = {"sorted": 2, "atomic": 0, "deterministic": 1}[args.emb_update]
    try:
        reduced = None
        if args.mlp_arith == "bf16":
            # bf16 operand rounding is an opt-in arithmetic, not the north_star fp32 path: the fp32 run below stays the parity claim for this
            # configuration; the bf16 run itself is held to its own stated bar on the loss (1e-3; its predictions were measured ~5e-4 relative off the reference's)
            rel16 = golden_tb.run_on_gpu(device, arith="bf16", mode=mode, check=False, overlap=bool(args.overlap) and not args.no_overlap,
                                         fuse=bool(args.fuse), name=args.parity_fixture)
            reduced = {"mlp_arith": "bf16", "rel_err_per_step": rel16, "bar": 1e-3, "pass": bool(max(rel16) <= 1e-3),
                       "note": "loss only: bf16 operands (2^-9 relative rounding) cannot meet the fp32 bars; the fp32 entry beside this is the parity claim"}
            torch.cuda.empty_cache()
        rel = golden_tb.run_on_gpu(device, arith="f32" if reduced else args.mlp_arith, mode=mode, check=True,
                                   overlap=bool(args.overlap) and not args.no_overlap, fuse=bool(args.fuse), name=args.parity_fixture)
        if reduced:
            return {"fixture": "tests/golden/%s.npz" % args.parity_fixture, "f32": {"rel_err_per_step": rel, "bar": 1e-5, "pass": bool(max(rel) <= 1e-5)},
                    "bf16": reduced, "pass": bool(max(rel) <= 1e-5 and reduced["pass"]), "mlp_arith": args.mlp_arith, "embedding_update": args.emb_update}
        return {"fixture": "tests/golden/%s.npz (3 training steps of the live reference at B=65536, T=26, D=128, "
                           "towers 13-512-256-128 / 479-1024-1024-512-256-1, lr 1.0, rows capped at %s)"
                           % (args.parity_fixture, "4000000" if args.parity_fixture.endswith("cap4m") else "2000"),
                "rel_err": max(rel), "rel_err_per_step": rel, "bar": 1e-5, "pass": bool(max(rel) <= 1e-5),
                "also_checked": "predictions rtol 2e-5, 3 step-0 gradients rtol 2e-4, final MLP parameters and table rows/column sums rtol 1e-4",
                "other_fixture": "tests/golden/terabyte_b65536_cap4m.npz (rows capped at 4 M: HBM-resident tables) is checked by "
                                 "tests/test_gpu_model.py and by --parity-fixture terabyte_b65536_cap4m",
                "mlp_arith": args.mlp_arith, "embedding_update": args.emb_update}
    except AssertionError as e:
        return {"fixture": "tests/golden/%s.npz" % args.parity_fixture, "pass": False, "error": str(e)[:400]}


def parity_check_v2(args, device, interaction):
    """--workload mlperf_v2_multihot: before anything is timed, this workload's configuration (torchrec model semantics, 214 int32
    lookups per sample through dlrm_amd.multihot, fused row-wise Adagrad + dense Adagrad lr 0.005 eps 1e-8, B = 65536, tables capped at
    200 k rows) trains 3 steps against tests/golden/mlperf_v2_{dot,dcn}_b65536.npz (oracle/make_golden_v2.py: the reference's own
    RWSAdagrad driving a torch-operator restatement of the torchrec model) — once in fp32 (bar 1e-5 where the configuration is
    well-conditioned, see tests/golden_v2.py) and once in the arithmetic being benchmarked (bf16: a measured, stated tolerance)."""
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import golden_v2
    if not golden_v2.available(interaction):
        return {"fixture": "tests/golden/mlperf_v2_%s_b65536.npz" % interaction, "pass": None, "error": "fixture not present"}
    out = {"fixture": "tests/golden/mlperf_v2_%s_b65536.npz (3 steps at B=65536, 214 lookups/sample, rows capped at 200000; optimizer = the "
                      "reference's optim/rwsadagrad.py, model = torch-operator restatement of torchrec's DLRM%s: UNPINNED third-party semantics)"
                      % (interaction, "_DCN" if interaction == "dcn" else ""),
           "bars": {"loss_rtol": golden_v2.LOSS_RTOL, "logit_atol": golden_v2.LOGIT_ATOL,
                    "note": "bench variant (zero initial accumulator): step 0 at the bar, steps 1-2 at max(bar, 4 x the fixture's own fp32-vs-fp64 "
                            "spread: Adagrad's first step is a sign descent); conditioned variant (initial accumulator 1.0): all 3 steps at the bar"}}
    ok = True
    for arith in sorted({"f32", args.mlp_arith}, key=lambda a: a != "f32"):
        try:
            out[arith] = golden_v2.run_on_gpu(device, interaction, arith)
            out[arith]["pass"] = True
        except AssertionError as e:
            out[arith] = {"pass": False, "error": str(e)[:400]}
            ok = False
        torch.cuda.empty_cache()
    out["pass"] = ok
    return out


# kernel category -> the sources its kernels are compiled from: PMC traffic measured on an older version of ANY of them is stale
KERNEL_SOURCES = {
    "emb_fwd": ["emb.hip", "common.h"],
    "emb_bwd_sgd": ["emb_sorted.hip", "sorted_common.h", "seg_sort.h", "common.h"],
    "emb_bwd_adagrad": ["adagrad.hip", "sorted_common.h", "seg_sort.h", "common.h"],
    "interact_fwd": ["interact.hip", "common.h"], "interact_bwd": ["interact.hip", "common.h"],
    "emb_interact_fwd": ["interact.hip", "common.h"], "emb_interact_bwd": ["interact.hip", "common.h"],
    "linear_fwd": ["gemm.hip", "gemv.hip", "common.h"], "linear_bwd_data": ["gemm.hip", "gemv.hip", "common.h"],
    "linear_bwd_weight": ["gemm.hip", "gemv.hip", "smallk.hip", "common.h"],
}


def full_size_parity(model, opt, wl, ln_top, batches, lr, device, steps=3):
    """"the 1e-5 loss bar has never been checked on the 96 GB model" — every golden fixture caps the tables (2000 / 4 M rows)
    because the reference's CPU path does not fit a host with the full tables.  288 GB of HBM hold BOTH models: the unmodified reference
    `DLRM_Net` (oracle/_ref; stock PyTorch-ROCm kernels, `--use-gpu` semantics) is given a copy of THIS model's present parameters — all 26 tables at
    full size — and both then train `steps` steps on the same batches; the losses must agree to 1e-5 relative (north_star's bar) and the rows the
    first samples looked up must agree afterwards.  The checker is the reference's own code; what it is NOT is the reference's CPU arithmetic
    (ATen's GPU kernels re-associate sums too: the fixtures cover the CPU path at capped sizes)."""
    from oracle import ref_baseline
    ref = ref_baseline.load_reference()
    if ref is None:
        return {"skipped": "oracle/_ref (the compiled reference) is absent"}
    free, _total = torch.cuda.mem_get_info(device)
    need = sum(e.weight.numel() * 4 for e in model.emb_l)
    if free < need + (8 << 30):
        return {"skipped": "not enough free HBM for a second copy of the tables (%.0f GB free, %.0f GB needed)" % (free / 1e9, need / 1e9)}
    D = wl["D"]
    tables = [e.weight.detach().clone() for e in model.emb_l]                       # device-to-device: the reference model's own storage
    mlp = {k: v.detach().clone() for k, v in model.state_dict().items() if not k.startswith("emb_l.")}
    refm = ref_baseline.build_model(ref, D, list(wl["bot"]), [int(v) for v in ln_top], tables, mlp, ndevices=1).to(device)
    ref.dlrm = refm
    optr = torch.optim.SGD(refm.parameters(), lr=lr)
    one = torch.ones((), dtype=torch.float32, device=device)
    ours, theirs = [], []
    for s_ in range(steps):
        X, off, idx, T = batches[s_ % len(batches)]
        Zr = refm(X, off, idx)
        Er = refm.loss_fn(Zr, T)
        optr.zero_grad()
        Er.backward()
        optr.step()
        Z = model(X, off, idx)
        E = model.loss_fn(Z, T)
        opt.zero_grad()
        E.backward(one)
        opt.step()
        ours.append(float(E.detach())); theirs.append(float(Er.detach()))
        pred_err = float((Z.detach() - Zr.detach()).abs().max())
        del E, Er, Z, Zr
    torch.cuda.synchronize()
    rel = [abs(a - b) / abs(b) for a, b in zip(ours, theirs)]
    # rows that WERE updated: what the first 256 samples of the first batch looked up, every table
    row_err = 0.0
    X, off, idx, T = batches[0]
    for t, (e, r) in enumerate(zip(model.emb_l, refm.emb_l)):
        rows_t = idx[t, :256] if idx.dim() == 2 else idx[t][:256]
        a, b = e.weight.detach()[rows_t], r.weight.detach()[rows_t]
        row_err = max(row_err, float(((a - b).abs() / (b.abs() + 1e-6)).max()))
    mlp_err = 0.0
    sd_o, sd_r = model.state_dict(), refm.state_dict()
    for k in mlp:
        mlp_err = max(mlp_err, float(((sd_o[k] - sd_r[k]).abs().max() / (sd_r[k].abs().max() + 1e-12))))
    out = {"table_rows_total": int(sum(e.weight.size(0) for e in model.emb_l)), "table_bytes": int(need), "steps": steps,
           "loss": ours, "reference_loss": theirs, "rel_err_per_step": rel, "bar": 1e-5, "pass": bool(max(rel) <= 1e-5),
           "max_prediction_abs_err_last_step": pred_err, "max_rel_err_updated_table_rows": row_err, "max_rel_err_mlp_parameters": mlp_err,
           "what": "the unmodified reference DLRM_Net (oracle/_ref) on the same MI355X holding a copy of this model's parameters at the FULL "
                   "table sizes, both trained on the same batches with SGD: loss within 1e-5 relative per step"}
    del refm, optr, tables
    torch.cuda.empty_cache()
    return out


def sync_device():
    """torch.cuda.synchronize() at the end of a timed region, reached by POLLING: a thread asleep in hipDeviceSynchronize wakes up whenever the box
    lets it (30 us ... 1.2 ms measured on this pool, profiles/round6/proof_wait.md), and that latency would be charged to the steps.  The device
    synchronisation itself still happens (it returns at once)."""
    from dlrm_amd import ops
    ops.wait_spinning(torch.cuda.current_stream(), limit_s=30.0)
    torch.cuda.synchronize()


def host_step_intervals(t0, returns, gc_before):
    """Host time between the returns of consecutive timed step() calls (ms).  With --offsets fresh every step ends the host's run-ahead once
    (the proof's verdict), so these intervals ARE the steps as the GPU ran them and a single long one is a host stall the GPU sat out
    (garbage collection, a descheduled thread, a slow wake-up): the headline is their mean, this says how it is distributed."""
    if len(returns) < 2:
        return None
    iv = sorted((b - a) * 1e3 for a, b in zip([t0] + returns[:-1], returns))
    n = len(iv)
    gc_now = [g["collections"] for g in gc.get_stats()]
    return {"min_ms": iv[0], "median_ms": iv[n // 2], "max_ms": iv[-1], "second_max_ms": iv[-2], "first_ms": (returns[0] - t0) * 1e3,
            "python_gc_collections_in_timed_region": [b - a for a, b in zip(gc_before, gc_now)], "gc_enabled": gc.isenabled(),
            "note": "intervals between the returns of step(); with the default --offsets fresh the host waits for the GPU once per step, so "
                    "median_ms is the undisturbed step and (ms_per_step - median_ms) * steps the sum of the host stalls of the region"}


def high_row_check(model, D, device, B=4096, seed=99):
    """The headline's embedding kernels on the benchmark's OWN tables at their full size, lookups drawn from the TOP EIGHTH of every table
    (byte offsets of up to 20 GB from a table base; tools/visualize.py:1195-1223 sizes): dlrm_emb_fwd and the fused lookup + interaction
    kernels against torch's index_select + the plain interaction kernels (bit-identical), and the sorted fused SGD update on distinct
    rows against w - lr*g computed by torch on the gathered rows (exact for lr = 0.5), the touched rows restored afterwards.  Runs
    after the timed region; tests/test_gpu_bigtables.py is the thorough version (closed-form tables, whole-table comparison)."""
    from dlrm_amd import ops
    Ws = [e.weight.detach() for e in model.emb_l]
    T = len(Ws)
    g_ = torch.Generator(device="cpu").manual_seed(seed)
    idx = []
    for W in Ws:
        n = W.size(0)
        span = max(n // 8, 1)
        if span >= B:                                  # distinct rows of the top eighth, the last row included
            r = n - 1 - torch.randperm(span, generator=g_)[:B]
        else:
            r = n - 1 - torch.randint(0, span, (B,), generator=g_)
        idx.append(r.to(torch.int64))
    I = torch.stack(idx).to(device)
    Ofs = torch.arange(B, device=device).repeat(T, 1)
    bags = ops.BagBatch(Ofs, I)
    F = T + 1
    x = torch.randn((B, D), device=device)
    feat = torch.empty((B, F * D), device=device)
    feat[:, :D] = x
    for t in range(T):
        feat[:, (1 + t) * D:(2 + t) * D] = Ws[t].index_select(0, I[t])
    pooled = torch.empty((B, T * D), device=device)
    ops.emb_fwd(Ws, bags, pooled)
    out = {"rows": "top eighth of each of the %d full-size tables (max row %d), %d lookups per table" % (T, max(w.size(0) for w in Ws) - 1, B),
           "emb_fwd": bool(torch.equal(pooled, feat[:, D:]))}
    ldr = (ops.interact_out_width(F, D, False) + 3) & ~3
    R0, R1 = torch.empty((B, ldr), device=device), torch.empty((B, ldr), device=device)
    ops.interact_fwd([feat[:, :D], feat[:, D:]], D, False, R0)
    ops.interact_fwd_gather(x, Ws, bags, D, False, R1)
    out["fused_lookup_interaction_fwd"] = bool(torch.equal(R0, R1))
    dR = torch.randn((B, ldr), device=device)
    d0 = torch.empty((B, F * D), device=device)
    ops.interact_bwd([feat[:, :D], feat[:, D:]], D, False, dR, [d0[:, :D], d0[:, D:]])
    dx, dE = torch.empty((B, D), device=device), torch.empty((B, T * D), device=device)
    ops.interact_bwd_gather(x, Ws, bags, D, False, dR, dx, dE)
    out["fused_lookup_interaction_bwd"] = bool(torch.equal(d0[:, :D], dx) and torch.equal(d0[:, D:], dE))
    big = [t for t in range(T) if Ws[t].size(0) // 8 >= B]                  # tables whose drawn rows are distinct
    before = {t: Ws[t].index_select(0, I[t]) for t in big}
    gsel = torch.randn((B, len(big) * D), device=device)
    ops.emb_bwd_sgd([Ws[t] for t in big], ops.BagBatch(Ofs[:len(big)], torch.stack([I[t] for t in big])), gsel, 0.5, ops.UPD_SORTED)
    ok = True
    for j, t in enumerate(big):
        ok = ok and bool(torch.equal(Ws[t].index_select(0, I[t]), before[t] - 0.5 * gsel[:, j * D:(j + 1) * D]))
        Ws[t].index_copy_(0, I[t], before[t])                               # the tables leave as they came
    out["sorted_sgd_update"] = ok
    ops.check_index_errors(sync=True)
    out["ok"] = bool(out["emb_fwd"] and out["fused_lookup_interaction_fwd"] and out["fused_lookup_interaction_bwd"] and ok)
    return out


def rccl_selfcheck(timeout=300):
    """`python -m dlrm_amd.selfcheck` in a child process: a one-rank RCCL group forced through the distributed code path, compared with
    the single-process steps (see that module).  A child, so that nothing RCCL does can hold up or break the headline line."""
    import socket
    import subprocess
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "LOCAL_WORLD_SIZE", "MASTER_PORT")}
    sk = socket.socket()
    sk.bind(("127.0.0.1", 0))
    env["MASTER_PORT"], env["MASTER_ADDR"] = str(sk.getsockname()[1]), "127.0.0.1"
    sk.close()
    try:
        r = subprocess.run([sys.executable, "-m", "dlrm_amd.selfcheck"], cwd=ROOT, env=env, capture_output=True, text=True, timeout=timeout)
        lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
        if lines:
            return json.loads(lines[-1])
        return {"ok": False, "error": "no JSON line (exit code %d): %s" % (r.returncode, r.stderr[-400:])}
    except subprocess.TimeoutExpired:
        return {"ok": False, "error": "timed out after %d s" % timeout}
    except Exception as e:                                  # noqa: BLE001 - a report, never allowed to break the headline line
        return {"ok": False, "error": "%s: %s" % (type(e).__name__, str(e)[:300])}


def source_hashes():
    import hashlib
    csrc = os.path.join(ROOT, "dlrm_amd", "csrc")
    return {f: hashlib.sha256(open(os.path.join(csrc, f), "rb").read()).hexdigest()[:16]
            for f in sorted(os.listdir(csrc)) if f.endswith((".hip", ".h"))}


def load_pmc_traffic():
    """Latest committed profiles/rNN/pmc_traffic.json (tools/gpu_pmc_traffic.sh + tools/pmc_to_json.py), or None.  The file
    is stamped with the SHA-256 of every kernel source it was measured on; a category whose sources have changed since
    gets `stale = True` and bench.py reports `traffic: null` for it instead of a number that no longer describes HEAD."""
    import glob
    files = sorted(glob.glob(os.path.join(ROOT, "profiles", "r*", "pmc_traffic.json")))
    if not files:
        return None
    d = json.load(open(files[-1]))
    d["_file"] = os.path.relpath(files[-1], ROOT)
    now, then = source_hashes(), d.get("sources")
    for cat, k in d["kernels"].items():
        k["stale"] = then is None or any(now.get(f) != then.get(f) for f in KERNEL_SOURCES.get(cat, list(now)))
    return d


def self_launch_command(args, argv, port=None):
    """`python bench.py --gpus N` with N > 1 and no launcher around it (WORLD_SIZE unset): the command line this file re-executes
    itself under — one rank per GPU, the reference's own launch pattern (README.md:345-346, torch.distributed.launch -> .run)."""
    port = port or int(os.environ.get("MASTER_PORT", 29500 + os.getpid() % 2000))
    return [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={args.gpus}", "--master-addr", "127.0.0.1",
            "--master-port", str(port), os.path.abspath(__file__)] + list(argv)


def resolve_world(args, argv=None, environ=None):
    """Decide how many ranks this invocation is.  Returns ("run", N) for a rank (or the single process) that measures, ("exec", cmd)
    when `--gpus N > 1` was given WITHOUT a launcher: the caller re-executes under torch.distributed.run.  A launcher whose
    WORLD_SIZE disagrees with --gpus is an ERROR (exit 2), never a line with a different n_gpus than asked for."""
    environ = os.environ if environ is None else environ
    argv = sys.argv[1:] if argv is None else argv
    ws = environ.get("WORLD_SIZE")
    if ws is None:
        if args.gpus > 1:
            if environ.get("DLRM_BENCH_NO_SELF_LAUNCH", "0") == "1":
                cmd = self_launch_command(args, argv)
                sys.exit("ERROR: --gpus %d needs one process per GPU; launch as:\n  %s" % (args.gpus, " ".join(cmd)))
            return "exec", self_launch_command(args, argv)
        return "run", 1
    world = int(ws)
    if world != args.gpus:
        print(f"ERROR: --gpus {args.gpus} but the launcher set WORLD_SIZE={world}; refusing to print a line for a GPU count "
              f"that was not asked for", file=sys.stderr)
        sys.exit(2)
    return "run", world


KERNELS_OF_CATEGORY = [      # launch category of dlrm_amd.ops -> substrings of the kernels one of its C-ABI calls launches (headline workload)
    ("emb_bwd_sgd", ["expand_kernel", "seg_hist_kernel", "seg_scan_kernel", "seg_colscan_kernel", "seg_groupscan_kernel", "seg_binscan_kernel",
                     "seg_scatter_kernel", "sorted_update_kernel", "emb_bwd_sgd_", "single_mask_kernel"]),
    ("linear_bwd_weight", ["gemm3_kernel<false, false", "splitk_reduce_kernel", "smallk_wgrad", "gemv_bwd_weight", "gemv_bwd_fused"]),
    ("linear_bwd_data", ["gemm3_kernel<true, false", "gemv_bwd_data"]),
    ("linear_fwd", ["gemm3_kernel<true, true", "gemv_fwd", "pad_cols_kernel"]),
    # (fused lookup + interaction: the fused kernel + the predicated two-kernel form that returns at once — ABI 16, three / two launches)
    ("emb_interact_fwd", ["interact_fwd_dma_kernel", ", true>(EmbArgs"]), ("emb_interact_bwd", ["interact_bwd_dma_kernel"]),
    ("sgd_dense", ["sgd_dense"]), ("bce_loss", ["bce_kernel", "loss_finish_kernel", "scale_kernel"]), ("act_bwd", ["act_bwd_kernel"]),
    ("iota_proof", ["offsets_iota_kernel"])]


def kernel_launches_per_step():
    """KERNEL launches per training step by launch category, counted from the committed rocprofv3 --kernel-trace --stats summary of this
    command (profiles/round*/rocprof_kernel_stats.csv; steps = launches of the dense-SGD kernel) — a C-ABI call is 1-8 kernels (the sorted
    embedding update: expand + two radix rounds of three kernels + the update), which `c_abi_calls_per_step` alone hides."""
    import csv
    import glob
    files = sorted(glob.glob(os.path.join(ROOT, "profiles", "round*", "rocprof_kernel_stats.csv")))
    if not files:
        return None
    rows = list(csv.DictReader(open(files[-1])))
    steps = sum(int(r["Calls"]) for r in rows if "sgd_dense_multi_kernel" in r["Name"])
    if steps <= 0:
        return None
    out, total = {}, 0
    for r in rows:
        for cat, subs in KERNELS_OF_CATEGORY:
            if any(s_ in r["Name"] for s_ in subs):
                out[cat] = out.get(cat, 0) + int(r["Calls"])
                total += int(r["Calls"])
                break
    return {"by_category": {c: round(v / steps, 2) for c, v in out.items()}, "total": round(total / steps, 1),
            "source": os.path.relpath(files[-1], ROOT), "steps_in_profile": steps}


def scaling_model(kernels, ms_step, B, T, D, emb_standalone, link_gbs=153.0, link_eff=0.7, allreduce_bytes=9.48e6):
    """PREDICTED step time at 1 / 2 / 4 / 8 GPUs of one node, made from THIS run's measured single-GPU kernel times and the guide's xGMI link
    rate, BEFORE any multi-GPU measurement exists (no multi-GPU node was reachable in rounds 1-6: SCALE_rNN is `skipped`) — so that the first
    measured curve lands beside a prediction (tools/scaling_model.py is the same arithmetic as a script).
    Strong scaling of the global batch B, the reference's partition (extend_distributed.py:47-62, dlrm_s_pytorch.py:528-585):
      * MLP towers, interaction, loss, dense step: batch-split -> measured time / N (the efficiency loss of N-times shorter GEMMs is NOT modelled:
        optimistic by a few % at N = 8, where M = 8192 rows still fill the chip twice);
      * lookups and the fused sparse update: table-split, the WHOLE batch for ceil(T / N) of T tables (the fused lookup + interaction kernels are
        single-process only: the distributed forward runs dlrm_emb_fwd -> measured stand-alone here);
      * all-to-all of pooled embeddings, each direction: a rank sends (B / N) * T_loc * D * 4 bytes to every peer over its own xGMI link
        (7 links x 153 GB/s per GPU, point to point): time = bytes per peer / (link rate x efficiency); the forward one overlaps the bottom
        MLP (dlrm_s_pytorch.py:563-568), the backward one the bottom tower's backward: only what exceeds them is exposed;
      * DDP all-reduce of the 9.48 MB of MLP gradients: ring over the slowest link, 2 (N - 1) / N x bytes / link rate, overlapped with the backward
        GEMMs -> exposed only beyond them (never at these sizes)."""
    k = {n: v["ms_per_step"] for n, v in kernels.items() if "ms_per_step" in v}
    gemm = k.get("linear_fwd", 0) + k.get("linear_bwd_data", 0) + k.get("linear_bwd_weight", 0)
    other_dense = k.get("act_bwd", 0) + k.get("bce_loss", 0) + k.get("sgd_dense", 0)
    inter = k.get("interact_fwd", 0) + k.get("interact_bwd", 0)
    emb_f = k.get("emb_fwd", 0)
    if "emb_interact_fwd" in k:                       # fused on one GPU: split into a lookup share (what table-sharding scales) and an interaction share
        emb_f = (emb_standalone or {}).get("ms", 0.33)
        inter = max(k["emb_interact_fwd"] + k.get("emb_interact_bwd", 0), 0.0) + 0.25      # + the pooled-buffer round trip of the two-kernel forward (DESIGN: 0.25 ms at B = 65536)
    emb_b = k.get("emb_bwd_sgd", k.get("emb_bwd_adagrad", 0))
    bot_share = 0.0722                                  # bottom tower's share of the tower FLOPs (340 992 of 4 730 368 per sample)
    out = {}
    for N in (1, 2, 4, 8):
        if N == 1:
            out["1"] = {"ms_per_step": ms_step, "samples_per_s": B / ms_step * 1e3, "measured": True}
            continue
        t_loc = -(-T // N)
        dense = (gemm + other_dense + inter) / N
        emb = (emb_f + emb_b) * t_loc / T
        a2a = (B / N) * t_loc * D * 4 / (link_gbs * link_eff * 1e9) * 1e3
        exposed_f = max(a2a - bot_share * k.get("linear_fwd", 0) / N, 0.0)
        exposed_b = max(a2a - bot_share * (k.get("linear_bwd_data", 0) + k.get("linear_bwd_weight", 0)) / N, 0.0)
        allred = 2.0 * (N - 1) / N * allreduce_bytes / (link_gbs * link_eff * 1e9) * 1e3
        exposed_ar = max(allred - (1 - bot_share) * (k.get("linear_bwd_data", 0) + k.get("linear_bwd_weight", 0)) / N, 0.0)
        step = dense + emb + exposed_f + exposed_b + exposed_ar
        out[str(N)] = {"ms_per_step": step, "samples_per_s": B / step * 1e3, "efficiency_vs_linear": ms_step / step / N,
                       "dense_ms": dense, "embedding_ms": emb, "all_to_all_ms_each_way": a2a, "all_to_all_exposed_ms": exposed_f + exposed_b,
                       "allreduce_ms": allred, "tables_per_rank_max": t_loc}
    out["assumptions"] = {"link_gbs": link_gbs, "link_efficiency": link_eff, "scaling": "strong (global batch fixed)",
                          "status": "PREDICTION from single-GPU measurements of this run; no multi-GPU measurement exists yet",
                          "not_modelled": ["shorter GEMMs' efficiency", "table-size imbalance between ranks (the 4-table ranks hold two 40 M-row tables)",
                                           "host launch path (~1 ms per step bounds the step from below at N = 8)"]}
    return out


def main():
    args = parse()
    what, val = resolve_world(args)
    if what == "exec":
        print("[bench] --gpus %d without a launcher: re-executing as %s" % (args.gpus, " ".join(val)), file=sys.stderr, flush=True)
        os.execv(val[0], val)
    if os.environ.get("DLRM_BENCH_WATCHDOG"):       # debugging aid: dump every thread's Python stack after N seconds
        import faulthandler
        faulthandler.dump_traceback_later(int(os.environ["DLRM_BENCH_WATCHDOG"]), repeat=False, exit=False)
    wl = dict(WORKLOADS[args.workload])
    if args.workload == "mlperf_v2_multihot":
        # the reference's configuration for this benchmark: Adagrad lr 0.005 eps 1e-8 (torchrec_dlrm/README.MD:177-194), bf16 MLP
        if "--optimizer" not in " ".join(sys.argv):
            args.optimizer = "rwsadagrad"
        if "--lr" not in " ".join(sys.argv):
            args.lr = 0.005
        if "--mlp-arith" not in " ".join(sys.argv) and "DLRM_MLP_ARITH" not in os.environ:
            args.mlp_arith = "bf16"
        args.no_cpu_baseline = True      # the CPU baseline leg times the headline workload only
        args.no_overlap = True           # 1.4 ms of multi-hot lookups beside 0.15 ms of bottom-MLP GEMMs: nothing to hide (measured)
    if args.batch:
        wl["batch"] = args.batch
    if args.row_cap:
        wl["rows"] = [min(r, args.row_cap) for r in wl["rows"]]
    N = val

    import dlrm_amd
    from dlrm_amd import ext_dist, ops
    from dlrm_amd.optim import FusedRWSAdagrad, FusedSGD

    import faulthandler

    import threading
    partial = {"json": None}      # the finished headline line: printed by the watchdog if a LATER, optional measurement hangs
    wd = {"deadline": None, "what": ""}

    def _watch():
        while True:
            time.sleep(1.0)
            dl = wd["deadline"]
            if dl is not None and time.monotonic() > dl:
                print(f"[bench rank {os.environ.get('RANK', '0')}] watchdog expired: {wd['what']}", file=sys.stderr, flush=True)
                faulthandler.dump_traceback(file=sys.stderr, all_threads=True)
                if partial["json"] is not None:
                    if os.environ.get("RANK", "0") == "0":
                        d_ = json.loads(partial["json"])
                        d_["incomplete"] = "an optional measurement after the headline did not finish (%s); headline value unaffected" % wd["what"]
                        print(json.dumps(d_), flush=True)
                    os._exit(0)
                os._exit(3)

    def watchdog(seconds, what):
        """(re-)arm the hang watchdog (N > 1): when it expires all Python stacks go to stderr and the process exits — with code 3
        while the headline is still being measured, with the finished headline JSON line (rank 0) and code 0 afterwards"""
        if N > 1 and seconds > 0:
            print(f"[bench rank {os.environ.get('RANK', '0')}] watchdog {seconds}s: {what}", file=sys.stderr, flush=True)
            wd["what"], wd["deadline"] = what, time.monotonic() + seconds
        else:
            wd["deadline"] = None

    if N > 1:
        threading.Thread(target=_watch, daemon=True).start()

    if N > 1:
        watchdog(args.hang_timeout, "rendezvous + RCCL all_to_all capability probe")
        # DLRM_BENCH_SELFTEST_GLOO=1 (development only): every rank on cuda:0 over gloo with host-staged exchanges — exercises
        # this file's whole N > 1 control flow on a 1-GPU box; the line it prints is marked and is not a measurement
        selftest = os.environ.get("DLRM_BENCH_SELFTEST_GLOO", "0") == "1"
        if selftest:
            ext_dist.init_distributed(local_rank=0, use_gpu=True, backend="gloo")
        else:
            ext_dist.init_distributed(use_gpu=True, backend="nccl")   # RCCL
        device = torch.device("cuda", ext_dist.my_local_rank)
    else:
        ext_dist.my_size, ext_dist.my_rank = 1, 0
        device = torch.device("cuda", 0)
    torch.cuda.set_device(device)
    rank = max(ext_dist.my_rank, 0)

    rows, D, B = wl["rows"], wl["D"], wl["batch"]
    hot_cfg = wl.get("hot")
    if hot_cfg and args.row_cap:
        wl["hot"] = hot_cfg = [min(h, r) for h, r in zip(hot_cfg, wl["rows"])]
    nf = len(rows) + 1
    ln_top = np.asarray([D + nf * (nf - 1) // 2] + wl["top"])
    np.random.seed(123)          # identical MLP parameters on every rank
    torch.manual_seed(123 + rank)
    dlrm_amd.set_embedding_init(device)
    sharded = bool(hot_cfg) and N > 1
    shard_plan = None
    if sharded:
        # BASELINE configs[4] on N GPUs (torchrec_dlrm/dlrm_main.py:654-673 leaves this to torchrec's planner + DMP):
        # planned sharding — tables nobody can absorb ROW-WISE over all ranks (MLPerf-v2 at 8 ranks: tables 20 and 21), the rest
        # table-wise longest-first —, per-rank input slices exchanged by ext_dist.kjt_input_dist, dot interaction (triu order)
        from dlrm_amd import sharding
        from dlrm_amd.torchrec_variant import ShardedDLRM
        if (args.interaction or "dot") != "dot":
            sys.exit("ERROR: --gpus N --workload mlperf_v2_multihot runs the dot interaction (ShardedDLRM); pass --interaction dot")
        args.interaction = "dot"
        shard_plan = sharding.plan(rows, hot_cfg, D, N, B)
        model = ShardedDLRM(rows, hot_cfg, D, wl["bot"][0], wl["bot"][1:], wl["top"], B, plan=shard_plan).to(device)
    elif hot_cfg:
        # BASELINE configs[4]: the torchrec trainer's model semantics (triu interaction order, logits, BCEWithLogitsLoss)
        from dlrm_amd.torchrec_variant import DLRM as TorchrecDLRM, DLRM_DCN
        if (args.interaction or "dcn") == "dcn":
            model = DLRM_DCN(rows, D, wl["bot"][0], wl["bot"][1:], wl["top"], dcn_num_layers=3, dcn_low_rank_dim=512).to(device)
            ln_top = np.asarray([nf * D] + wl["top"])
        else:
            model = TorchrecDLRM(rows, D, wl["bot"][0], wl["bot"][1:], wl["top"]).to(device)
    else:
        model = dlrm_amd.DLRM_Net(D, np.asarray(rows), np.asarray(wl["bot"]), ln_top, "dot", sigmoid_top=ln_top.size - 2,
                                  loss_function="bce").to(device)
    model.set_mlp_arith(args.mlp_arith)
    model.overlap_streams = bool(args.overlap) and not args.no_overlap
    model.fuse_emb_interact = bool(args.fuse)
    if args.update_in_backward:
        model.update_in_backward = True
    # (what DLRM_Net.sequential_forward checks: dot interaction, one lookup per bag, D = 128, at most 26 tables, single process)
    fused_active = bool(args.fuse) and N == 1 and not hot_cfg and int(D) == 128 and len(rows) + 1 <= 27
    model.a2a_chunks = max(args.a2a_chunks, 1) if (N > 1 and not sharded) else 1
    model.emb_update_mode = {"sorted": ops.UPD_SORTED, "atomic": ops.UPD_ATOMIC, "deterministic": ops.UPD_DETERMINISTIC}[args.emb_update]
    if N > 1:
        wrap = ext_dist.FlatDDP if args.dense_sync == "flat" else ext_dist.DDP
        model.bot_l = wrap(model.bot_l, device_ids=[device.index])
        model.top_l = wrap(model.top_l, device_ids=[device.index])
        groups = [{"params": [p for e in model.emb_l for p in e.parameters()], "lr": args.lr},
                  {"params": model.bot_l.parameters(), "lr": args.lr},
                  {"params": model.top_l.parameters(), "lr": args.lr}]
        opt = FusedSGD(groups, lr=args.lr) if args.optimizer == "sgd" else FusedRWSAdagrad(groups, lr=args.lr)
    else:
        opt = FusedSGD(model.parameters(), lr=args.lr) if args.optimizer == "sgd" else \
            FusedRWSAdagrad(model.parameters(), lr=args.lr, eps=1e-8 if hot_cfg else 1e-10)

    parity = None
    if N == 1 and args.workload == "criteo_terabyte" and not args.no_parity_check:
        parity = parity_check(args, device)
        torch.cuda.empty_cache()
    if N == 1 and args.workload == "mlperf_v2_multihot" and not args.no_parity_check and not args.row_cap and not args.batch:
        parity = parity_check_v2(args, device, args.interaction or "dcn")
        torch.cuda.empty_cache()
    model_a2a_chunks = model.a2a_chunks if N > 1 else 1
    hot = wl.get("hot")
    expand_ms = None
    if hot:
        batches, expand_ms = make_batches(4, B, rows, device, seed=727, hot=hot,
                                          local_rows=ext_dist.get_my_slice(B) if sharded else None)
    else:
        batches = make_batches(4, B, rows, device, seed=727)     # every rank reads the whole global batch (reference :1541)
    # ---- what the steps receive as bag offsets (--offsets).  The fused lookup + interaction path is only valid for one lookup per bag
    # and proves it per offsets tensor OBJECT (ops.offsets_are_iota).  The reference loop hands the module a brand-new tensor every
    # iteration (loader -> dlrm_wrap's .to(device), dlrm_s_pytorch.py:129-145), so the default here does the same: every step gets a
    # fresh, untagged copy made BEFORE the timed region (input production is outside the reference's timed region too) and pays the
    # device pass + stream synchronisation of the proof INSIDE it.
    fresh_off = None
    if N == 1 and not hot and args.offsets != "resident":
        n_fresh = args.warmup + args.steps + 2
        if n_fresh * batches[0][1].numel() * batches[0][1].element_size() > (8 << 30):
            sys.exit("ERROR: --offsets %s with %d steps needs more than 8 GiB of offsets copies; use --offsets resident" % (args.offsets, n_fresh))
        fresh_off = []
        for i in range(n_fresh):
            o_ = batches[i % len(batches)][1].clone()             # a new tensor object: no cached verdict, no tag
            if args.offsets == "tagged":
                ops.mark_one_lookup_per_bag(o_)                    # (a copy of rows dlrm_amd.datagen wrote as 0..B-1: the producer's proof)
            fresh_off.append(o_)
        torch.cuda.synchronize()
    my_rows = ext_dist.get_my_slice(B) if N > 1 else slice(0, B)
    if sharded:
        local_tables = list(model.tw_mine)           # + a 1/N row range of every row-wise table (accounted below)
    else:
        local_tables = list(range(len(rows)))[model.local_emb_slice] if N > 1 else list(range(len(rows)))

    one = torch.ones((), dtype=torch.float32, device=device)      # the seed of backward(): E.backward() would launch an ATen fill for it

    def eager_step(i):
        X, off, idx, T = batches[i % len(batches)]
        if fresh_off is not None and i < len(fresh_off):
            off = fresh_off[i]                         # a NEW offsets tensor object every step (--offsets)
        if sharded:                                   # (X, values, None, T): this rank's samples only
            Z = model(X, off)
            E = model.loss_fn(Z, T)
        else:
            Z = model(X, off, idx)
            E = model.loss_fn(Z, T[my_rows])
        opt.zero_grad()
        E.backward(one)
        opt.step()
        return E

    graphed = None
    if args.graph:
        if N > 1:
            sys.exit("ERROR: --graph is single-process only (RCCL collectives are not captured)")
        from dlrm_amd.graph import GraphedTrainStep
        graphed = GraphedTrainStep(model, opt)
        args.no_kernel_timers = True

    def step(i):
        if graphed is None:
            return eager_step(i)
        X, off, idx, T = batches[i % len(batches)]
        if fresh_off is not None and i < len(fresh_off):
            off = fresh_off[i]
        return graphed(X, off, idx, T)

    dist_info = None
    if N > 1:
        # what the process group really is: rank count as RCCL sees it and one DISTINCT physical GPU per rank
        uuid = str(getattr(torch.cuda.get_device_properties(device), "uuid", "cuda:%d" % device.index))
        uuids = [None] * N
        torch.distributed.all_gather_object(uuids, uuid)
        dist_info = {"backend": torch.distributed.get_backend(), "world_size": torch.distributed.get_world_size(),
                     "distinct_gpus": len(set(uuids)),
                     "tables_per_rank": (shard_plan.tables_per_rank() if sharded else
                                         (list(model.n_emb_per_rank) if model.n_emb_per_rank else None))}
        if sharded:
            dist_info.update({"row_wise_tables": shard_plan.row_wise(), "plan_imbalance": shard_plan.imbalance(),
                              "reference_block_partition_imbalance": _block_partition_imbalance(shard_plan, N),
                              "inputs": "per-rank batch slices; ids exchanged by ext_dist.kjt_input_dist (one all-to-all + one all-gather)"})
        if len(set(uuids)) != N and not selftest:
            sys.exit("ERROR: %d ranks share %d GPUs; one process per GPU is required" % (N, len(set(uuids))))
        watchdog(args.hang_timeout, "first training step (RCCL all-to-all + DDP all-reduce for the first time)")
    calls_per_step = None
    box0 = node0 = None
    if args.calibrate == "first" and not args.no_box_calibration:
        # (the rocm-smi subprocess first — a second of idle queue — then the probes: the warm-up steps start on a busy GPU)
        node0 = node_state() if int(os.environ.get("RANK", "0")) == 0 else None
        box0 = measure_box_or_none(device)
        if box0 is None:
            node0 = None
    for i in range(args.warmup):
        c0_ = ops.CALL_COUNT[0]
        step(i)
        if i <= 1:                  # C-ABI calls of one EAGER step (the graphed step's first two calls are eager too); a call is 1-4 kernels
            calls_per_step = ops.CALL_COUNT[0] - c0_
        if i == 0 and N > 1:
            torch.cuda.synchronize()
            watchdog(args.hang_timeout + 20 * (args.warmup + args.steps), "warm-up + timed region")
    if args.calibrate != "first":
        box0 = measure_box_or_none(device) if not args.no_box_calibration else None
        node0 = node_state() if (box0 is not None and int(os.environ.get("RANK", "0")) == 0) else None
    if not args.no_box_calibration and (args.calibrate != "first" or args.extra_warmup_step):      # (every rank, whether or not ITS probes succeeded: the step contains collectives)
        step(args.warmup)                # one more untimed step: the timed region starts from the step's own steady state, not the probe's
    if N > 1:
        torch.distributed.barrier()
    torch.cuda.synchronize()
    ops.timers = None if args.no_kernel_timers else ops.KernelTimers()
    iota0 = dict(ops.IOTA_STATS)
    if args.gc == "freeze":
        gc.collect(); gc.freeze()
    elif args.gc == "off":
        gc.collect(); gc.disable()
    step_returns = []                     # host time at which every timed step() call returned (diagnostics: host_step_intervals)
    gc_before = [g["collections"] for g in gc.get_stats()]
    t0 = time.perf_counter()
    timed_steps = 0
    for i in range(args.steps):
        if ops.timers is not None:
            ops.timers.enabled = (i % max(args.timer_every, 1) == 0)
            timed_steps += int(ops.timers.enabled)
        loss = step(args.warmup + 1 + i)          # (index args.warmup was the extra untimed step above: every timed step sees a NEW offsets object)
        step_returns.append(time.perf_counter())
    sync_device()
    if N > 1:
        torch.distributed.barrier()
    dt = time.perf_counter() - t0
    if args.gc == "off":
        gc.enable()
    iota = {k: ops.IOTA_STATS[k] - iota0[k] for k in iota0}
    tagged_ms = None
    if fresh_off is not None and args.offsets == "fresh" and graphed is None:
        # the same steps on offsets tensors that carry their producer's proof (what dlrm_amd.datagen / CriteoBinBatches hand out): the
        # difference to the headline is what the per-step device proof of untagged tensors costs; reported beside it, never instead
        keep_timers, ops.timers = ops.timers, None
        fresh_off[:] = [ops.mark_one_lookup_per_bag(batches[i % len(batches)][1].clone()) for i in range(len(fresh_off))]
        for i in range(2):
            step(i)
        torch.cuda.synchronize()
        t1 = time.perf_counter()
        for i in range(args.steps):
            step(2 + i)
        sync_device()
        tagged_ms = (time.perf_counter() - t1) / args.steps * 1e3
        fresh_off = None                               # (every later measurement runs on the resident batches)
        ops.timers = keep_timers
    ksum = ops.timers.summary() if ops.timers is not None else {}
    ops.timers = None
    if N > 1:
        tt = torch.tensor([dt], device=device, dtype=torch.float64)
        torch.distributed.all_reduce(tt, op=torch.distributed.ReduceOp.MAX)
        dt = float(tt.item())
    node1 = node_state() if node0 is not None else None       # (right after the timed region: this GPU still counts among the busy ones)
    box1 = measure_box_or_none(device) if box0 is not None else None
    box = merge_box(box0, box1) if (box0 is not None and box1 is not None) else None
    if box is not None and node0 is not None:
        box["node"] = {"before": node0, "after": node1}
    ms = dt / args.steps * 1e3
    value = B / (dt / args.steps)
    final_loss = float(loss.detach())
    del loss                                      # keeps no autograd graph alive past the timed region

    # ---- algorithmic work per launch -------------------------------
    Bl = B // N if N > 1 else B
    R = 4 * D
    Tl = len(local_tables)
    hots = [hot[t] for t in local_tables] if hot else [1] * Tl
    isz = 4 if hot else 8                               # int32 multi-hot ids (torchrec KJT) / int64 (dlrm_s)
    L = sum(hots)                                       # lookups per sample over the local tables
    if sharded:                                         # row-wise tables: every rank pools the 1/N of the lookups that hit its rows
        L += sum(hot[t] for t in shard_plan.row_wise()) / N
        Tl += len(shard_plan.row_wise())
    emb_fwd_bytes = B * (L * (R + isz) + Tl * (R + isz))    # per lookup: row read + index; per bag: pooled row write + offset
    emb_bwd_bytes = B * L * (3 * R + isz)               # per lookup: dV row read + W row read + W row write + index
    bot, top = list(wl["bot"]), list(ln_top)
    fwd_fl = sum(2.0 * Bl * a * b for ln in (bot, top) for a, b in zip(ln[:-1], ln[1:]))
    wgrad_fl = fwd_fl
    dgrad_fl = fwd_fl - 2.0 * Bl * bot[0] * bot[1]      # the first bottom layer needs no data gradient
    dcn = hot is not None and (args.interaction or "dcn") == "dcn"
    if dcn:                                             # 3 cross layers x two [F*D x 512] products, each forward / dgrad / wgrad
        cross = 3 * 2 * 2.0 * Bl * (nf * D) * 512
        fwd_fl, dgrad_fl, wgrad_fl = fwd_fl + cross, dgrad_fl + cross, wgrad_fl + cross
    F = nf
    inter_bytes = Bl * (F * D * 4 + (D + F * (F - 1) // 2) * 4)
    # fused lookups + interaction (one lookup per bag): forward reads T rows + indices + bag starts + x, writes R; backward reads the
    # same plus dR and writes the (1 + T) gradient rows
    gi_fwd_bytes = B * (Tl * (R + 2 * isz) + D * 4 + (D + F * (F - 1) // 2) * 4)
    gi_bwd_bytes = B * (Tl * (R + 2 * isz) + D * 4 + (D + F * (F - 1) // 2) * 4 + (1 + Tl) * R)

    mfma_peak, mfma_issue_ratio, mfma_measured = ARITH_PEAK[args.mlp_arith]
    hbm_measured = MEASURED_HBM_GBS
    if box is not None:
        mfma_measured = {"f32": box["mfma_f32_tflops"], "bf16": box["mfma_bf16_tflops"], "bf16x6": box["mfma_bf16_tflops"] / 6.0}[args.mlp_arith]
        hbm_measured = box["hbm_copy_gbps"]
    kernels = {}
    for name, work, unit, peak, bound in (
            ("emb_fwd", emb_fwd_bytes, "GB/s", HBM_PEAK_GBS, "hbm"),
            ("emb_bwd_sgd", emb_bwd_bytes, "GB/s", HBM_PEAK_GBS, "hbm"),
            ("emb_bwd_adagrad", emb_bwd_bytes + L * B * 8, "GB/s", HBM_PEAK_GBS, "hbm"),    # + row-wise state read/write per touched row
            ("emb_interact_fwd", gi_fwd_bytes, "GB/s", HBM_PEAK_GBS, "hbm"),
            ("emb_interact_bwd", gi_bwd_bytes, "GB/s", HBM_PEAK_GBS, "hbm"),
            ("interact_fwd", inter_bytes, "GB/s", HBM_PEAK_GBS, "hbm"),
            ("interact_bwd", 2 * inter_bytes, "GB/s", HBM_PEAK_GBS, "hbm"),
            ("linear_fwd", fwd_fl, "TFLOP/s", mfma_peak, "mfma"),
            ("linear_bwd_data", dgrad_fl, "TFLOP/s", mfma_peak, "mfma"),
            ("linear_bwd_weight", wgrad_fl, "TFLOP/s", mfma_peak, "mfma")):
        k = ksum.get(name)
        if not k:
            continue
        per_step_ms = k["total_ms"] / max(timed_steps, 1)
        scale = 1e9 if unit == "GB/s" else 1e12
        ach = work / (per_step_ms * 1e-3) / scale
        kernels[name] = {"ms_per_step": per_step_ms, "launches_per_step": k["calls"] / max(timed_steps, 1),
                         "avg_launch_ms": k["avg_ms"], "bound": bound, "achieved": ach, "peak": peak, "unit": unit,
                         "frac": ach / peak, "frac_of_measured_peak": ach / (hbm_measured if unit == "GB/s" else mfma_measured),
                         "algorithmic_work_per_step": work}
    for name in ("act_bwd", "bce_loss", "sgd_dense", "cross_ew"):
        if name in ksum:
            kernels[name] = {"ms_per_step": ksum[name]["total_ms"] / max(timed_steps, 1),
                             "launches_per_step": ksum[name]["calls"] / max(timed_steps, 1)}
    # ---- counter-byte rates beside the algorithmic ones (SURVEY 8d "report both"): HBM bytes per call from the committed
    # rocprofv3 PMC passes of this workload x the calls of one step / the category's measured time.  For the embedding categories the
    # algorithmic figure counts bytes the caches serve (small tables, duplicate rows), so `achieved` can exceed what the HBM interface
    # moved: `traffic_gbps` is the rate the memory system really sustained and `frac_traffic` prices THAT against the 8 TB/s peak.
    # Nothing is capped: an algorithmic frac_of_measured_peak above 1 means cache hits, and the traffic columns show it.
    pmc = load_pmc_traffic() if (N == 1 and args.workload == "criteo_terabyte" and not args.batch and not args.row_cap) else None
    for name, k in kernels.items():
        t = pmc["kernels"].get(name) if pmc else None
        if t and not t.get("stale") and t.get("traffic_bytes") and "ms_per_step" in k:
            per_step = t["traffic_bytes"] * t.get("calls_per_step", 1)
            k["traffic_bytes_per_step"] = per_step
            k["traffic_gbps"] = per_step / (k["ms_per_step"] * 1e-3) / 1e9
            k["frac_traffic"] = k["traffic_gbps"] / HBM_PEAK_GBS
            k["frac_traffic_of_measured_peak"] = k["traffic_gbps"] / hbm_measured
    heavy = [n for n in kernels if "achieved" in kernels[n]]
    dom = max(heavy, key=lambda n: kernels[n]["ms_per_step"]) if heavy else None
    kname = {"linear_fwd": "gemm3_kernel<KC,KC> (Y = X*W^T + bias, ReLU, sign bits; 128x128x16 tiles, two-stage LDS-DMA ring = 4 workgroups per CU, straight-line epilogue)",
             "linear_bwd_data": "gemm3_kernel<KC,KS> (dX = dY*W, previous layer's ReLU derivative from sign bits in the straight-line epilogue; 128x128x16 tiles, 4 workgroups per CU)",
             "linear_bwd_weight": "gemm3_kernel<KS,KS> (dW = dY^T*X split over the batch into slabs + bias-grad row sums; 256x128x16 tiles at one round of 512 workgroups for the wide layers) + splitk_reduce_kernel",
             "emb_bwd_adagrad": "expand + lookup sort (seg_sort.h, or rocPRIM for segments > 262144 lookups) + adagrad_groups_kernel + adagrad_fixup_kernel",
             "emb_fwd": "emb_fwd_kernel", "emb_bwd_sgd": "expand + segmented radix sort (seg_hist / seg_scan / seg_scatter per round, csrc/seg_sort.h) + sorted_update_kernel" if args.emb_update == "sorted" else "emb_bwd_sgd_{atomic,lds}_kernel",
             "interact_fwd": "interact_fwd_dma_kernel", "interact_bwd": "interact_bwd_dma_kernel",
             "emb_interact_fwd": "interact_fwd_dma_kernel<gather>: one-hot embedding lookups fetched by the interaction kernel (K1 + K6 fused)",
             "emb_interact_bwd": "interact_bwd_dma_kernel<gather>"}
    if hot:
        result_extra = {"multihot": {"lookups_per_sample": L, "index_dtype": "int32", "lookup_table_bytes": int(sum(4 * r * h for r, h in zip(rows, hot))),
                                     "expand_ms_per_batch": sum(expand_ms) / len(expand_ms),
                                     "expand_gbps": 8.0 * L * B / (sum(expand_ms) / len(expand_ms) * 1e-3) / 1e9,
                                     "note": "1-hot -> multi-hot expansion on the device (dlrm_multihot_expand, 8 B per produced id); done "
                                             "by the data pipeline outside the training step, as in the reference (multi_hot.py)"}}
    else:
        result_extra = {}

    mode_of = {"sorted": ops.UPD_SORTED, "atomic": ops.UPD_ATOMIC, "deterministic": ops.UPD_DETERMINISTIC}
    lookup_sort = None
    if args.emb_update == "sorted" or args.optimizer == "rwsadagrad":
        try:
            if sharded:
                lookup_sort = "per shard (csrc/seg_sort.h where a shard's segments hold <= 262144 lookups)"
            else:
                o_, i_ = batches[0][1], batches[0][2]
                if N > 1:                                     # table-wise shards: this rank sorts the lookups of its own tables only
                    o_, i_ = o_[model.local_emb_slice], i_[model.local_emb_slice]
                own = ops.sort_is_graph_safe([e.weight for e in model.emb_l], ops.BagBatch(o_, i_))
                lookup_sort = ("segmented radix sort of csrc/seg_sort.h (the library's own kernels, HIP-graph replayable)" if own else
                               "rocPRIM radix_sort_pairs (a table segment exceeds 262144 lookups, or DLRM_SORT=rocprim)")
        except Exception as e:                            # noqa: BLE001 - a label, never allowed to break the line
            lookup_sort = "unknown (%s)" % type(e).__name__

    def roof(n):
        k = kernels[n]
        t = pmc["kernels"].get(n) if pmc else None
        stale = bool(t and t.get("stale"))
        if stale:
            note = "stale: %s was measured on an older version of this kernel's sources (re-run tools/gpu_prof.sh + tools/pmc_to_json.py)" % pmc["_file"]
        elif t and k["unit"] == "GB/s":
            note = ("HBM bytes per call from rocprofv3 PMC passes of this workload (%s; FETCH_SIZE x2 gfx950 correction + "
                    "WRITE_SIZE; source hashes match HEAD), algorithmic bytes per call = %d" %
                    (pmc["_file"], k["algorithmic_work_per_step"] // max(int(round(k["launches_per_step"])), 1)))
        else:
            note = ("HBM bytes per call from rocprofv3 PMC passes (%s; source hashes match HEAD)" % pmc["_file"]) if t else None
        measured_peak = hbm_measured if k["unit"] == "GB/s" else mfma_measured
        # the whole per-category table travels INSIDE `roofline` (the driver's record keeps this object): [ms per step, frac of the
        # spec peak, frac of this box's measured peak]; categories without an algorithmic-work figure carry None fractions
        by_cat = {c: [round(v["ms_per_step"], 4), round(v["frac"], 4) if "frac" in v else None,
                      round(v["frac_of_measured_peak"], 4) if "frac_of_measured_peak" in v else None,
                      round(v["traffic_gbps"], 1) if "traffic_gbps" in v else None,
                      round(v["frac_traffic"], 4) if "frac_traffic" in v else None] for c, v in kernels.items()}
        return {"kernel": kname.get(n, n), "bound": k["bound"], "achieved": k["achieved"], "peak": k["peak"],
                "peak_note": None if k["unit"] == "GB/s" else
                {"f32": "dense fp32 MFMA peak", "bf16": "dense bf16 MFMA peak (v_mfma_f32_32x32x16_bf16)",
                 "bf16x6": "dense bf16 MFMA peak / 6: six bf16 products are issued per fp32 product"}[args.mlp_arith],
                "unit": k["unit"], "frac": k["frac"],
                "frac_of_measured_peak": k["achieved"] / measured_peak, "measured_peak": measured_peak,
                "traffic": t["traffic_bytes"] if (t and not stale) else None, "traffic_note": note,
                "avg_launch_ms": k["avg_launch_ms"], "ms_per_step": k["ms_per_step"],
                "traffic_gbps": k.get("traffic_gbps"), "frac_traffic": k.get("frac_traffic"),
                "by_category": by_cat,
                "by_category_columns": ["ms_per_step", "frac (algorithmic work / spec peak)", "frac_of_measured_peak",
                                        "traffic_gbps (PMC HBM bytes / time)", "frac_traffic (traffic_gbps / 8000)"],
                "embedding_hbm_gbps": emb_gbps, "box": box}

    klaunch = kernel_launches_per_step() if (N == 1 and args.workload == "criteo_terabyte" and graphed is None) else None
    for name, k in kernels.items():
        k["c_abi_calls_per_step"] = k.get("launches_per_step")          # ("launches_per_step" stays as the older name of the same number: C-ABI calls)
        k["kernel_launches_per_step"] = (klaunch or {}).get("by_category", {}).get(name)
    cat_sum = sum(k["ms_per_step"] for k in kernels.values() if "ms_per_step" in k)
    emb_gbps = {"fwd": kernels.get("emb_fwd", {}).get("achieved"), "bwd_sgd": kernels.get("emb_bwd_sgd", {}).get("achieved"),
                "fwd_fused_with_interaction": kernels.get("emb_interact_fwd", {}).get("achieved"),
                "fwd_traffic": kernels.get("emb_interact_fwd", kernels.get("emb_fwd", {})).get("traffic_gbps"),
                "bwd_sgd_traffic": kernels.get("emb_bwd_sgd", {}).get("traffic_gbps"),
                "note": "BASELINE.json's second metric.  fwd / bwd_sgd / fwd_fused_with_interaction: algorithmic bytes (SURVEY 8d) / kernel time; "
                        "*_traffic: rocprofv3 PMC HBM bytes / kernel time (cache hits excluded)"}
    result = {
        "metric": "samples/sec (global batch) + embedding HBM GB/s, Criteo-TB config",
        "value": value, "unit": "samples/s", "n_gpus": N, "steps": args.steps, "warmup": args.warmup,
        "ms_per_step": ms, "higher_is_better": True, "scaling": "strong", "vs_baseline": None,
        "dtype": "bf16" if args.mlp_arith == "bf16" else "f32",
        "data": "synthetic (the reference's random generator distributions, one lookup per bag, produced on the device by "
                "dlrm_amd.datagen; random-init parameters)",
        "config": {"workload": args.workload + {"criteo_terabyte": ": MLPerf Criteo-Terabyte shapes (BASELINE.json configs[2])",
                                                "mlperf_v2_multihot": ": MLPerf-v2 multi-hot synthetic inputs, 214 lookups/sample (BASELINE.json "
                                                                      "configs[4]: torchrec model semantics — DCN-v2 (3 layers, rank 512) or triu dot interaction, logits + "
                                                                      "BCEWithLogits, fused row-wise Adagrad) — NOT the headline"}.get(args.workload, ""),
                   "tables": len(rows), "emb_dim": D, "global_batch": B, "table_rows_total": int(sum(rows)),
                   "mlp_bot": "-".join(map(str, bot)), "mlp_top": "-".join(map(str, top)), "optimizer": args.optimizer,
                   "interaction": ("dcn_v2 (3 layers, rank 512)" if dcn else "dot (torchrec triu order)") if hot else "dot",
                   "loss": "bce_with_logits" if hot else "bce", "parallelism": (("planned sharding x%d (dlrm_amd.sharding: row-wise tables %s, the rest table-wise longest-first) + data-parallel MLPs (dense gradients: %s), per-rank input slices"
                                    % (N, shard_plan.row_wise(), args.dense_sync)) if sharded else
                                   ("table-wise embeddings x%d + data-parallel MLPs (dense gradients: %s)" % (N, args.dense_sync))) if N > 1 else "single GPU",
                   "embedding_update": (args.emb_update if model.emb_update_mode == mode_of[args.emb_update] else
                                        "atomic (the HIP-graph path fell back: a table segment needs the general sorter, which cannot be replayed; dlrm_amd/graph.py)"),
                   "lookup_sort": lookup_sort,
                   "sparse_update_schedule": ("single-lookup rows inside the fused backward, the rest at optimizer.step() (--update-in-backward, ABI 17)"
                                              if (getattr(model, "update_in_backward", False) and fused_active) else "every row at optimizer.step() (the reference loop's)"),
                   "a2a_chunks": model_a2a_chunks,
                   "embedding_interaction": ("fused: the interaction kernels gather the one-hot embedding rows themselves "
                                             "(no pooled-embedding buffer)") if fused_active else "two kernels (dlrm_emb_fwd, dlrm_interact_*)",
                   "streams": ("single stream" if (not args.overlap or args.no_overlap or graphed is not None) else
                               "2 HIP streams: embedding lookups / fused sparse update on a side stream beside the bottom-MLP GEMMs "
                               "(per-kernel event times then overlap: their sum exceeds the step time)"),
                   "launch": "HIP graph replay of the captured step (dlrm_amd.graph)" if graphed is not None else "eager (one C-ABI call per kernel)",
                   "mlp_arith": {"f32": "f32: native fp32 MFMA (v_mfma_f32_32x32x2_f32)",
                                 "bf16x6": "bf16x6: fp32 operands as an exact 3-term bf16 split (pre-split planes through dlrm_gemm_bf16x6 where its shapes hold, else split in the k-loop), 6 bf16 MFMA products, fp32 accumulate",
                                 "bf16": "bf16: forward / data-gradient GEMMs read bf16 copies of activations and weights (dlrm_gemm_bf16, nothing converted "
                                         "in the k-loop; activations written in fp32 + bf16), weight gradient rounds its fp32 operands in the loop; one "
                                         "v_mfma_f32_32x32x16_bf16 per 16 k, fp32 accumulate, fp32 master weights (reduced precision: NOT the headline "
                                         "configuration)"}[args.mlp_arith]},
        "final_loss": final_loss,
        "launches": {"c_abi_calls_per_step": calls_per_step, "kernel_launches_per_step": klaunch, "us_per_step": ms * 1e3,
                     "note": "categorised C-ABI calls of one eager step (one call = 1-4 kernel launches; rocprofv3 kernel counts per step: "
                             "profiles/round6/step_trace.txt — 52 at Criteo-Terabyte shapes —, profiles/round5/step_trace_kaggle_graph_towers_v5.txt — 19 at "
                             "Criteo-Kaggle shapes with the small-batch tower kernels); the whole-step HIP graph (--graph) replays them with one launch"},
        "box": box,
        "parity_check": parity,
        "kernel_timing": "HIP events on the launch stream, one per change of launch category (a run of consecutive launches of one category "
                         "is one event pair; dlrm_amd.ops.KernelTimers), on %d of the %d timed steps.  An event is a queue barrier (~5 us) charged to "
                         "the category it closes: the categories sum to %.3f ms against the %.3f ms step (+%.1f %%) — every frac of this line is "
                         "that much pessimistic, none optimistic" % (timed_steps, args.steps, cat_sum, ms, (cat_sum / ms - 1) * 100 if ms else 0),
        "roofline": roof(dom) if dom else None,
        "roofline_embedding": roof("emb_fwd") if "emb_fwd" in kernels else None,
        "embedding_hbm_gbps": emb_gbps,
        "iota_proof": None if (N > 1 or hot) else {
            "offsets": args.offsets, "verdict": "device predicate (ABI 16)" if dlrm_amd.dlrm_net.DEVICE_PREDICATE else "host proof (DLRM_DEVICE_PREDICATE=0)",
            "device_predicates_in_timed_region": iota.get("device_predicates", 0), "host_proofs_in_timed_region": iota["checked"],
            "tagged": iota["tagged"], "cached": iota["cached"],
            "launch_us_per_step": iota["host_us"] / max(args.steps, 1), "host_wait_us_per_step": iota["wait_us"] / max(args.steps, 1),
            "ms_per_step_with_producer_tagged_offsets": tagged_ms,
            "iota_proof_us_per_step": None if tagged_ms is None else (ms - tagged_ms) * 1e3,
            "note": "with --offsets fresh (default) every timed step hands the module a NEW untagged offsets tensor, as the reference loop does "
                    "(dlrm_s_pytorch.py:129-145): whether it holds one lookup per bag is counted by a device pass INSIDE the timed region "
                    "(dlrm_offsets_iota_flags, on the step's stream) and the verdict stays on the device — the fused lookup + interaction "
                    "kernels and the two-kernel form are both enqueued behind that launch predicate (three launches per step return at once) "
                    "and the host never waits (rounds 4-6 had the host wait for the verdict once per step: 30-110 us on a quiet box, "
                    "1.25 ms per step on a bad day — profiles/round6/proof_wait.md).  iota_proof_us_per_step = headline ms_per_step - the same "
                    "steps re-run on producer-tagged offsets (no pass, no predicate); the second leg of a process runs warmer, so it is an "
                    "upper bound of what the proof and the three empty launches cost"},
        "host_step_intervals": host_step_intervals(t0, step_returns, gc_before),
        "kernels": kernels,
    }
    result.update(result_extra)
    partial["json"] = json.dumps(result)            # from here on a hang in an optional measurement cannot lose the headline
    if N == 1 and "emb_fwd" not in kernels and not args.no_standalone_emb:
        # fused forward: no separate embedding kernel ran inside the step.  BASELINE.json's second metric is the embedding kernel's
        # HBM rate, so the stand-alone dlrm_emb_fwd (what the unfused / multi-hot / distributed paths launch) is measured here.
        from dlrm_amd.ops import BagBatch, emb_fwd
        X0, off0, idx0, _ = batches[0]
        bags0 = BagBatch(off0, idx0)
        out0 = torch.empty((B, len(rows) * D), dtype=torch.float32, device=device)
        ws0 = [e.weight for e in model.emb_l]
        for _ in range(2):
            emb_fwd(ws0, bags0, out0)
        ea, eb = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        ea.record()
        for _ in range(5):
            emb_fwd(ws0, bags0, out0)
        eb.record()
        torch.cuda.synchronize()
        ems = ea.elapsed_time(eb) / 5
        ach = emb_fwd_bytes / (ems * 1e-3) / 1e9
        result["embedding_kernel_standalone"] = {"kernel": "emb_fwd_kernel (dlrm_emb_fwd, all tables in one launch), measured outside the step: "
                                                           "the step itself runs the lookups inside the interaction kernels",
                                                 "ms": ems, "achieved": ach, "unit": "GB/s", "peak": HBM_PEAK_GBS, "frac": ach / HBM_PEAK_GBS,
                                                 "frac_of_measured_peak": ach / hbm_measured, "algorithmic_bytes": emb_fwd_bytes}
        emb_gbps["fwd"] = ach
        if pmc and (pmc["kernels"].get("emb_fwd") or {}).get("traffic_bytes") and not pmc["kernels"]["emb_fwd"].get("stale"):
            tb_ = pmc["kernels"]["emb_fwd"]["traffic_bytes"]
            result["embedding_kernel_standalone"].update({"traffic_bytes": tb_, "traffic_gbps": tb_ / (ems * 1e-3) / 1e9,
                                                          "frac_traffic": tb_ / (ems * 1e-3) / 1e9 / HBM_PEAK_GBS})
            emb_gbps["fwd_standalone_traffic"] = tb_ / (ems * 1e-3) / 1e9
        partial["json"] = json.dumps(result)
        del out0
    if N == 1 and fused_active and graphed is None and not args.no_high_row_check:
        try:
            result["high_row_check"] = high_row_check(model, D, device)
        except Exception as e:                              # noqa: BLE001 - a report beside the headline
            result["high_row_check"] = {"ok": False, "error": "%s: %s" % (type(e).__name__, str(e)[:300])}
        partial["json"] = json.dumps(result)
    if N == 1 and graphed is None and not hot and not args.no_reference_region:
        # ---- the reference's FULL timed region ("stock_gpu_baseline pays both, ours excludes both").  Between its time_wrap
        # calls (dlrm_s_pytorch.py:1558,1626) the reference loop also moves the batch to the device (dlrm_wrap :129-145: X / lS_o / lS_i .to(device)
        # from the loader's host tensors; T in loss_fn_wrap :150-156) and reads the loss back (`L = E.detach().cpu().numpy()`, :1592, before
        # backward).  The headline keeps inputs resident (the contract of this bench; f-2 moved the producer to the device) and never reads
        # the loss; THIS leg runs the same steps with both inside the timing, from pageable host tensors as a DataLoader hands them over:
        # what the headline omits is a number in the line, and stock_gpu_baseline has a like-for-like partner.
        try:
            keep_timers, ops.timers = ops.timers, None
            host = [tuple(t.cpu() for t in b) for b in batches]
            nref = max(min(args.steps, 10), 3)

            def ref_region_step(i):
                Xh, oh, ih, Th = host[i % len(host)]
                X, off, idx = Xh.to(device), oh.to(device), ih.to(device)          # dlrm_wrap
                Z = model(X, off, idx)
                E = model.loss_fn(Z, Th.to(device))                                  # loss_fn_wrap
                L = E.detach().cpu().numpy()                                         # :1592
                opt.zero_grad()
                E.backward(one)
                opt.step()
                return float(L)
            for i in range(2):
                ref_region_step(i)
            torch.cuda.synchronize()
            t0r = time.perf_counter()
            for i in range(nref):
                lr_ = ref_region_step(2 + i)
            torch.cuda.synchronize()
            dtr = (time.perf_counter() - t0r) / nref
            h2d = sum(t.numel() * t.element_size() for t in host[0])
            result["reference_timed_region"] = {
                "ms_per_step": dtr * 1e3, "value": B / dtr, "unit": "samples/s", "steps": nref, "final_loss": lr_,
                "h2d_bytes_per_step": h2d, "headline_ms_per_step": ms, "cost_of_what_the_headline_omits_ms": dtr * 1e3 - ms,
                "what": "the headline's steps with the reference loop's per-step host work INSIDE the timing: four .to(device) copies of a "
                        "pageable host batch (dlrm_wrap / loss_fn_wrap, dlrm_s_pytorch.py:129-156) and the loss read-back before backward "
                        "(:1592, a host synchronisation); every step therefore also gets a new untagged offsets tensor (proof inside).  "
                        "stock_gpu_baseline times the unmodified reference the same way: value / stock_gpu_baseline.value is like for like"}
            ops.timers = keep_timers
            del host
        except Exception as e:                              # noqa: BLE001 - a report beside the headline
            result["reference_timed_region"] = {"error": "%s: %s" % (type(e).__name__, str(e)[:300])}
        partial["json"] = json.dumps(result)
    if N == 1 and graphed is None and not hot and args.optimizer == "sgd" and args.mlp_arith == "f32" and not args.no_full_size_parity and not args.row_cap:
        try:
            keep_timers, ops.timers = ops.timers, None
            result["full_size_parity"] = full_size_parity(model, opt, wl, ln_top, batches, args.lr, device)
            ops.timers = keep_timers
        except Exception as e:                              # noqa: BLE001 - a report beside the headline
            result["full_size_parity"] = {"error": "%s: %s" % (type(e).__name__, str(e)[:300])}
            torch.cuda.empty_cache()
        partial["json"] = json.dumps(result)
    if N == 1 and not hot:
        try:
            result["scaling_model"] = scaling_model(kernels, ms, B, len(rows), D, result.get("embedding_kernel_standalone"))
        except Exception as e:                              # noqa: BLE001
            result["scaling_model"] = {"error": "%s: %s" % (type(e).__name__, str(e)[:200])}
        partial["json"] = json.dumps(result)
    if N == 1 and not args.no_rccl_selfcheck:
        result["rccl_selfcheck"] = rccl_selfcheck()
        partial["json"] = json.dumps(result)
    if args.alts and N == 1 and args.mlp_arith == "f32" and not args.no_alt_arith:
        # the same step with the opt-in bf16x6 MLP arithmetic (fp32 round-off class, see include/dlrm_hip.h): reported
        # beside the headline value, never instead of it
        model.set_mlp_arith("bf16x6")
        for i in range(2):
            step(i)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for i in range(args.steps):
            loss_alt = step(i)
        torch.cuda.synchronize()
        dta = (time.perf_counter() - t0) / args.steps
        model.set_mlp_arith("f32")
        result["alt_mlp_arith"] = {"mlp_arith": "bf16x6", "value": B / dta, "unit": "samples/s", "ms_per_step": dta * 1e3,
                                   "final_loss": float(loss_alt.detach()),
                                   "note": "opt-in (--mlp-arith bf16x6); parity-tested at the fp32 tolerances; GEMM layers on pre-split bf16 planes (csrc/gemm_bf16.hip PL = 3; "
                                           "DLRM_BF16X6_PLANES=0: split inside every k-loop, bit-identical products)"}
        del loss_alt
    if args.alts and N == 1 and graphed is None and not hot and args.fuse and getattr(model, "fuse_emb_interact", False) and not args.no_alt_fuse:
        # the same step with the lookups and the interaction as two kernels (rounds 1-2's forward; what multi-hot and distributed runs launch)
        model.fuse_emb_interact = False
        for i in range(3):
            step(i)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for i in range(args.steps):
            loss_u = step(i)
        torch.cuda.synchronize()
        dtu = (time.perf_counter() - t0) / args.steps
        model.fuse_emb_interact = True
        result["alt_two_kernel_lookup"] = {"value": B / dtu, "unit": "samples/s", "ms_per_step": dtu * 1e3, "final_loss": float(loss_u.detach()),
                                           "note": "--no-fuse: dlrm_emb_fwd + dlrm_interact_fwd / _bwd through the pooled-embedding buffer; bit-identical results"}
        del loss_u
    if (N == 1 and graphed is None and not hot and fused_active and args.optimizer == "sgd" and args.emb_update == "sorted"
            and not getattr(model, "update_in_backward", False) and not args.no_alt_update_in_backward):
        # the same step with the opt-in schedule of ABI 17 (DLRM_Net.update_in_backward): single-lookup rows take their SGD step inside the
        # fused backward; beside the headline, never instead of it.  Same final tables as the headline's schedule (tests/test_gpu_kernels.py)
        model.update_in_backward = True
        for i in range(3):
            step(i)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for i in range(args.steps):
            loss_b = step(i)
        torch.cuda.synchronize()
        dtb = (time.perf_counter() - t0) / args.steps
        model.update_in_backward = False
        result["alt_update_in_backward"] = {"value": B / dtb, "unit": "samples/s", "ms_per_step": dtb * 1e3, "final_loss": float(loss_b.detach()),
                                            "note": "opt-in (DLRM_Net.update_in_backward / --update-in-backward; ABI 17: dlrm_emb_presort + "
                                                    "dlrm_interact_bwd_gather_sgd + dlrm_emb_bwd_sgd_presorted): the rows one lookup of the batch names are "
                                                    "stepped inside the fused backward, the others at optimizer.step(); the tables after a step hold the "
                                                    "same values, WHEN single-lookup rows change differs from the reference loop"}
        del loss_b
        partial["json"] = json.dumps(result)
    if args.alts and N == 1 and graphed is None and not hot and not (args.overlap and not args.no_overlap) and not args.no_alt_overlap:
        # the same step on two HIP streams (embedding kernels beside the bottom-MLP GEMMs): beside the headline, never instead
        model.overlap_streams = True
        for i in range(3):
            step(i)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for i in range(args.steps):
            loss_o = step(i)
        torch.cuda.synchronize()
        dto = (time.perf_counter() - t0) / args.steps
        model.overlap_streams = False
        model._join_side_stream()
        result["alt_stream_overlap"] = {"value": B / dto, "unit": "samples/s", "ms_per_step": dto * 1e3, "final_loss": float(loss_o.detach()),
                                        "note": "DLRM_Net.overlap_streams: pooled lookups beside the bottom-MLP forward GEMMs, fused sparse "
                                                "update beside the bottom-MLP backward + dense step; same kernels, same results"}
        del loss_o
    if N == 1 and graphed is None and args.alt_graph:
        # the same step captured once in a HIP graph and replayed (dlrm_amd.graph): removes the host launch path; reported
        # beside the headline value.  Never allowed to break the headline line.
        try:
            from dlrm_amd.graph import GraphedTrainStep
            gs = GraphedTrainStep(model, opt)
            for i in range(4):                       # 2 eager warm-up calls, capture, first replays
                X, off, idx, T = batches[i % len(batches)]
                gs(X, off, idx, T)
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for i in range(args.steps):
                X, off, idx, T = batches[i % len(batches)]
                loss_g = gs(X, off, idx, T)
            torch.cuda.synchronize()
            dtg = (time.perf_counter() - t0) / args.steps
            result["alt_hip_graph"] = {"value": B / dtg, "unit": "samples/s", "ms_per_step": dtg * 1e3,
                                       "final_loss": float(loss_g), "captures": gs.captures,
                                       "note": "whole step (fwd+loss+bwd+fused updates) replayed as one HIP graph; inputs copied "
                                               "into static buffers each step"}
            del gs
        except Exception as e:                       # noqa: BLE001 - diagnostic only
            result["alt_hip_graph"] = {"error": "%s: %s" % (type(e).__name__, str(e)[:300])}
    if N > 1 and sharded:
        result["distributed"] = dist_info
        if selftest:
            result["selftest"] = "DLRM_BENCH_SELFTEST_GLOO=1: all ranks on ONE GPU over gloo with host-staged exchanges — control-flow test, NOT a measurement"
        watchdog(0, "")
    elif N > 1:
        # ---- the OTHER exchange schedule, same process, same model: never instead of the headline ------------------------
        alt_c = 1 if model_a2a_chunks > 1 else (args.alt_a2a_chunks or {2: 4, 4: 2, 8: 2}.get(N, 2))
        if args.alts and (alt_c == 1 or (B // N) % alt_c == 0):
            watchdog(args.hang_timeout + 20 * args.steps, "alternative all-to-all schedule (%d chunk(s))" % alt_c)
            model.a2a_chunks = alt_c
            for i in range(2):
                step(i)
            torch.distributed.barrier()
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for i in range(args.steps):
                loss_alt = step(i)
            torch.cuda.synchronize()
            torch.distributed.barrier()
            tt = torch.tensor([time.perf_counter() - t0], device=device, dtype=torch.float64)
            torch.distributed.all_reduce(tt, op=torch.distributed.ReduceOp.MAX)
            dta = float(tt.item()) / args.steps
            result["alt_a2a_pipelined" if alt_c > 1 else "alt_a2a_reference"] = {
                "a2a_chunks": alt_c, "value": B / dta, "unit": "samples/s", "ms_per_step": dta * 1e3, "final_loss": float(loss_alt.detach()),
                "note": ("pooled-embedding all-to-all pipelined in %d batch chunks (DLRM_Net._pipelined_exchange_forward)" % alt_c)
                        if alt_c > 1 else "the reference schedule: one all-to-all per direction, bottom MLP overlapping it"}
            del loss_alt
            model.a2a_chunks = model_a2a_chunks
        # ---- the OTHER dense-gradient synchronisation, same process, same parameters -------------------------------------------
        if args.alts and not hot:
            try:
                other = "flat" if args.dense_sync == "ddp" else "ddp"
                watchdog(args.hang_timeout + 20 * args.steps, "alternative dense-gradient all-reduce (%s)" % other)
                if os.environ.get("DLRM_BENCH_SELFTEST_HANG") == "alt":        # development: proves the watchdog still prints the headline
                    time.sleep(10 ** 6)
                inner_b, inner_t = model.bot_l.module, model.top_l.module
                for w_ in (model.bot_l, model.top_l):
                    if hasattr(w_, "release"):
                        w_.release()                           # FlatDDP: hooks and flat-buffer slots leave the parameters
                model.bot_l = model.top_l = None
                gc.collect()                                   # the old wrappers (their autograd hooks) go away
                if other == "flat":
                    model.bot_l, model.top_l = ext_dist.FlatDDP(inner_b), ext_dist.FlatDDP(inner_t)
                else:
                    model.bot_l = ext_dist.DDP(inner_b, device_ids=[device.index])
                    model.top_l = ext_dist.DDP(inner_t, device_ids=[device.index])
                for i in range(2):
                    step(i)
                torch.distributed.barrier()
                torch.cuda.synchronize()
                t0 = time.perf_counter()
                for i in range(args.steps):
                    loss_alt = step(i)
                torch.cuda.synchronize()
                torch.distributed.barrier()
                tt = torch.tensor([time.perf_counter() - t0], device=device, dtype=torch.float64)
                torch.distributed.all_reduce(tt, op=torch.distributed.ReduceOp.MAX)
                dtf = float(tt.item()) / args.steps
                result["alt_dense_sync"] = {
                    "dense_sync": other, "value": B / dtf, "unit": "samples/s", "ms_per_step": dtf * 1e3, "final_loss": float(loss_alt.detach()),
                    "note": "ext_dist.FlatDDP: the weight-gradient GEMMs write into one flat buffer per tower, one in-place all-reduce launched "
                            "when the tower's last gradient is ready, no bucket copies" if other == "flat" else "torch DistributedDataParallel"}
                del loss_alt
            except Exception as e:                       # noqa: BLE001 - never allowed to break the headline line
                result["alt_dense_sync"] = {"error": "%s: %s" % (type(e).__name__, str(e)[:300])}
        # ---- the collectives alone, at the step's exact sizes, HIP events on the stream they are enqueued on ---------------
        try:
            watchdog(args.hang_timeout, "collective micro-measurements")
            Tl_, D_ = len(local_tables), D
            splits = model.n_emb_per_rank or [Tl_] * N
            send = torch.empty(B * Tl_ * D_, device=device)
            recv = torch.empty((B // N) * sum(splits) * D_, device=device)
            send_counts = [(B // N) * Tl_ * D_] * N
            recv_counts = [(B // N) * t * D_ for t in splits]
            flat = torch.empty(sum(p.numel() for p in model.bot_l.parameters()) + sum(p.numel() for p in model.top_l.parameters()), device=device)

            def timed(fn, iters=10):
                for _ in range(3):
                    fn()
                torch.cuda.synchronize()
                torch.distributed.barrier()
                a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                a.record()
                for _ in range(iters):
                    fn()
                b.record()
                torch.cuda.synchronize()
                t = torch.tensor([a.elapsed_time(b) / iters], device=device, dtype=torch.float64)
                torch.distributed.all_reduce(t, op=torch.distributed.ReduceOp.MAX)
                return float(t.item())
            a2a_f = timed(lambda: torch.distributed.all_to_all_single(recv, send, recv_counts, send_counts))
            a2a_b = timed(lambda: torch.distributed.all_to_all_single(send, recv, send_counts, recv_counts))
            ar = timed(lambda: torch.distributed.all_reduce(flat))
            off_node = send.numel() * 4 * (N - 1) / N
            result["collectives"] = {
                "a2a_fwd_ms": a2a_f, "a2a_bwd_ms": a2a_b, "allreduce_ms": ar, "timing": "HIP events, max over ranks, 10 back-to-back calls",
                "a2a_send_bytes_per_rank": send.numel() * 4, "a2a_bytes_leaving_rank": off_node,
                "a2a_fwd_gbps_per_rank": off_node / (a2a_f * 1e-3) / 1e9, "allreduce_bytes": flat.numel() * 4,
                "in_step": "2 all-to-alls (forward + backward) + the dense-gradient all-reduce of each tower (--dense-sync %s) per step" % args.dense_sync}
        except Exception as e:                           # noqa: BLE001 - diagnostic section, never breaks the headline line
            result["collectives"] = {"error": "%s: %s" % (type(e).__name__, str(e)[:300])}
        result["distributed"] = dist_info
        if selftest:
            result["selftest"] = "DLRM_BENCH_SELFTEST_GLOO=1: all ranks on ONE GPU over gloo with host-staged exchanges — control-flow test, NOT a measurement"
        watchdog(0, "")
    if rank == 0 and N == 1 and not args.no_cpu_baseline:
        state = baseline_state(model, batches[0], wl, ln_top, args)
        del model, opt, batches
        torch.cuda.empty_cache()
        result.update(baseline_legs(state, args, device))
    if rank == 0:
        print(json.dumps(result))
    if N > 1:
        torch.distributed.barrier()
        torch.distributed.destroy_process_group()


if __name__ == "__main__":
    main()
