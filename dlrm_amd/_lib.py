"""ctypes binding of libdlrm_hip.so (C ABI declared in include/dlrm_hip.h).

The library is built in-tree (dlrm_amd/csrc/Makefile -> dlrm_amd/libdlrm_hip.so) so that it travels
with the repository snapshot to a GPU box.  There is NO fallback: if the shared object is missing
and cannot be built, or a call returns non-zero, a RuntimeError is raised.
"""
from __future__ import annotations

import ctypes as C
import fcntl
import os
import subprocess
import threading

_HERE = os.path.dirname(os.path.abspath(__file__))
# DLRM_HIP_LIB: load another build of the same library (tools/probes: the TUNING build with its timing-only switches); never rebuilt
LIB_PATH = os.environ.get("DLRM_HIP_LIB") or os.path.join(_HERE, "libdlrm_hip.so")
CSRC = os.path.join(_HERE, "csrc")

ACT_NONE, ACT_RELU, ACT_SIGMOID = 0, 1, 2
UPD_ATOMIC, UPD_DETERMINISTIC, UPD_SORTED = 0, 1, 2
ARITH_F32, ARITH_BF16X6, ARITH_BF16 = 0, 1, 2
EXPECTED_ABI = 17          # dlrm_hip_abi_version() of the library these bindings (SIGNATURES) were written against

_lock = threading.Lock()
_lib = None

_vp, _i64, _i32, _f32 = C.c_void_p, C.c_int64, C.c_int, C.c_float
_pp = C.POINTER(C.c_void_p)
_pi64 = C.POINTER(C.c_int64)

# name -> (restype, argtypes); must list every symbol of include/dlrm_hip.h
SIGNATURES = {
    "dlrm_hip_abi_version": (_i32, []),
    "dlrm_hip_build_info": (C.c_char_p, []),
    "dlrm_hip_device_info": (_i32, [_i32, C.POINTER(_i32), C.POINTER(_i32), _pi64, C.c_char_p, _i32]),
    "dlrm_calib_mfma": (_i32, [_i32, _i32, _vp, C.POINTER(C.c_double), _vp]),
    "dlrm_calib_hbm_copy": (_i32, [_vp, _vp, _i64, _vp]),
    "dlrm_calib_hbm_gather": (_i32, [_vp, _i64, _i64, C.c_uint32, _vp, _vp, _vp]),
    "dlrm_stream_create_cu_range": (_i32, [_i32, _i32, C.POINTER(C.c_void_p)]),
    "dlrm_stream_destroy": (_i32, [_vp]),
    "dlrm_emb_fwd": (_i32, [_i32, _i64, _i32, _pp, _pi64, _pp, _pp, _pi64, _pp, _i32, _vp, _i64, _vp, _vp]),
    "dlrm_emb_bwd_workspace_bytes": (_i64, [_i32, _pi64, _pi64]),
    "dlrm_emb_sort_kind": (_i32, [_i32, _pi64, _pi64]),
    "dlrm_cast_bf16": (_i32, [_i64, _i32, _i32, _vp, _i64, _vp, _i64, _vp]),
    "dlrm_cast_bf16_transposed": (_i32, [_i32, _i32, _i32, _vp, _i64, _vp, _i64, _vp]),
    "dlrm_cast_bf16_multi": (_i32, [_i32, _pp, _pi64, C.POINTER(_i32), C.POINTER(_i32), _pp, _pi64, C.POINTER(_i32), _pp, _pi64, C.POINTER(_i32), _vp]),
    "dlrm_gemm_bf16": (_i32, [_i64, _i32, _i32, _vp, _i64, _vp, _i64, _vp, _i32, _vp, _vp, _vp, _i64, _vp, _i64, _vp, _i64, _vp, _i64, _vp]),
    "dlrm_split_bf16x3": (_i32, [_i64, _i32, _i32, _vp, _i64, _vp, _i64, _i64, _vp]),
    "dlrm_split_bf16x3_transposed": (_i32, [_i32, _i32, _i32, _vp, _i64, _vp, _i64, _i64, _vp]),
    "dlrm_gemm_bf16x6_supported": (_i32, [_i64, _i32, _i32, _i64, _i64]),
    "dlrm_gemm_bf16x6": (_i32, [_i64, _i32, _i32, _vp, _i64, _i64, _vp, _i64, _i64, _vp, _i32, _vp, _vp, _vp, _i64, _vp, _i64, _i64, _vp]),
    "dlrm_linear_bwd_weight_bf16x6": (_i32, [_i64, _i32, _i32, _i32, _vp, _i64, _i64, _vp, _i64, _i64, _vp, _i64, _vp, _i32, _vp, _i64, _vp]),
    "dlrm_linear_bwd_weight_bf16_workspace_bytes": (_i64, [_i64, _i32, _i32]),
    "dlrm_linear_bwd_weight_bf16": (_i32, [_i64, _i32, _i32, _i32, _vp, _i64, _vp, _i64, _vp, _i64, _vp, _i32, _vp, _i64, _vp]),
    "dlrm_emb_sort_lookups": (_i32, [_i32, _i64, _pi64, _pp, _pp, _pi64, _i32, _vp, _i64, _vp, _vp, _vp, C.POINTER(_i32), _vp, _vp]),
    "dlrm_emb_bwd_sgd": (_i32, [_i32, _i64, _i32, _pp, _pi64, _pp, _pp, _pi64, _pp, _i32, _vp, _i64,
                                _f32, _vp, _i32, _vp, _i64, _vp, _vp]),
    "dlrm_pool_weights_gather": (_i32, [_i32, _pi64, _pp, _pi64, _i32, _pp, _pp, _vp, _vp]),
    "dlrm_emb_psw_grad": (_i32, [_i32, _i64, _i32, _pp, _pi64, _pp, _pp, _pi64, _i32, _vp, _i64, _pp, _vp]),
    "dlrm_emb_bwd_coo": (_i32, [_i32, _i64, _i32, _pp, _pi64, _pp, _i32, _vp, _i64, _pp, _vp]),
    "dlrm_emb_adagrad_workspace_bytes": (_i64, [_i32, _i32, _pi64, _pi64]),
    "dlrm_emb_bwd_rowwise_adagrad": (_i32, [_i32, _i64, _i32, _pp, _pp, _pi64, _pp, _pp, _pi64, _pp, _i32,
                                            _vp, _i64, _f32, _vp, _f32, _vp, _i64, _vp, _vp]),
    "dlrm_interact_fwd": (_i32, [_i64, _i32, _i32, _pp, _pi64, _i32, _vp, _i64, _vp]),
    "dlrm_interact_bwd": (_i32, [_i64, _i32, _i32, _pp, _pi64, _i32, _vp, _i64, _pp, _pi64, _vp]),
    "dlrm_interact_gather_ok": (_i32, [_i32, _i32]),
    "dlrm_offsets_are_iota": (_i32, [_i32, _i64, _pp, _i32, _vp, _vp]),
    "dlrm_interact_fwd_gather": (_i32, [_i64, _i32, _i32, _pp, _pi64, _pp, _pp, _pi64, _i32, _i32, _vp, _i64, _vp, _vp]),
    "dlrm_interact_bwd_gather": (_i32, [_i64, _i32, _i32, _pp, _pi64, _pp, _pp, _pi64, _i32, _i32, _vp, _i64, _pp, _pi64, _vp, _vp]),
    "dlrm_offsets_iota_flags": (_i32, [_i32, _i64, _pp, _i32, _vp, _vp, _vp]),
    "dlrm_emb_fwd_pred": (_i32, [_i32, _i64, _i32, _pp, _pi64, _pp, _pp, _pi64, _pp, _i32, _vp, _i64, _vp, _vp, _i32, _vp]),
    "dlrm_interact_fwd_pred": (_i32, [_i64, _i32, _i32, _pp, _pi64, _pp, _pp, _pi64, _i32, _i32, _vp, _i64, _vp, _vp, _i32, _vp]),
    "dlrm_interact_bwd_pred": (_i32, [_i64, _i32, _i32, _pp, _pi64, _pp, _pp, _pi64, _i32, _i32, _vp, _i64, _pp, _pi64, _vp, _vp, _i32, _vp]),
    "dlrm_emb_presort": (_i32, [_i32, _i64, _pi64, _pp, _pp, _pi64, _i32, _vp, _i64, _vp, _vp, _vp]),
    "dlrm_interact_bwd_gather_sgd": (_i32, [_i64, _i32, _i32, _pp, _pi64, _pp, _pp, _pi64, _i32, _i32, _vp, _i64, _pp, _pi64, _vp, _f32, _vp, _vp,
                                            _vp, _i32, _vp]),
    "dlrm_emb_bwd_sgd_presorted": (_i32, [_i32, _i64, _i32, _pp, _pi64, _pi64, _vp, _i64, _f32, _vp, _vp, _i64, _i32, _vp, _i32, _vp]),
    "dlrm_relu_bits_bytes": (_i64, [_i64, _i32]),
    "dlrm_linear_fwd": (_i32, [_i64, _i32, _i32, _vp, _i64, _vp, _i64, _vp, _i32, _vp, _i64, _vp, _i32, _vp]),
    "dlrm_linear_bwd_data": (_i32, [_i64, _i32, _i32, _vp, _i64, _vp, _i64, _vp, _i64, _i32, _vp, _vp, _i64, _i32, _vp]),
    "dlrm_linear_bwd_weight_workspace_bytes": (_i64, [_i64, _i32, _i32]),
    "dlrm_linear_bwd_weight": (_i32, [_i64, _i32, _i32, _vp, _i64, _vp, _i64, _vp, _i64, _vp, _i32, _vp, _i64, _i32, _vp]),
    "dlrm_linear_head_bwd": (_i32, [_i64, _i32, _vp, _i64, _vp, _i64, _i32, _vp, _i64, _vp, _i32, _vp, _i64, _vp, _vp, _i32, _vp, _i64, _vp]),
    "dlrm_linear_bwd_weight_padded": (_i32, [_i64, _i32, _i32, _i32, _vp, _i64, _vp, _i64, _vp, _i64, _vp, _i32, _vp, _i64, _i32, _vp]),
    "dlrm_pad_cols": (_i32, [_i64, _i32, _i32, _vp, _i64, _vp, _i64, _vp]),
    "dlrm_act_bwd": (_i32, [_i64, _i32, _vp, _i64, _vp, _i64, _i32, _vp, _i64, _vp, _vp]),
    "dlrm_loss_workspace_bytes": (_i64, [_i64]),
    "dlrm_bce_loss": (_i32, [_i64, _vp, _vp, _vp, _f32, _f32, _f32, _vp, _vp, _vp, _vp]),
    "dlrm_bce_logits_loss": (_i32, [_i64, _vp, _vp, _f32, _vp, _vp, _vp, _vp]),
    "dlrm_mse_loss": (_i32, [_i64, _vp, _vp, _f32, _vp, _vp, _vp, _vp]),
    "dlrm_scale_by_device_scalar": (_i32, [_i64, _vp, _vp, _vp, _vp]),
    "dlrm_sgd_dense": (_i32, [_i64, _vp, _vp, _f32, _vp, _vp]),
    "dlrm_sgd_dense_multi": (_i32, [_i32, _pp, _pp, _pi64, _f32, _vp, _vp]),
    "dlrm_adagrad_dense": (_i32, [_i64, _vp, _vp, _vp, _f32, _vp, _f32, _vp]),
    "dlrm_set_f32": (_i32, [_i32, _pp, C.POINTER(_f32), _vp]),
    "dlrm_binary_metrics_workspace_bytes": (_i64, [_i64]),
    "dlrm_binary_metrics": (_i32, [_i64, _vp, _vp, _vp, _vp, _i64, _vp]),
    "dlrm_gen_workspace_bytes": (_i64, [_i32, _i64]),
    "dlrm_gen_uniform_bags": (_i32, [_i32, _i64, _pi64, _i32, _i32, C.c_uint64, _i32, _pp, _pp, _vp, _vp, _i64, _vp]),
    "dlrm_gen_uniform_dense": (_i32, [_i64, _vp, _i32, C.c_uint64, _vp]),
    "dlrm_criteo_bin_transform": (_i32, [_i64, _vp, _i64, _i32, _vp, _i64, _vp, _vp, _i64, _vp, _vp]),
    "dlrm_multihot_gen_table": (_i32, [_i32, _i64, _i32, _i32, C.c_uint64, _vp, _vp]),
    "dlrm_multihot_expand": (_i32, [_i32, _i64, _vp, _i32, _pp, _pi64, C.POINTER(_i32), _vp, _vp, _vp, _vp, _vp]),
    "dlrm_copy_blocks": (_i32, [_i64, _i32, _pp, _pi64, _pp, _pi64, C.POINTER(_i32), _vp]),
    "dlrm_bce_elementwise": (_i32, [_i64, _vp, _vp, _vp, _vp]),
    "dlrm_bce_elementwise_bwd": (_i32, [_i64, _vp, _vp, _vp, _vp, _vp]),
    "dlrm_cross_fwd": (_i32, [_i64, _vp, _vp, _vp, _vp, _vp, _vp]),
    "dlrm_cross_bwd": (_i32, [_i64, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _i32, _vp]),
    "dlrm_gemm_bf16_cross": (_i32, [_i64, _i32, _i32, _vp, _i64, _vp, _i64, _vp, _vp, _i64, _vp, _i64, _vp, _i64, _vp, _i64, _vp, _i64, _vp]),
    "dlrm_add": (_i32, [_i64, _vp, _vp, _vp, _vp]),
    "dlrm_graph_replay": (_i32, [_i32, _pp, _pp, _pi64, _i32, _pp, C.POINTER(_f32), _vp, _i32, _vp]),
    "dlrm_tower_fwd": (_i32, [_i64, _i32, C.POINTER(_i32), C.POINTER(_i32), _vp, _i64, _pp, _pi64, _pp, _pp, _pi64, _vp]),
    "dlrm_tower_bwd": (_i32, [_i64, _i32, C.POINTER(_i32), C.POINTER(_i32), _vp, _i64, _i32, _pp, _pi64, _pp, _pi64, _pp, _pi64, _vp, _i64, _vp]),
    "dlrm_tower_wgrad_workspace_bytes": (_i64, [_i64, _i32, C.POINTER(_i32)]),
    "dlrm_tower_wgrad": (_i32, [_i64, _i32, C.POINTER(_i32), C.POINTER(_i32), _pp, _pi64, _pp, _pi64, _pp, _pi64, _pp, _vp, _i64, _vp]),
    "dlrm_clamp": (_i32, [_i64, _vp, _f32, _f32, _vp, _vp]),
    "dlrm_clamp_bwd": (_i32, [_i64, _vp, _f32, _f32, _vp, _vp, _vp]),
}

_ERR = {-1: "DLRM_E_ARG (null pointer / bad size)", -2: "DLRM_E_ALIGN", -3: "DLRM_E_RANGE (compiled limit exceeded)",
        -4: "DLRM_E_MODE (unknown mode / not implemented)"}


def build(force: bool = False, verbose: bool = False) -> str:
    """Compile libdlrm_hip.so for gfx950 with hipcc (cross-compiles without a GPU).

    Safe to call from several processes at once (`torchrun --nproc-per-node 8` on a fresh checkout): the build is
    serialised by an exclusive flock on csrc/.build.lock, and the Makefile links to a temporary name that is renamed
    over the target, so no process can ever dlopen a half-written library."""
    with open(os.path.join(CSRC, ".build.lock"), "w") as lk:
        fcntl.flock(lk, fcntl.LOCK_EX)
        try:
            if force:
                subprocess.run(["make", "-C", CSRC, "clean"], check=False, capture_output=not verbose)
            r = subprocess.run(["make", "-C", CSRC, "-j", str(os.cpu_count() or 4)], capture_output=True, text=True)
            if r.returncode != 0:
                raise RuntimeError("building libdlrm_hip.so failed:\n" + r.stdout[-4000:] + r.stderr[-4000:])
            if verbose:
                print(r.stdout[-2000:])
        finally:
            fcntl.flock(lk, fcntl.LOCK_UN)
    return LIB_PATH


def _sources_newer_than_lib() -> bool:
    if not os.path.exists(LIB_PATH):
        return True
    t = os.path.getmtime(LIB_PATH)
    files = [os.path.join(CSRC, f) for f in os.listdir(CSRC) if f.endswith((".hip", ".h")) or f == "Makefile"]
    files.append(os.path.join(_HERE, "..", "include", "dlrm_hip.h"))
    return any(os.path.exists(f) and os.path.getmtime(f) > t for f in files)


def load():
    """Return the loaded library (building it first when sources are newer and hipcc exists)."""
    global _lib
    if _lib is not None:
        return _lib
    with _lock:
        if _lib is not None:
            return _lib
        if not os.environ.get("DLRM_HIP_LIB") and _sources_newer_than_lib():
            if os.path.exists("/opt/rocm/bin/hipcc"):
                build()          # cross-process safe; a rank that lost the race finds everything up to date
            elif not os.path.exists(LIB_PATH):
                raise RuntimeError(
                    "libdlrm_hip.so is missing and hipcc is not available; the HIP extension is required "
                    "(there is no fallback path). Run `python -c 'import __graft_entry__ as g; g.build()'`.")
        lib = C.CDLL(LIB_PATH)
        lib.dlrm_hip_abi_version.restype = C.c_int
        got = int(lib.dlrm_hip_abi_version())
        if got != EXPECTED_ABI:
            # a stale or foreign build (DLRM_HIP_LIB, or a prebuilt .so without hipcc to refresh it) would be called with shifted arguments
            raise RuntimeError("%s reports C-ABI version %d, these bindings need %d (include/dlrm_hip.h); rebuild it: "
                               "python -c 'import __graft_entry__ as g; g.build()'" % (LIB_PATH, got, EXPECTED_ABI))
        for name, (res, args) in SIGNATURES.items():
            fn = getattr(lib, name)  # AttributeError if the .so lacks a declared symbol
            fn.restype = res
            fn.argtypes = args
        _lib = lib
    return _lib


def check(rc: int, what: str) -> None:
    if rc != 0:
        if rc < 0:
            raise RuntimeError(f"{what} failed: {_ERR.get(rc, rc)}")
        raise RuntimeError(f"{what} failed: hipError_t {rc}")


def ptr_array(ptrs):
    n = len(ptrs)
    return (C.c_void_p * n)(*[C.c_void_p(int(p)) if p else None for p in ptrs])


def i64_array(vals):
    return (C.c_int64 * len(vals))(*[int(v) for v in vals])
