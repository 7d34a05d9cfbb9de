// common.h — shared device/host helpers for libdlrm_hip.so (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include "../../include/dlrm_hip.h"

#define DLRM_WAVE 64

// kernel launch wrapper: returns the hipError_t of the launch as a positive int
#define DLRM_LAUNCH_CHECK()                      \
    do {                                         \
        hipError_t e__ = hipGetLastError();      \
        if (e__ != hipSuccess) return (int)e__;  \
    } while (0)

#define DLRM_REQUIRE(cond, code, msg)                                        \
    do {                                                                     \
        if (!(cond)) {                                                       \
            fprintf(stderr, "libdlrm_hip: %s: %s\n", __func__, msg);         \
            return (code);                                                   \
        }                                                                    \
    } while (0)

#define DLRM_MAX_DEVICES 64
static inline int dlrm_current_device() { int d = 0; (void)hipGetDevice(&d); return (d >= 0 && d < DLRM_MAX_DEVICES) ? d : 0; }

static inline bool dlrm_aligned16(const void* p) { return (((uintptr_t)p) & 15u) == 0; }

// timing-only tuning switches (DLRM_GEMM_DEBUG, DLRM_INTERACT_DEBUG, DLRM_SEG_DEBUG) make kernels skip work: results are WRONG.
// They — and since round 6 EVERY environment variable the library ever read — exist ONLY in a tuning build (`make TUNING=1` -> -DDLRM_TUNING, tools/probes/): the product library never reads these
// environment variables — the function folds to the constant 0, so no environment can turn a kernel of the shipped library into
// a no-op (tests/test_host_logic.py checks that the names are absent from the binary).
#ifdef DLRM_TUNING
static inline int dlrm_debug_env(const char* name, int mask) {
    const char* e = getenv(name);
    const int v = e ? (atoi(e) & mask) : 0;
    if (v) fprintf(stderr, "libdlrm_hip: %s=%d is a tuning switch: its timing-only bits make kernels skip work and their results are WRONG\n", name, v);
    return v;
}
#define DLRM_DEBUG_ENV(name, mask) dlrm_debug_env(name, mask)
// launch-plan / schedule knobs of the A/B visits (tools/gpu_visit.sh abx with DLRM_HIP_LIB = the tuning build): read once per process
static inline int dlrm_tune_env(const char* name, int dflt) { const char* e = getenv(name); return e ? atoi(e) : dflt; }
static inline const char* dlrm_tune_env_str(const char* name) { return getenv(name); }
#define DLRM_TUNE_ENV(name, dflt) dlrm_tune_env(name, dflt)
#define DLRM_TUNE_ENV_STR(name) dlrm_tune_env_str(name)
#else
#define DLRM_DEBUG_ENV(name, mask) 0
// the product library has NO environment switches: every knob folds to its default at compile time and the variable's name is not in the
// binary (tests/test_host_logic.py); behaviour travels with the call's arguments only
#define DLRM_TUNE_ENV(name, dflt) (dflt)
#define DLRM_TUNE_ENV_STR(name) ((const char*)nullptr)
#endif

// A step size (learning rate) that travels BY VALUE in the kernarg or is READ FROM DEVICE MEMORY when the kernel runs (dev != nullptr:
// value = sign * *dev).  The second form is what a captured HIP graph needs: the reference's LRPolicyScheduler changes lr every
// iteration during warm-up and decay (dlrm_s_pytorch.py:169-203, stepped at :1621); a graph whose update kernels read lr from a device
// scalar is replayed unchanged, the new value written in front of the replay (dlrm_graph_replay) — no re-capture.  sign = -1 gives the
// exact negation the kernels multiply with (fma(-lr, g, w)), bit-identical to passing -lr by value.
struct DlrmStep {
    float v; const float* dev; float sign;
    __device__ __forceinline__ operator float() const { return dev ? sign * *dev : v; }
};
static inline DlrmStep dlrm_step_neg(float lr, const float* lr_dev) { DlrmStep s; s.v = -lr; s.dev = lr_dev; s.sign = -1.f; return s; }
static inline DlrmStep dlrm_step_pos(float lr, const float* lr_dev) { DlrmStep s; s.v = lr; s.dev = lr_dev; s.sign = 1.f; return s; }

__device__ __forceinline__ float dlrm_wave_sum(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}

// Zero fill as a KERNEL, never hipMemsetAsync: a memset node inside a captured HIP graph replays unreliably on ROCm 7.2 (the bias gradient
// of a small layer came back inf on the second replay of a whole-step graph when a host synchronisation preceded the capture —
// tools/probes/graph_prove_probe2.py; rocPRIM's memsets are why its sort cannot be replayed either).  rows x cols floats at pitch ld.
namespace {
__global__ __launch_bounds__(256) void dlrm_zero2d_kernel(float* __restrict__ p, long long ld, long long cols, long long total) {
    for (long long e = (long long)blockIdx.x * 256 + threadIdx.x; e < total; e += (long long)gridDim.x * 256) {
        const long long r = e / cols;
        p[r * ld + (e - r * cols)] = 0.f;
    }
}
}  // namespace
static inline int dlrm_zero2d(float* p, long long ld, long long cols, long long rows, hipStream_t st) {
    const long long total = rows * cols;
    if (total <= 0) return 0;
    long long nb = (total + 255) / 256; if (nb > 4096) nb = 4096;
    hipLaunchKernelGGL(dlrm_zero2d_kernel, dim3((unsigned)nb), dim3(256), 0, st, p, ld, cols, total);
    hipError_t e = hipGetLastError();
    return e == hipSuccess ? 0 : (int)e;
}

// A launch PREDICATE read on the device (ABI 16): a kernel given one runs only if (*flag != 0) == (nonzero != 0), else every workgroup
// returns at once.  It lets a caller enqueue BOTH implementations of a step whose choice depends on a device-side fact — the fused lookup +
// interaction kernels if the batch has one lookup per bag (dlrm_offsets_iota_flags counted no violation), dlrm_emb_fwd + dlrm_interact_*
// otherwise — without the host waiting for that fact (the host wait ended the host's run-ahead once per step: profiles/round6/proof_wait.md).
struct DlrmPred {
    const int* flag; int nonzero;
    __device__ __forceinline__ bool skip() const { return flag != nullptr && ((*flag != 0) != (nonzero != 0)); }
};

// Tables are passed to the embedding kernels BY VALUE in the kernarg segment (no H2D copy of a
// pointer table, HIP-graph friendly).  32 tables x 6 x 8 B = 1.5 KiB.
#define DLRM_MAX_TABLES_PER_LAUNCH 32
struct EmbArgs {
    float*      w[DLRM_MAX_TABLES_PER_LAUNCH];
    const void* idx[DLRM_MAX_TABLES_PER_LAUNCH];
    const void* off[DLRM_MAX_TABLES_PER_LAUNCH];
    const float* psw[DLRM_MAX_TABLES_PER_LAUNCH];
    long long   nnz[DLRM_MAX_TABLES_PER_LAUNCH];
    long long   rows[DLRM_MAX_TABLES_PER_LAUNCH];
    int         slot[DLRM_MAX_TABLES_PER_LAUNCH];  // feature slot (column block) of the table in out/dout
    long long*  err;                               // device-visible int64[4] out-of-range report (nullable), see dlrm_hip.h
    DlrmPred    pred;                              // launch predicate (dlrm_emb_fwd_pred); flag == nullptr: always run
};

// Out-of-range index: the lookup is SKIPPED (never read or written out of bounds) and, when the caller passed an error
// block, reported as {1, table, index, rows}.  Plain stores: concurrent reporters race benignly (any one bad index wins).
__device__ __forceinline__ void dlrm_report_bad_index(long long* err, int table, long long idx, long long rows) {
    if (err) { volatile long long* e = err; e[1] = table; e[2] = idx; e[3] = rows; e[0] = 1; }
}
__device__ __forceinline__ bool dlrm_index_ok(long long idx, long long rows) {
    return (unsigned long long)idx < (unsigned long long)rows;
}
#define DLRM_DEAD_BAG 0xFFFFFFFFu   // bag_of[] marker of a skipped (out-of-range) lookup in the sorted update paths

// emb_sorted.hip: sort-based fused backward + SGD (mode DLRM_UPD_SORTED of dlrm_emb_bwd_sgd)
int dlrm_emb_bwd_sgd_sorted_impl(int T, int64_t B, int D, void* const* weight_host, const int64_t* rows_host,
                                 const void* const* indices_host, const void* const* offsets_host,
                                 const int64_t* nnz_host, const void* const* psw_host, int idx_bits,
                                 const float* dout, int64_t dout_ld, float lr, const float* lr_dev, void* workspace,
                                 int64_t workspace_bytes, int64_t* err, void* stream);

// gemv.hip: the N == 1 MLP layer as HBM-streaming kernels; each returns 0 when it handled the call and
// DLRM_GEMV_NOT_HANDLED when the shape/alignment is outside its fast path (the caller then uses the GEMM kernels).
#define DLRM_GEMV_NOT_HANDLED (-100)
int64_t dlrm_gemv_bwd_weight_workspace_bytes(int64_t M, int K);
int dlrm_gemv_fwd(int64_t M, int K, const float* X, int64_t ldx, const float* w, const float* bias, int act, float* Y,
                  int64_t ldy, hipStream_t st);
int dlrm_gemv_bwd_data(int64_t M, int K, const float* dY, int64_t lddy, const float* w, const float* Xact, int64_t ldxa,
                       int kind, float* dX, int64_t lddx, hipStream_t st);
int dlrm_gemv_bwd_weight(int64_t M, int K, const float* dY, int64_t lddy, const float* X, int64_t ldx, float* dW,
                         float* dbias, int accumulate, void* workspace, int64_t workspace_bytes, hipStream_t st);

int dlrm_gemv_bwd_fused(int64_t M, int K, const float* dY, int64_t lddy, const float* Yout, int64_t ldy, int act_out, const float* X,
                        int64_t ldx, const float* w, int xact_kind, float* dX, int64_t lddx, float* dW, float* dbias, int accumulate,
                        void* workspace, int64_t workspace_bytes, hipStream_t st);

// smallk.hip: weight gradient of layers with K <= 16 (the first bottom-MLP layer: 13 dense features padded to 16)
int64_t dlrm_smallk_bwd_weight_workspace_bytes(int64_t M, int N, int K);
int dlrm_smallk_bwd_weight(int64_t M, int N, int K, int K_store, const float* dY, int64_t lddy, const float* X, int64_t ldx, float* dW,
                           int64_t lddw, float* dbias, int accumulate, void* workspace, int64_t workspace_bytes,
                           hipStream_t st);

// gemm_bf16.hip: the bf16-shaped (256 x 256 x 64, four phases per k-tile) GEMM; 0 = handled, DLRM_GEMV_NOT_HANDLED = outside its preconditions
int dlrm_gemm_bf16_phased(int64_t M, int N, int K, const uint16_t* A, int64_t lda, const uint16_t* B, int64_t ldb, const float* bias, int act,
                          uint64_t* relu_bits_out, const uint64_t* relu_bits_in, const float* addend, int64_t ldadd, const float* addend2,
                          int64_t ldadd2, float* C, int64_t ldc, uint16_t* Cb, int64_t ldcb, hipStream_t st, const float* mul = nullptr,
                          int64_t ldmul = 0, uint16_t* Ub = nullptr, int64_t ldub = 0);
bool dlrm_gemm_bf16_wgrad_ok(int64_t Mb, int N_out, int K_in, int64_t lddz, int64_t ldx);
void dlrm_gemm_bf16_wgrad_plan(int64_t Mb, int N_out, int K_in, int* splits_out, int64_t* kchunk_out);
int dlrm_gemm_bf16_wgrad_phased(int64_t Mb, int N_out, int K_in, const uint16_t* dZ, int64_t lddz, const uint16_t* X, int64_t ldx,
                                float* slabs, int64_t ldc, int64_t slab_stride, float* rowsum_parts, int splits, int64_t kchunk, hipStream_t st,
                                int planes, int64_t planeZ, int64_t planeX);
bool dlrm_gemm_bf16x6_ok(int64_t M, int N, int K, int64_t lda, int64_t ldb);
int dlrm_gemm_bf16x6_phased(int64_t M, int N, int K, const uint16_t* A, int64_t lda, int64_t planeA, const uint16_t* B, int64_t ldb, int64_t planeB,
                            const float* bias, int act, uint64_t* relu_bits_out, const uint64_t* relu_bits_in, float* C, int64_t ldc, uint16_t* Cp,
                            int64_t ldcp, int64_t planeC, hipStream_t st);
