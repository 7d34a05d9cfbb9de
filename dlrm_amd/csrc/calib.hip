// calib.hip — in-run calibration of the box bench.py is measuring on: boxes of the pool differ by up to 6 % in what their
// matrix pipes and HBM sustain, so a roofline fraction against a constant says nothing about whether a slow line is a slow box or a
// regression.  Two probes, ~50 ms each, timed by the caller with HIP events on `stream`:
//   dlrm_calib_mfma       back-to-back v_mfma_f32_32x32x2_f32 (kind 0) or v_mfma_f32_32x32x16_bf16 (kind 1) on every SIMD, no memory
//                         traffic -> the matrix rate this chip sustains at its power budget (and the implied clock);
//   dlrm_calib_hbm_copy   float4 copy src -> dst, one float4 per thread (the access pattern MI355X_MICROARCH.md quotes 6.29 TB/s for);
//   dlrm_calib_hbm_gather 512-byte rows at pseudo-random places of a multi-GiB buffer, one row per half-wave, eight in flight — the access
//                         pattern of the embedding kernels.  Added after a visit whose MFMA and copy probes read 156.6 TFLOP/s / 6.30 TB/s (the
//                         fastest seen) while its embedding kernels ran 17-22 % and its GEMMs 3 % slower than on other boxes
//                         (profiles/round4/box_classes.md): streaming probes do not see whatever separates those boxes.
// Not on the training path; nothing here is called by the model.
#include "common.h"

namespace {
typedef float floatx16 __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

// RANDOM operands (kinds 2 / 3): eight different pseudo-random operand register sets per lane, rotated through the unrolled loop, so that
// consecutive MFMAs multiply different bit patterns the way a GEMM's operand stream does.  Constant operands barely toggle the multiplier
// array: a GPU that holds 2.4 GHz on kinds 0 / 1 may not hold it here — round 4 met a GPU whose constant-operand probes read the pool's best
// (156.6 / 2472 TFLOP/s) while its bf16 GEMM step ran 26 % and its fp32 step 5 % slower than on other GPUs (profiles/round4/box_classes.md).
__device__ __forceinline__ unsigned calib_mix(unsigned x) { x ^= x >> 16; x *= 0x7feb352du; x ^= x >> 15; x *= 0x846ca68bu; x ^= x >> 16; return x; }
template <int KIND>
__global__ __launch_bounds__(256) void calib_mfma_random_kernel(float* out, int iters, unsigned seed) {
    floatx16 acc[8];
#pragma unroll
    for (int i = 0; i < 8; ++i)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;
    const unsigned tid = blockIdx.x * 256 + threadIdx.x;
    float af[8], bf[8];
    bf16x8 av[8], bv[8];
#pragma unroll
    for (int k = 0; k < 8; ++k) {
        // values in (-1, 1), both signs: the accumulators do a bounded random walk
        af[k] = (float)(int)(calib_mix(seed + tid * 16 + k) >> 8) * (1.f / 8388608.f) - 1.f;
        bf[k] = (float)(int)(calib_mix(seed + tid * 16 + 8 + k) >> 8) * (1.f / 8388608.f) - 1.f;
#pragma unroll
        for (int r = 0; r < 8; ++r) {
            av[k][r] = (__bf16)((float)(int)(calib_mix(seed + (tid * 16 + k) * 8 + r) >> 8) * (1.f / 8388608.f) - 1.f);
            bv[k][r] = (__bf16)((float)(int)(calib_mix(~seed + (tid * 16 + k) * 8 + r) >> 8) * (1.f / 8388608.f) - 1.f);
        }
    }
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int k = 0; k < 8; ++k)
#pragma unroll
            for (int i = 0; i < 8; ++i) {
                if (KIND == 0) acc[i] = __builtin_amdgcn_mfma_f32_32x32x2f32(af[k], bf[(k + i) & 7], acc[i], 0, 0, 0);
                else           acc[i] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(av[k], bv[(k + i) & 7], acc[i], 0, 0, 0);
            }
    }
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < 8; ++i)
#pragma unroll
        for (int r = 0; r < 16; ++r) s += acc[i][r];
    if (s == 123.456f) out[0] = s;          // (practically) never true: keeps the accumulators live
}

template <int KIND>
__global__ __launch_bounds__(256) void calib_mfma_kernel(float* out, int iters, float a, float b) {
    floatx16 acc[8];
#pragma unroll
    for (int i = 0; i < 8; ++i)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;
    bf16x8 av, bv;
#pragma unroll
    for (int r = 0; r < 8; ++r) { av[r] = (__bf16)a; bv[r] = (__bf16)b; }
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int k = 0; k < 8; ++k)
#pragma unroll
            for (int i = 0; i < 8; ++i) {
                if (KIND == 0) acc[i] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc[i], 0, 0, 0);
                else           acc[i] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(av, bv, acc[i], 0, 0, 0);
            }
    }
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < 8; ++i)
#pragma unroll
        for (int r = 0; r < 16; ++r) s += acc[i][r];
    if (s == 123.456f) out[0] = s;          // never true: keeps the accumulators live
}

// ONE float4 per thread, no loop: the fastest of the copy shapes swept on the box (tools/probes/hbm_copy_sweep.hip, profiles/round4: 6.18 TB/s
// vs 4.2-5.7 TB/s for grid-stride loops of any unroll / grid, 5.17 TB/s for hipMemcpyAsync) — the "float4 copy" MI355X_MICROARCH.md quotes
__global__ __launch_bounds__(256) void calib_copy_kernel(const float4* __restrict__ a, float4* __restrict__ b, size_t n) {
    const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (i < n) b[i] = a[i];
}
// one 512-byte row per half-wave and step, 8 independent rows in flight per half-wave; rows by a 64-bit mix of (half-wave, step)
__global__ __launch_bounds__(256) void calib_gather_kernel(const float4* __restrict__ t, unsigned long long nrows, float* out, unsigned seed) {
    const unsigned long long hw = ((unsigned long long)blockIdx.x * 256 + threadIdx.x) >> 5;
    const int l = threadIdx.x & 31;
    float4 s = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
    for (int r = 0; r < 8; ++r) {
        unsigned long long x = (hw * 8 + r) * 0x9E3779B97F4A7C15ull + seed;
        x ^= x >> 29; x *= 0xBF58476D1CE4E5B9ull; x ^= x >> 32;
        const float4 v = t[(x % nrows) * 32 + l];
        s.x += v.x; s.y += v.y; s.z += v.z; s.w += v.w;
    }
    if (s.x + s.y + s.z + s.w == 123.456f) out[0] = s.x;      // never true for a zero-filled buffer: keeps the loads live
}
}  // namespace

// reads `rows_to_read` (rounded down to a multiple of 64) pseudo-random 512-byte rows of table[0 .. table_bytes); *bytes_out = bytes read
extern "C" int dlrm_calib_hbm_gather(const void* table, int64_t table_bytes, int64_t rows_to_read, uint32_t seed, float* scratch, double* bytes_out,
                                     void* stream) {
    if (!table || !scratch || !bytes_out || table_bytes < 512 || rows_to_read < 64) return DLRM_E_ARG;
    if (!dlrm_aligned16(table)) return DLRM_E_ALIGN;
    const unsigned long long nrows = (unsigned long long)(table_bytes / 512);
    const long long blocks = rows_to_read / 64;                // 8 half-waves x 8 rows per block
    if (blocks > 0x7fffffffll) return DLRM_E_RANGE;
    hipLaunchKernelGGL(calib_gather_kernel, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, (const float4*)table, nrows, scratch, seed);
    DLRM_LAUNCH_CHECK();
    *bytes_out = (double)blocks * 64.0 * 512.0;
    return 0;
}

extern "C" int dlrm_calib_mfma(int kind, int iters, float* scratch, double* flop_out, void* stream) {
    if (iters <= 0 || !scratch || !flop_out || kind < 0 || kind > 3) return DLRM_E_ARG;     // 0 fp32 / 1 bf16 on constant operands, 2 / 3 the same on random ones
    hipDeviceProp_t p;
    hipError_t e = hipGetDeviceProperties(&p, dlrm_current_device());
    if (e != hipSuccess) return (int)e;
    const int grid = p.multiProcessorCount * 2;                 // 2 waves per SIMD
    if (kind == 0)      hipLaunchKernelGGL(calib_mfma_kernel<0>, dim3(grid), dim3(256), 0, (hipStream_t)stream, scratch, iters, 1.0f, 2.0f);
    else if (kind == 1) hipLaunchKernelGGL(calib_mfma_kernel<1>, dim3(grid), dim3(256), 0, (hipStream_t)stream, scratch, iters, 1.0f, 2.0f);
    else if (kind == 2) hipLaunchKernelGGL(calib_mfma_random_kernel<0>, dim3(grid), dim3(256), 0, (hipStream_t)stream, scratch, iters, 0x9e3779b9u);
    else                hipLaunchKernelGGL(calib_mfma_random_kernel<1>, dim3(grid), dim3(256), 0, (hipStream_t)stream, scratch, iters, 0x9e3779b9u);
    DLRM_LAUNCH_CHECK();
    // per wave and iteration: 64 MFMAs of 32 x 32 x {2, 16} multiply-adds
    *flop_out = (double)grid * 4.0 * iters * 64.0 * (2.0 * 32 * 32 * ((kind & 1) == 0 ? 2 : 16));
    return 0;
}

extern "C" int dlrm_calib_hbm_copy(const void* src, void* dst, int64_t bytes, void* stream) {
    if (!src || !dst || bytes < 16 || (bytes & 15)) return DLRM_E_ARG;
    if (!dlrm_aligned16(src) || !dlrm_aligned16(dst)) return DLRM_E_ALIGN;
    const size_t n = (size_t)(bytes / 16);
    if ((n + 255) / 256 > 0x7fffffffull) return DLRM_E_RANGE;
    hipLaunchKernelGGL(calib_copy_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, (hipStream_t)stream, (const float4*)src, (float4*)dst, n);
    DLRM_LAUNCH_CHECK();
    return 0;
}

// ---- CU-partitioned streams (does an HBM-bound kernel on a few CUs compose with an MFMA-bound one on the rest?) ------------
// dlrm_stream_create_cu_range: a HIP stream whose kernels may only run on CUs [first, first + count) (hipExtStreamCreateWithCUMask; CU i = bit i
// of the mask, 32 bits per word).  The caller owns the stream (dlrm_stream_destroy).  Used by tools/probes/cu_mask_probe.py only.
extern "C" int dlrm_stream_create_cu_range(int first, int count, void** stream_out) {
    if (!stream_out || first < 0 || count <= 0) return DLRM_E_ARG;
    hipDeviceProp_t p;
    hipError_t e = hipGetDeviceProperties(&p, dlrm_current_device());
    if (e != hipSuccess) return (int)e;
    const int cus = p.multiProcessorCount;
    if (first + count > cus) return DLRM_E_RANGE;
    uint32_t mask[16] = {};
    for (int i = first; i < first + count; ++i) mask[i >> 5] |= 1u << (i & 31);
    hipStream_t st = nullptr;
    e = hipExtStreamCreateWithCUMask(&st, (uint32_t)((cus + 31) / 32), mask);
    if (e != hipSuccess) return (int)e;
    *stream_out = (void*)st;
    return 0;
}
extern "C" int dlrm_stream_destroy(void* stream) {
    if (!stream) return DLRM_E_ARG;
    hipError_t e = hipStreamDestroy((hipStream_t)stream);
    return e == hipSuccess ? 0 : (int)e;
}
