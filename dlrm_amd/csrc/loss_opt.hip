// loss_opt.hip — loss (forward + gradient in one pass), dense SGD, strided block copies, clamp, introspection.
//
// Reference call sites replaced:
//   torch.nn.BCELoss(reduction="mean") / MSELoss via loss_fn_wrap   dlrm_s_pytorch.py:386-393,148-156
//   torch.optim.SGD.step on dense MLP parameters                      dlrm_s_pytorch.py:1343-1369,1620
//   All2All_Wait.forward's split + view of the receive buffer          extend_distributed.py:446-465
#include <chrono>
#include "common.h"

namespace {

constexpr int kLossBlock = 256;
constexpr int kLossPerThread = 4;

// BCE per torch: log terms clamped at -100; grad = w*(p-t)/max((1-p)*p, 1e-12)/B
__global__ __launch_bounds__(kLossBlock) void bce_kernel(long long B, const float* __restrict__ p,
                                                         const float* __restrict__ t,
                                                         const float* __restrict__ w, float w_neg, float w_pos, float gscale,
                                                         float* __restrict__ dp, float* __restrict__ partials) {
    __shared__ float red[kLossBlock / 64];
    float local = 0.f;
    const long long base = ((long long)blockIdx.x * kLossBlock + threadIdx.x) * kLossPerThread;
#pragma unroll
    for (int k = 0; k < kLossPerThread; ++k) {
        const long long i = base + k;
        if (i < B) {
            const float pi = p[i], ti = t[i];
            // class weight = loss_ws[target.long()] of the reference's wbce path (truncation: only t >= 1 selects w_pos)
            const float wi = (w ? w[i] : 1.f) * (ti >= 1.f ? w_pos : w_neg);
            const float lp = fmaxf(logf(pi), -100.f);
            const float l1p = fmaxf(log1pf(-pi), -100.f);
            local += wi * -(ti * lp + (1.f - ti) * l1p);
            if (dp) dp[i] = wi * (pi - ti) / fmaxf((1.f - pi) * pi, 1e-12f) * gscale;
        }
    }
    local = dlrm_wave_sum(local);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = local;
    __syncthreads();
    if (threadIdx.x == 0) {
        float s = 0.f;
#pragma unroll
        for (int k = 0; k < kLossBlock / 64; ++k) s += red[k];
        partials[blockIdx.x] = s;
    }
}

// BCEWithLogitsLoss(mean) per torch: loss = (1 - t) * x + m + log(exp(-m) + exp(-x - m)), m = max(-x, 0); grad = (sigmoid(x) - t) / B
__global__ __launch_bounds__(kLossBlock) void bce_logits_kernel(long long B, const float* __restrict__ x, const float* __restrict__ t,
                                                                float gscale, float* __restrict__ dx, float* __restrict__ partials) {
    __shared__ float red[kLossBlock / 64];
    float local = 0.f;
    const long long base = ((long long)blockIdx.x * kLossBlock + threadIdx.x) * kLossPerThread;
#pragma unroll
    for (int k = 0; k < kLossPerThread; ++k) {
        const long long i = base + k;
        if (i < B) {
            const float xi = x[i], ti = t[i];
            const float m = fmaxf(-xi, 0.f);
            local += (1.f - ti) * xi + m + logf(expf(-m) + expf(-xi - m));
            if (dx) dx[i] = (1.f / (1.f + expf(-xi)) - ti) * gscale;
        }
    }
    local = dlrm_wave_sum(local);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = local;
    __syncthreads();
    if (threadIdx.x == 0) {
        float s = 0.f;
#pragma unroll
        for (int k = 0; k < kLossBlock / 64; ++k) s += red[k];
        partials[blockIdx.x] = s;
    }
}

__global__ __launch_bounds__(kLossBlock) void mse_kernel(long long B, const float* __restrict__ p,
                                                         const float* __restrict__ t, float gscale,
                                                         float* __restrict__ dp, float* __restrict__ partials) {
    __shared__ float red[kLossBlock / 64];
    float local = 0.f;
    const long long base = ((long long)blockIdx.x * kLossBlock + threadIdx.x) * kLossPerThread;
#pragma unroll
    for (int k = 0; k < kLossPerThread; ++k) {
        const long long i = base + k;
        if (i < B) {
            const float d = p[i] - t[i];
            local += d * d;
            if (dp) dp[i] = 2.f * d * gscale;
        }
    }
    local = dlrm_wave_sum(local);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = local;
    __syncthreads();
    if (threadIdx.x == 0) {
        float s = 0.f;
#pragma unroll
        for (int k = 0; k < kLossBlock / 64; ++k) s += red[k];
        partials[blockIdx.x] = s;
    }
}

// deterministic final reduction of the per-block partials in fp64, then the mean
__global__ __launch_bounds__(256) void loss_finish_kernel(int n, const float* __restrict__ partials, double inv_count,
                                                          float* __restrict__ out) {
    __shared__ double red[256];
    double s = 0.0;
    for (int i = threadIdx.x; i < n; i += 256) s += (double)partials[i];
    red[threadIdx.x] = s;
    __syncthreads();
    for (int k = 128; k > 0; k >>= 1) {
        if (threadIdx.x < k) red[threadIdx.x] += red[threadIdx.x + k];
        __syncthreads();
    }
    if (threadIdx.x == 0) out[0] = (float)(red[0] * inv_count);
}

__global__ __launch_bounds__(256) void sgd_dense_kernel(long long n4, long long n, float4* __restrict__ w4,
                                                        const float4* __restrict__ g4, float* __restrict__ w,
                                                        const float* __restrict__ g, DlrmStep neg_lr_) {
    const float neg_lr = neg_lr_;        // (by value, or read from the device scalar: common.h DlrmStep)
    const long long stride = (long long)gridDim.x * 256;
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < n4; i += stride) {
        float4 a = w4[i];
        const float4 b = g4[i];
        a.x = __builtin_fmaf(neg_lr, b.x, a.x); a.y = __builtin_fmaf(neg_lr, b.y, a.y);
        a.z = __builtin_fmaf(neg_lr, b.z, a.z); a.w = __builtin_fmaf(neg_lr, b.w, a.w);
        w4[i] = a;
    }
    for (long long i = n4 * 4 + (long long)blockIdx.x * 256 + threadIdx.x; i < n; i += stride)
        w[i] = __builtin_fmaf(neg_lr, g[i], w[i]);
}

// y[i] = x[i] * s[0]   (s = the upstream gradient of the scalar loss: a device scalar, no host read)
__global__ __launch_bounds__(256) void scale_kernel(long long n, const float* __restrict__ x, const float* __restrict__ s,
                                                    float* __restrict__ y) {
    const float f = s[0];
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long long)gridDim.x * 256) y[i] = x[i] * f;
}

// dst[m, 0:K] = src[m, 0:K], dst[m, K:Kp] = 0: 16-byte aligned rows for operands whose width is not a multiple of 4
// (13 dense features, 479 interaction outputs and the first-layer weights that multiply them)
__global__ __launch_bounds__(256) void pad_cols_kernel(long long M, int K, int Kp, const float* __restrict__ src, long long lds_,
                                                       float* __restrict__ dst, long long ldd) {
    const long long total = M * Kp;
    for (long long e = (long long)blockIdx.x * 256 + threadIdx.x; e < total; e += (long long)gridDim.x * 256) {
        const long long m = e / Kp;
        const int k = (int)(e - m * Kp);
        dst[m * ldd + k] = k < K ? src[m * lds_ + k] : 0.f;
    }
}

// dense SGD for MANY parameter tensors in one launch (the 16 weights/biases of the two towers): pointers and sizes
// travel by value in the kernarg segment, every workgroup owns one 4096-float chunk of one tensor.
#define DLRM_MAX_DENSE_TENSORS 48
constexpr int kSgdChunk = 4096;
struct MultiSgdArgs {
    float*       w[DLRM_MAX_DENSE_TENSORS];
    const float* g[DLRM_MAX_DENSE_TENSORS];
    long long    n[DLRM_MAX_DENSE_TENSORS];
    int          blk0[DLRM_MAX_DENSE_TENSORS + 1];     // first workgroup of tensor k; blk0[count] = grid size
    int          count;
};

__global__ __launch_bounds__(256) void sgd_dense_multi_kernel(MultiSgdArgs a, DlrmStep neg_lr_) {
    const float neg_lr = neg_lr_;        // (by value, or read from the device scalar: common.h DlrmStep)
    int k = 0;
    while (k + 1 < a.count && (int)blockIdx.x >= a.blk0[k + 1]) ++k;     // wave-uniform scan over <= 48 entries
    float* __restrict__ w = a.w[k];
    const float* __restrict__ g = a.g[k];
    const long long n = a.n[k];
    const long long e0 = (long long)((int)blockIdx.x - a.blk0[k]) * kSgdChunk;
    const long long e1 = (e0 + kSgdChunk < n) ? e0 + kSgdChunk : n;
    if (((((uintptr_t)w) | ((uintptr_t)g)) & 15u) == 0) {
        for (long long i = e0 + 4 * threadIdx.x; i + 3 < e1; i += 4 * 256) {
            float4 x = *(const float4*)(w + i);
            const float4 y = *(const float4*)(g + i);
            x.x = __builtin_fmaf(neg_lr, y.x, x.x); x.y = __builtin_fmaf(neg_lr, y.y, x.y);
            x.z = __builtin_fmaf(neg_lr, y.z, x.z); x.w = __builtin_fmaf(neg_lr, y.w, x.w);
            *(float4*)(w + i) = x;
        }
        const long long tail = e0 + ((e1 - e0) & ~3LL);                  // e0 is a multiple of 4
        for (long long i = tail + threadIdx.x; i < e1; i += 256) w[i] = __builtin_fmaf(neg_lr, g[i], w[i]);
    } else {
        for (long long i = e0 + threadIdx.x; i < e1; i += 256) w[i] = __builtin_fmaf(neg_lr, g[i], w[i]);
    }
}

// strided block copy: dst_k[m, 0:w_k] = src_k[m, 0:w_k] for nblk (pointer, row stride) pairs — torch.cat / torch.split along
// dim 1 without ATen: the "cat" interaction (dlrm_s_pytorch.py:505-507) and the re-layout of all-to-all receive
// buffers (extend_distributed.py:446-465) in both directions.
#define DLRM_MAX_COPY_BLOCKS 64
struct CopyBlocksArgs {
    const float* src[DLRM_MAX_COPY_BLOCKS]; long long src_ld[DLRM_MAX_COPY_BLOCKS];
    float*       dst[DLRM_MAX_COPY_BLOCKS]; long long dst_ld[DLRM_MAX_COPY_BLOCKS];
    int          width[DLRM_MAX_COPY_BLOCKS];
};

__global__ __launch_bounds__(256) void copy_blocks_kernel(CopyBlocksArgs a, long long M) {
    const int k = blockIdx.y;
    const int w = a.width[k];
    const float* __restrict__ src = a.src[k];
    float* __restrict__ dst = a.dst[k];
    const long long sld = a.src_ld[k], dld = a.dst_ld[k];
    const bool v4 = (w % 4 == 0) && (sld % 4 == 0) && (dld % 4 == 0) && ((((uintptr_t)src) | ((uintptr_t)dst)) & 15u) == 0;
    if (v4) {
        const int w4 = w / 4;
        const long long total = M * w4;
        for (long long e = (long long)blockIdx.x * 256 + threadIdx.x; e < total; e += (long long)gridDim.x * 256) {
            const long long m = e / w4; const int c = (int)(e - m * w4) * 4;
            *(float4*)(dst + m * dld + c) = *(const float4*)(src + m * sld + c);
        }
    } else {
        const long long total = M * w;
        for (long long e = (long long)blockIdx.x * 256 + threadIdx.x; e < total; e += (long long)gridDim.x * 256) {
            const long long m = e / w; const int c = (int)(e - m * w);
            dst[m * dld + c] = src[m * sld + c];
        }
    }
}

// BCELoss(reduction="none") — the per-sample loss `wbce` needs (dlrm_s_pytorch.py:388-391, 150-156) — and its backward
__global__ __launch_bounds__(256) void bce_elementwise_kernel(long long n, const float* __restrict__ p, const float* __restrict__ t,
                                                              float* __restrict__ loss) {
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long long)gridDim.x * 256) {
        const float pi = p[i], ti = t[i];
        loss[i] = -(ti * fmaxf(logf(pi), -100.f) + (1.f - ti) * fmaxf(log1pf(-pi), -100.f));
    }
}
__global__ __launch_bounds__(256) void bce_elementwise_bwd_kernel(long long n, const float* __restrict__ p, const float* __restrict__ t,
                                                                  const float* __restrict__ dloss, float* __restrict__ dp) {
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long long)gridDim.x * 256) {
        const float pi = p[i];
        dp[i] = dloss[i] * (pi - t[i]) / fmaxf((1.f - pi) * pi, 1e-12f);
    }
}
// DCN-v2 low-rank cross layer, elementwise part (torchrec LowRankCrossNet: x_{l+1} = x_0 * (W_l (V_l x_l) + b_l) + x_l; the two
// products are dlrm_linear_fwd calls): forward out = x0 * u + xl; backward du = g * x0 and dx0 (+)= g * u.
typedef __attribute__((ext_vector_type(2))) __bf16 lo_bf16x2;
__device__ __forceinline__ unsigned lo_cvt_pk_bf16(float lo, float hi) { const lo_bf16x2 v = {(__bf16)lo, (__bf16)hi}; return __builtin_bit_cast(unsigned, v); }
// out = x0 * u + xl;  out16 (nullable) = bf16(out): the operand copy the next cross layer's GEMM reads (bf16-storage towers)
__global__ __launch_bounds__(256) void cross_fwd_kernel(long long n4, const float4* __restrict__ x0, const float4* __restrict__ u,
                                                        const float4* __restrict__ xl, float4* __restrict__ out, uint2* __restrict__ out16) {
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < n4; i += (long long)gridDim.x * 256) {
        const float4 a = x0[i], b = u[i], c = xl[i];
        const float4 r = make_float4(__builtin_fmaf(a.x, b.x, c.x), __builtin_fmaf(a.y, b.y, c.y), __builtin_fmaf(a.z, b.z, c.z), __builtin_fmaf(a.w, b.w, c.w));
        out[i] = r;
        if (out16) out16[i] = make_uint2(lo_cvt_pk_bf16(r.x, r.y), lo_cvt_pk_bf16(r.z, r.w));
    }
}
// du = g * x0 (fp32, nullable) and / or du16 = bf16(g * x0) (nullable);  dx0 (+)= g * u
// (u as stored by the forward pass: fp32, or — U16 — the bf16 copy the fused cross GEMM wrote: a bf16 value IS the upper half of its fp32 form)
template <bool U16>
__global__ __launch_bounds__(256) void cross_bwd_kernel(long long n4, const float4* __restrict__ g, const float4* __restrict__ x0,
                                                        const void* __restrict__ uv, float4* __restrict__ du, uint2* __restrict__ du16,
                                                        float4* __restrict__ dx0, int accumulate) {
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < n4; i += (long long)gridDim.x * 256) {
        const float4 gg = g[i], a = x0[i];
        float4 b;
        if constexpr (U16) {
            const uint2 w = ((const uint2*)uv)[i];
            b = make_float4(__uint_as_float(w.x << 16), __uint_as_float(w.x & 0xffff0000u), __uint_as_float(w.y << 16), __uint_as_float(w.y & 0xffff0000u));
        } else {
            b = ((const float4*)uv)[i];
        }
        const float4 r = make_float4(gg.x * a.x, gg.y * a.y, gg.z * a.z, gg.w * a.w);
        if (du) du[i] = r;
        if (du16) du16[i] = make_uint2(lo_cvt_pk_bf16(r.x, r.y), lo_cvt_pk_bf16(r.z, r.w));
        float4 d = accumulate ? dx0[i] : make_float4(0.f, 0.f, 0.f, 0.f);
        d.x = __builtin_fmaf(gg.x, b.x, d.x); d.y = __builtin_fmaf(gg.y, b.y, d.y); d.z = __builtin_fmaf(gg.z, b.z, d.z); d.w = __builtin_fmaf(gg.w, b.w, d.w);
        dx0[i] = d;
    }
}
__global__ __launch_bounds__(256) void add_kernel(long long n4, const float4* __restrict__ a, const float4* __restrict__ b, float4* __restrict__ out) {
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < n4; i += (long long)gridDim.x * 256) {
        const float4 x = a[i], y = b[i];
        out[i] = make_float4(x.x + y.x, x.y + y.y, x.z + y.z, x.w + y.w);
    }
}

// torch.clamp(x, lo, hi) of the predictions (--loss-threshold, dlrm_s_pytorch.py:580-583,607-610) and its backward
// (gradient passes where lo <= x <= hi, torch's clamp_backward mask)
__global__ __launch_bounds__(256) void clamp_kernel(long long n, const float* __restrict__ x, float lo, float hi, float* __restrict__ y) {
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long long)gridDim.x * 256)
        y[i] = fminf(fmaxf(x[i], lo), hi);
}
__global__ __launch_bounds__(256) void clamp_bwd_kernel(long long n, const float* __restrict__ x, float lo, float hi,
                                                        const float* __restrict__ dy, float* __restrict__ dx) {
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long long)gridDim.x * 256) {
        const float v = x[i];
        dx[i] = (v >= lo && v <= hi) ? dy[i] : 0.f;
    }
}

}  // namespace

extern "C" int dlrm_hip_abi_version(void) { return 17; }

extern "C" const char* dlrm_hip_build_info(void) {
    return "libdlrm_hip gfx950 (CDNA4) " __DATE__ " " __TIME__ " clang " __clang_version__;
}

extern "C" int dlrm_hip_device_info(int device, int* cu_count, int* lds_bytes, int64_t* hbm_bytes, char* name,
                                    int name_len) {
    hipDeviceProp_t p;
    hipError_t e = hipGetDeviceProperties(&p, device);
    if (e != hipSuccess) return (int)e;
    if (cu_count) *cu_count = p.multiProcessorCount;
    if (lds_bytes) *lds_bytes = (int)p.maxSharedMemoryPerMultiProcessor;
    if (hbm_bytes) *hbm_bytes = (int64_t)p.totalGlobalMem;
    if (name && name_len > 0) { snprintf(name, (size_t)name_len, "%s (%s)", p.name, p.gcnArchName); }
    return 0;
}

// One replay of a captured training step (dlrm_amd.graph.GraphedTrainStep) from ONE host call: wait for the previous replay, copy the
// step's inputs into the graph's static buffers, launch.  At Criteo-Kaggle shapes the GPU finishes a replay in ~0.3 ms and then waits for
// the host; the same sequence issued from Python (stream.synchronize(), one copy_ per input, CUDAGraph.replay()) was ~60 us of that wait.
namespace {
#define DLRM_MAX_SCALARS 16
struct SetF32Args { float* dst[DLRM_MAX_SCALARS]; float v[DLRM_MAX_SCALARS]; int n; };
__global__ void set_f32_kernel(SetF32Args a) { if ((int)threadIdx.x < a.n) *a.dst[threadIdx.x] = a.v[threadIdx.x]; }
}  // namespace

// dst_host[i][0] = values_host[i] for i < n (n <= 16), values BY VALUE in the kernarg: no host buffer has to outlive the call, nothing is
// copied from pageable memory.  The device scalars the update kernels read their learning rate from (DlrmStep, common.h).
extern "C" int dlrm_set_f32(int n, float* const* dst_host, const float* values_host, void* stream) {
    if (n < 0 || n > DLRM_MAX_SCALARS || (n > 0 && (!dst_host || !values_host))) return DLRM_E_ARG;
    if (n == 0) return 0;
    SetF32Args a = {};
    a.n = n;
    for (int i = 0; i < n; ++i) { if (!dst_host[i]) return DLRM_E_ARG; a.dst[i] = dst_host[i]; a.v[i] = values_host[i]; }
    hipLaunchKernelGGL(set_f32_kernel, dim3(1), dim3(64), 0, (hipStream_t)stream, a);
    DLRM_LAUNCH_CHECK();
    return 0;
}

extern "C" int dlrm_graph_replay(int n, void* const* dst_host, const void* const* src_host, const int64_t* bytes_host,
                                 int n_scalars, float* const* scalar_dst_host, const float* scalar_values_host, void* graph_exec,
                                 int sync_first, void* stream) {
    if (n < 0 || !graph_exec || (n > 0 && (!dst_host || !src_host || !bytes_host))) return DLRM_E_ARG;
    hipStream_t st = (hipStream_t)stream;
    if (sync_first) {
        // the previous replay is POLLED for (hipStreamQuery), not slept on: how long a thread blocked in hipStreamSynchronize takes to wake up
        // after the GPU signalled is the box's business (30 us ... 1.2 ms measured on this pool, profiles/round6/proof_wait.md) and a
        // launch-bound step (Criteo-Kaggle: 0.32 ms per replay) pays it every step.  The wait is one step long; after 50 ms the thread sleeps.
        hipError_t e = hipStreamQuery(st);
        if (e == hipErrorNotReady) {
            const auto t_end = std::chrono::steady_clock::now() + std::chrono::milliseconds(50);
            while ((e = hipStreamQuery(st)) == hipErrorNotReady)
                if (std::chrono::steady_clock::now() > t_end) { e = hipStreamSynchronize(st); break; }
        }
        if (e != hipSuccess) return (int)e;
    }
    for (int i = 0; i < n; ++i) {
        if (!dst_host[i] || !src_host[i] || bytes_host[i] < 0) return DLRM_E_ARG;
        if (dst_host[i] == src_host[i] || bytes_host[i] == 0) continue;
        hipError_t e = hipMemcpyAsync(dst_host[i], src_host[i], (size_t)bytes_host[i], hipMemcpyDeviceToDevice, st);
        if (e != hipSuccess) return (int)e;
    }
    if (n_scalars > 0) {          // the step sizes of this replay (a schedule changed them): written behind the previous replay, in front of this one
        const int rc = dlrm_set_f32(n_scalars, scalar_dst_host, scalar_values_host, stream);
        if (rc) return rc;
    }
    hipError_t e = hipGraphLaunch((hipGraphExec_t)graph_exec, st);
    return e == hipSuccess ? 0 : (int)e;
}

extern "C" int64_t dlrm_loss_workspace_bytes(int64_t B) {
    const int64_t per_block = (int64_t)kLossBlock * kLossPerThread;
    return ((B + per_block - 1) / per_block) * (int64_t)sizeof(float);
}

extern "C" int dlrm_bce_loss(int64_t B, const float* p, const float* target, const float* weights, float w_neg, float w_pos,
                             float grad_scale, float* loss_out, float* dp, void* partials, void* stream) {
    if (B <= 0 || !p || !target || !loss_out || !partials) return DLRM_E_ARG;
    const int64_t per_block = (int64_t)kLossBlock * kLossPerThread;
    const int nblk = (int)((B + per_block - 1) / per_block);
    hipStream_t st = (hipStream_t)stream;
    hipLaunchKernelGGL(bce_kernel, dim3(nblk), dim3(kLossBlock), 0, st, (long long)B, p, target, weights, w_neg, w_pos,
                       grad_scale / (float)B, dp, (float*)partials);
    DLRM_LAUNCH_CHECK();
    hipLaunchKernelGGL(loss_finish_kernel, dim3(1), dim3(256), 0, st, nblk, (const float*)partials, 1.0 / (double)B, loss_out);
    DLRM_LAUNCH_CHECK();
    return 0;
}

extern "C" int dlrm_bce_logits_loss(int64_t B, const float* logits, const float* target, float grad_scale,
                                    float* loss_out, float* dlogits, void* partials, void* stream) {
    if (B <= 0 || !logits || !target || !loss_out || !partials) return DLRM_E_ARG;
    const int64_t per_block = (int64_t)kLossBlock * kLossPerThread;
    const int nblk = (int)((B + per_block - 1) / per_block);
    hipStream_t st = (hipStream_t)stream;
    hipLaunchKernelGGL(bce_logits_kernel, dim3(nblk), dim3(kLossBlock), 0, st, (long long)B, logits, target,
                       grad_scale / (float)B, dlogits, (float*)partials);
    DLRM_LAUNCH_CHECK();
    hipLaunchKernelGGL(loss_finish_kernel, dim3(1), dim3(256), 0, st, nblk, (const float*)partials, 1.0 / (double)B, loss_out);
    DLRM_LAUNCH_CHECK();
    return 0;
}

extern "C" int dlrm_mse_loss(int64_t B, const float* p, const float* target, float grad_scale,
                             float* loss_out, float* dp, void* partials, void* stream) {
    if (B <= 0 || !p || !target || !loss_out || !partials) return DLRM_E_ARG;
    const int64_t per_block = (int64_t)kLossBlock * kLossPerThread;
    const int nblk = (int)((B + per_block - 1) / per_block);
    hipStream_t st = (hipStream_t)stream;
    hipLaunchKernelGGL(mse_kernel, dim3(nblk), dim3(kLossBlock), 0, st, (long long)B, p, target,
                       grad_scale / (float)B, dp, (float*)partials);
    DLRM_LAUNCH_CHECK();
    hipLaunchKernelGGL(loss_finish_kernel, dim3(1), dim3(256), 0, st, nblk, (const float*)partials, 1.0 / (double)B, loss_out);
    DLRM_LAUNCH_CHECK();
    return 0;
}

extern "C" int dlrm_sgd_dense(int64_t n, float* w, const float* g, float lr, const float* lr_dev, void* stream) {
    if (n <= 0 || !w || !g) return DLRM_E_ARG;
    const bool vec = dlrm_aligned16(w) && dlrm_aligned16(g);
    const long long n4 = vec ? n / 4 : 0;
    long long nblk = (n / 4 + 255) / 256; if (nblk < 1) nblk = 1; if (nblk > 2048) nblk = 2048;
    hipLaunchKernelGGL(sgd_dense_kernel, dim3((unsigned)nblk), dim3(256), 0, (hipStream_t)stream, n4, (long long)n,
                       (float4*)w, (const float4*)g, w, g, dlrm_step_neg(lr, lr_dev));
    DLRM_LAUNCH_CHECK();
    return 0;
}

extern "C" int dlrm_scale_by_device_scalar(int64_t n, const float* x, const float* scalar_dev, float* y, void* stream) {
    if (n <= 0 || !x || !scalar_dev || !y) return DLRM_E_ARG;
    long long nblk = (n + 255) / 256; if (nblk > 1024) nblk = 1024;
    hipLaunchKernelGGL(scale_kernel, dim3((unsigned)nblk), dim3(256), 0, (hipStream_t)stream, (long long)n, x, scalar_dev, y);
    DLRM_LAUNCH_CHECK();
    return 0;
}

extern "C" int dlrm_pad_cols(int64_t M, int K, int Kp, const float* src, int64_t ld_src, float* dst, int64_t ld_dst, void* stream) {
    if (M <= 0 || K <= 0 || Kp < K || !src || !dst || ld_src < K || ld_dst < Kp) return DLRM_E_ARG;
    long long nblk = ((long long)M * Kp + 255) / 256; if (nblk > 4096) nblk = 4096;
    hipLaunchKernelGGL(pad_cols_kernel, dim3((unsigned)nblk), dim3(256), 0, (hipStream_t)stream, (long long)M, K, Kp, src,
                       (long long)ld_src, dst, (long long)ld_dst);
    DLRM_LAUNCH_CHECK();
    return 0;
}

extern "C" int dlrm_sgd_dense_multi(int count, float* const* w_host, const float* const* g_host, const int64_t* n_host,
                                    float lr, const float* lr_dev, void* stream) {
    if (count <= 0 || !w_host || !g_host || !n_host) return DLRM_E_ARG;
    for (int c0 = 0; c0 < count; c0 += DLRM_MAX_DENSE_TENSORS) {
        MultiSgdArgs a = {};
        const int m = (count - c0 < DLRM_MAX_DENSE_TENSORS) ? count - c0 : DLRM_MAX_DENSE_TENSORS;
        long long blocks = 0;
        for (int k = 0; k < m; ++k) {
            if (!w_host[c0 + k] || !g_host[c0 + k] || n_host[c0 + k] <= 0) return DLRM_E_ARG;
            a.w[k] = w_host[c0 + k]; a.g[k] = g_host[c0 + k]; a.n[k] = n_host[c0 + k];
            a.blk0[k] = (int)blocks;
            blocks += (n_host[c0 + k] + kSgdChunk - 1) / kSgdChunk;
            if (blocks > 0x7fffffffLL) return DLRM_E_RANGE;
        }
        a.blk0[m] = (int)blocks; a.count = m;
        hipLaunchKernelGGL(sgd_dense_multi_kernel, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, a, dlrm_step_neg(lr, lr_dev));
        DLRM_LAUNCH_CHECK();
    }
    return 0;
}

extern "C" int dlrm_copy_blocks(int64_t M, int nblk, const void* const* src_host, const int64_t* src_ld_host,
                                void* const* dst_host, const int64_t* dst_ld_host, const int* width_host, void* stream) {
    if (M <= 0 || nblk <= 0 || !src_host || !src_ld_host || !dst_host || !dst_ld_host || !width_host) return DLRM_E_ARG;
    for (int k0 = 0; k0 < nblk; k0 += DLRM_MAX_COPY_BLOCKS) {
        const int n = (nblk - k0 < DLRM_MAX_COPY_BLOCKS) ? nblk - k0 : DLRM_MAX_COPY_BLOCKS;
        CopyBlocksArgs a = {};
        long long widest = 1;
        for (int k = 0; k < n; ++k) {
            const int w = width_host[k0 + k];
            if (!src_host[k0 + k] || !dst_host[k0 + k] || w <= 0 || src_ld_host[k0 + k] < w || dst_ld_host[k0 + k] < w) return DLRM_E_ARG;
            a.src[k] = (const float*)src_host[k0 + k]; a.src_ld[k] = src_ld_host[k0 + k];
            a.dst[k] = (float*)dst_host[k0 + k]; a.dst_ld[k] = dst_ld_host[k0 + k]; a.width[k] = w;
            if (w > widest) widest = w;
        }
        long long nblkx = (M * widest / 4 + 255) / 256; if (nblkx < 1) nblkx = 1; if (nblkx > 1024) nblkx = 1024;
        hipLaunchKernelGGL(copy_blocks_kernel, dim3((unsigned)nblkx, (unsigned)n), dim3(256), 0, (hipStream_t)stream, a, (long long)M);
        DLRM_LAUNCH_CHECK();
    }
    return 0;
}

static inline unsigned ew_blocks(int64_t n) { long long b = (n + 255) / 256; if (b > 2048) b = 2048; if (b < 1) b = 1; return (unsigned)b; }
// streaming kernels over hundreds of MB (the DCN-v2 elementwise halves): ONE item per thread — the copy sweep of round 4 measured 6.18 TB/s for that
// shape against 4.3-5.7 TB/s for grid-stride loops of any grid / unroll (profiles/round4/hbm_copy_sweep.txt); the kernels keep their loops, so any grid is valid
static inline unsigned ew_blocks_full(int64_t n) { long long b = (n + 255) / 256; if (b > 0x7fffffffll) b = 0x7fffffffll; if (b < 1) b = 1; return (unsigned)b; }

extern "C" int dlrm_bce_elementwise(int64_t n, const float* p, const float* target, float* loss, void* stream) {
    if (n <= 0 || !p || !target || !loss) return DLRM_E_ARG;
    hipLaunchKernelGGL(bce_elementwise_kernel, dim3(ew_blocks(n)), dim3(256), 0, (hipStream_t)stream, (long long)n, p, target, loss);
    DLRM_LAUNCH_CHECK();
    return 0;
}

extern "C" int dlrm_bce_elementwise_bwd(int64_t n, const float* p, const float* target, const float* dloss, float* dp, void* stream) {
    if (n <= 0 || !p || !target || !dloss || !dp) return DLRM_E_ARG;
    hipLaunchKernelGGL(bce_elementwise_bwd_kernel, dim3(ew_blocks(n)), dim3(256), 0, (hipStream_t)stream, (long long)n, p, target, dloss, dp);
    DLRM_LAUNCH_CHECK();
    return 0;
}

extern "C" int dlrm_clamp(int64_t n, const float* x, float lo, float hi, float* y, void* stream) {
    if (n <= 0 || !x || !y || !(lo <= hi)) return DLRM_E_ARG;
    hipLaunchKernelGGL(clamp_kernel, dim3(ew_blocks(n)), dim3(256), 0, (hipStream_t)stream, (long long)n, x, lo, hi, y);
    DLRM_LAUNCH_CHECK();
    return 0;
}

extern "C" int dlrm_clamp_bwd(int64_t n, const float* x, float lo, float hi, const float* dy, float* dx, void* stream) {
    if (n <= 0 || !x || !dy || !dx || !(lo <= hi)) return DLRM_E_ARG;
    hipLaunchKernelGGL(clamp_bwd_kernel, dim3(ew_blocks(n)), dim3(256), 0, (hipStream_t)stream, (long long)n, x, lo, hi, dy, dx);
    DLRM_LAUNCH_CHECK();
    return 0;
}

static inline bool vec4_ok(int64_t n, const void* a, const void* b, const void* c, const void* d, const void* e) {
    return n % 4 == 0 && dlrm_aligned16(a) && dlrm_aligned16(b) && (!c || dlrm_aligned16(c)) && (!d || dlrm_aligned16(d)) && (!e || dlrm_aligned16(e));
}

extern "C" int dlrm_cross_fwd(int64_t n, const float* x0, const float* u, const float* xl, float* out, uint16_t* out16, void* stream) {
    if (n <= 0 || !x0 || !u || !xl || !out) return DLRM_E_ARG;
    if (!vec4_ok(n, x0, u, xl, out, nullptr) || (out16 && (((uintptr_t)out16) & 7u))) return DLRM_E_ALIGN;
    hipLaunchKernelGGL(cross_fwd_kernel, dim3(ew_blocks_full(n / 4)), dim3(256), 0, (hipStream_t)stream, (long long)(n / 4), (const float4*)x0,
                       (const float4*)u, (const float4*)xl, (float4*)out, (uint2*)out16);
    DLRM_LAUNCH_CHECK();
    return 0;
}

extern "C" int dlrm_cross_bwd(int64_t n, const float* g, const float* x0, const float* u, const uint16_t* u16, float* du, uint16_t* du16, float* dx0,
                              int accumulate, void* stream) {
    if (n <= 0 || !g || !x0 || (!u == !u16) || (!du && !du16) || !dx0) return DLRM_E_ARG;
    if (!vec4_ok(n, g, x0, u, du, dx0) || (du16 && (((uintptr_t)du16) & 7u)) || (u16 && (((uintptr_t)u16) & 7u))) return DLRM_E_ALIGN;
    if (u16)
        hipLaunchKernelGGL(cross_bwd_kernel<true>, dim3(ew_blocks_full(n / 4)), dim3(256), 0, (hipStream_t)stream, (long long)(n / 4), (const float4*)g,
                           (const float4*)x0, (const void*)u16, (float4*)du, (uint2*)du16, (float4*)dx0, accumulate ? 1 : 0);
    else
        hipLaunchKernelGGL(cross_bwd_kernel<false>, dim3(ew_blocks_full(n / 4)), dim3(256), 0, (hipStream_t)stream, (long long)(n / 4), (const float4*)g,
                           (const float4*)x0, (const void*)u, (float4*)du, (uint2*)du16, (float4*)dx0, accumulate ? 1 : 0);
    DLRM_LAUNCH_CHECK();
    return 0;
}

extern "C" int dlrm_add(int64_t n, const float* a, const float* b, float* out, void* stream) {
    if (n <= 0 || !a || !b || !out) return DLRM_E_ARG;
    if (!vec4_ok(n, a, b, out, nullptr, nullptr)) return DLRM_E_ALIGN;
    hipLaunchKernelGGL(add_kernel, dim3(ew_blocks_full(n / 4)), dim3(256), 0, (hipStream_t)stream, (long long)(n / 4), (const float4*)a,
                       (const float4*)b, (float4*)out);
    DLRM_LAUNCH_CHECK();
    return 0;
}
