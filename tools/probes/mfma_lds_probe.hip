// mfma_lds_probe.hip — what the fp32 GEMM main loop can reach at best: v_mfma_f32_32x32x2_f32 streams fed the way gemm3_kernel feeds them.
// Question (round 6): the bare main loop of gemm3_kernel (no DMA, no barrier, no epilogue: DLRM_GEMM_DEBUG=7) runs at 0.91 of the MFMA peak with
// three waves per SIMD, and prefetching its fragment reads across k-tiles changes nothing.  Is it the ds_read traffic beside the MFMAs, the
// accumulators living in architectural VGPRs instead of AGPRs, or the number of independent accumulator chains?
//   variant 0: MFMAs only (operands loaded once)            variant 1: + (TM + TN) x 2 ds_read_b128 per 16-k tile, as the kernel
//   ACC = 'v' / 'a': accumulators in VGPRs / AGPRs           TM = 2 / 4 (TN = 2): 4 / 8 accumulator chains, 32 / 64 MFMAs per tile
// build: hipcc --offload-arch=gfx950 -O3 tools/probes/mfma_lds_probe.hip -o tools/probes/mfma_lds_probe
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
typedef float floatx16 __attribute__((ext_vector_type(16)));

template <bool AGPR>
__device__ __forceinline__ void mfma(floatx16& c, float a, float b) {
    if (AGPR) asm volatile("v_mfma_f32_32x32x2_f32 %0, %1, %2, %0" : "+a"(c) : "v"(a), "v"(b));
    else      asm volatile("v_mfma_f32_32x32x2_f32 %0, %1, %2, %0" : "+v"(c) : "v"(a), "v"(b));
}

template <int TM, bool AGPR, int READS>
__global__ __launch_bounds__(256, TM == 2 ? 4 : 2) void loop_kernel(float* out, const float* in, int iters, int lds_pad) {
    extern __shared__ __attribute__((aligned(16))) float lds[];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    for (int i = tid; i < 3 * 6144; i += 256) lds[i] = in[i];          // three stages of a 256 x 16 + 128 x 16 tile of random data
    __syncthreads();
    floatx16 acc[TM][2];
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
    const int l31 = lane & 31, h = lane >> 5;
    // the kernel's conflict-free fragment addresses (k-contiguous tile [rows][16], XOR-swizzled 16-byte slots)
    const char* base = (const char*)lds;
    unsigned fa_off[2], fb_off[2];
#pragma unroll
    for (int j = 0; j < 2; ++j) {
        fa_off[j] = (unsigned)(((wave >> 1) * 32 * TM + l31) * 64 + (((2 * j + h) ^ ((l31 >> 2) & 3)) * 16));
        fb_off[j] = (unsigned)(64 * TM * 64) + (unsigned)(((wave & 1) * 64 + l31) * 64 + (((2 * j + h) ^ ((l31 >> 2) & 3)) * 16));
    }
    float4 fa[2][TM], fb[2][2];
#pragma unroll
    for (int j = 0; j < 2; ++j) {
#pragma unroll
        for (int t = 0; t < TM; ++t) fa[j][t] = *(const float4*)(base + fa_off[j] + t * 2048);
#pragma unroll
        for (int t = 0; t < 2; ++t) fb[j][t] = *(const float4*)(base + fb_off[j] + t * 2048);
    }
    unsigned cur = 0;
    const unsigned stage = (64 * TM + 128) * 64;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            if (READS) {
#pragma unroll
                for (int t = 0; t < TM; ++t) fa[j][t] = *(const float4*)(base + cur + fa_off[j] + t * 2048);
#pragma unroll
                for (int t = 0; t < 2; ++t) fb[j][t] = *(const float4*)(base + cur + fb_off[j] + t * 2048);
            }
#pragma unroll
            for (int tm = 0; tm < TM; ++tm)
#pragma unroll
                for (int tn = 0; tn < 2; ++tn) mfma<AGPR>(acc[tm][tn], fb[j][tn].x, fa[j][tm].x);
#pragma unroll
            for (int tm = 0; tm < TM; ++tm)
#pragma unroll
                for (int tn = 0; tn < 2; ++tn) mfma<AGPR>(acc[tm][tn], fb[j][tn].y, fa[j][tm].y);
#pragma unroll
            for (int tm = 0; tm < TM; ++tm)
#pragma unroll
                for (int tn = 0; tn < 2; ++tn) mfma<AGPR>(acc[tm][tn], fb[j][tn].z, fa[j][tm].z);
#pragma unroll
            for (int tm = 0; tm < TM; ++tm)
#pragma unroll
                for (int tn = 0; tn < 2; ++tn) mfma<AGPR>(acc[tm][tn], fb[j][tn].w, fa[j][tm].w);
        }
        cur = (cur == 2 * stage) ? 0 : cur + stage;
    }
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) s += acc[i][j][r];
    if (s == 123.456f) out[0] = s;
}

template <int TM, bool AGPR, int READS>
static void run(const char* tag, float* out, const float* in, int cus, int wgs_per_cu) {
    const int iters = 4000;           // x (32 | 64) MFMAs per wave: ~3.4 / 6.8 ms alone on a SIMD
    const size_t lds = 160 * 1024 / wgs_per_cu > 73728 ? 73728 : 160 * 1024 / wgs_per_cu / 1024 * 1024;      // occupancy through the LDS size
    hipFuncSetAttribute((const void*)loop_kernel<TM, AGPR, READS>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    const size_t want = wgs_per_cu == 1 ? 160 * 1024 : wgs_per_cu == 2 ? 80 * 1024 : wgs_per_cu == 3 ? 53 * 1024 : 40 * 1024;
    (void)lds;
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    const int grid = cus * wgs_per_cu;
    for (int w = 0; w < 3; ++w) hipLaunchKernelGGL((loop_kernel<TM, AGPR, READS>), dim3(grid), dim3(256), want, 0, out, in, iters, 0);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    for (int r = 0; r < 5; ++r) hipLaunchKernelGGL((loop_kernel<TM, AGPR, READS>), dim3(grid), dim3(256), want, 0, out, in, iters, 0);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1); ms /= 5;
    const double flop = (double)grid * 4 * iters * (16.0 * TM) * 4096.0;
    printf("%-28s waves/SIMD %d  %.3f ms  %.1f TFLOP/s  %.3f of 157.3\n", tag, wgs_per_cu, ms, flop / ms / 1e9, flop / ms / 1e9 / 157.3);
}

int main() {
    float *out, *in;
    hipMalloc(&out, 4); hipMalloc(&in, 3 * 6144 * 4);
    float* h = (float*)malloc(3 * 6144 * 4);
    srand(1);
    for (int i = 0; i < 3 * 6144; ++i) h[i] = (float)rand() / RAND_MAX - 0.5f;
    hipMemcpy(in, h, 3 * 6144 * 4, hipMemcpyHostToDevice);
    hipDeviceProp_t p; hipGetDeviceProperties(&p, 0);
    const int cus = p.multiProcessorCount;
    printf("%s, %d CUs\n", p.name, cus);
    for (int w = 1; w <= 4; ++w) {
        run<2, false, 0>("TM2 acc=VGPR mfma only", out, in, cus, w);
        run<2, true, 0>("TM2 acc=AGPR mfma only", out, in, cus, w);
        run<2, false, 1>("TM2 acc=VGPR + ds_read", out, in, cus, w);
        run<2, true, 1>("TM2 acc=AGPR + ds_read", out, in, cus, w);
    }
    for (int w = 1; w <= 2; ++w) {
        run<4, false, 0>("TM4 acc=VGPR mfma only", out, in, cus, w);
        run<4, true, 0>("TM4 acc=AGPR mfma only", out, in, cus, w);
        run<4, false, 1>("TM4 acc=VGPR + ds_read", out, in, cus, w);
        run<4, true, 1>("TM4 acc=AGPR + ds_read", out, in, cus, w);
    }
    return 0;
}
