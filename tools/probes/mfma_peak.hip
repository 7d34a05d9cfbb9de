// mfma_peak.hip — sustained v_mfma_f32_32x32x2_f32 rate of the whole chip, no memory traffic: the practical ceiling the
// fp32 GEMM kernels can be compared with (spec: 256 CUs x 256 FLOP/clk x 2.4 GHz = 157.3 TFLOP/s).
// build: hipcc --offload-arch=gfx950 -O3 tools/probes/mfma_peak.hip -o tools/probes/mfma_peak     run: tools/probes/mfma_peak
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef float floatx16 __attribute__((ext_vector_type(16)));

template <int NACC>
__global__ __launch_bounds__(256) void mfma_loop(float* out, int iters, float a, float b) {
    floatx16 acc[NACC];
#pragma unroll
    for (int i = 0; i < NACC; ++i)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int k = 0; k < 8; ++k)
#pragma unroll
            for (int i = 0; i < NACC; ++i) acc[i] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc[i], 0, 0, 0);
    }
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < NACC; ++i)
#pragma unroll
        for (int r = 0; r < 16; ++r) s += acc[i][r];
    if (s == 123.456f) out[0] = s;
}

int main() {
    float* out; hipMalloc(&out, 4);
    hipDeviceProp_t p; hipGetDeviceProperties(&p, 0);
    const int cus = p.multiProcessorCount;
    printf("%s, %d CUs, clock %d kHz\n", p.name, cus, p.clockRate);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    for (int wgs_per_cu = 1; wgs_per_cu <= 2; ++wgs_per_cu)
        for (int iters : {200, 2000, 20000, 100000}) {
            const int grid = cus * wgs_per_cu;
            hipLaunchKernelGGL(mfma_loop<8>, dim3(grid), dim3(256), 0, 0, out, 10, 1.0f, 2.0f);   // warm
            hipDeviceSynchronize();
            hipEventRecord(e0);
            hipLaunchKernelGGL(mfma_loop<8>, dim3(grid), dim3(256), 0, 0, out, iters, 1.0f, 2.0f);
            hipEventRecord(e1); hipEventSynchronize(e1);
            float ms; hipEventElapsedTime(&ms, e0, e1);
            const double flop = (double)grid * 4 /*waves*/ * iters * 64.0 /*mfma*/ * 4096.0;
            printf("waves/SIMD %d  iters %6d  %.3f ms  %.1f TFLOP/s  (implied clock %.0f MHz at 256 FLOP/clk/CU)\n", wgs_per_cu, iters, ms,
                   flop / ms / 1e9, flop / ms / 1e9 * 1e12 / (cus * 256.0) / 1e6);
        }
    // back-to-back launches of ~1 ms (what a training step looks like), 20 of them
    hipEventRecord(e0);
    for (int r = 0; r < 20; ++r) hipLaunchKernelGGL(mfma_loop<8>, dim3(cus * 2), dim3(256), 0, 0, out, 20000, 1.0f, 2.0f);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    const double flop = 20.0 * cus * 2 * 4 * 20000.0 * 64 * 4096;
    printf("20 back-to-back launches: %.3f ms total, %.1f TFLOP/s sustained\n", ms, flop / ms / 1e9);
    return 0;
}
