// hbm_copy_sweep.hip — which float4 copy kernel shape sustains the most HBM bandwidth on this box (feeds dlrm_calib_hbm_copy's shape).
// build + run on the GPU box: hipcc --offload-arch=gfx950 -O3 tools/probes/hbm_copy_sweep.hip -o /tmp/hbm_copy_sweep && /tmp/hbm_copy_sweep
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef float f4 __attribute__((ext_vector_type(4)));
#define float4 f4
template <int U, bool NT>
__global__ __launch_bounds__(256) void k_copy(const float4* __restrict__ a, float4* __restrict__ b, size_t n) {
    const size_t stride = (size_t)gridDim.x * 256;
    size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    for (; i + (U - 1) * stride < n; i += U * stride) {
        float4 v[U];
#pragma unroll
        for (int u = 0; u < U; ++u) v[u] = NT ? __builtin_nontemporal_load(a + i + u * stride) : a[i + u * stride];
#pragma unroll
        for (int u = 0; u < U; ++u) { if (NT) __builtin_nontemporal_store(v[u], b + i + u * stride); else b[i + u * stride] = v[u]; }
    }
    for (; i < n; i += stride) b[i] = a[i];
}
// contiguous chunk per block (each block streams its own 1/grid of the buffer)
template <int U>
__global__ __launch_bounds__(256) void k_copy_chunk(const float4* __restrict__ a, float4* __restrict__ b, size_t n) {
    const size_t per = (n + gridDim.x - 1) / gridDim.x, lo = per * blockIdx.x, hi = lo + per < n ? lo + per : n;
    for (size_t i = lo + threadIdx.x; i < hi; i += 256 * U) {
        float4 v[U];
#pragma unroll
        for (int u = 0; u < U; ++u) if (i + u * 256 < hi) v[u] = a[i + u * 256];
#pragma unroll
        for (int u = 0; u < U; ++u) if (i + u * 256 < hi) b[i + u * 256] = v[u];
    }
}
int main() {
    const size_t bytes = 1ull << 30, n = bytes / 16;
    float4 *a, *b; hipMalloc(&a, bytes); hipMalloc(&b, bytes); hipMemset(a, 0, bytes); hipMemset(b, 0, bytes);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    auto timeit = [&](const char* name, int grid, auto launch) {
        for (int i = 0; i < 3; ++i) launch(); hipDeviceSynchronize();
        hipEventRecord(e0); for (int i = 0; i < 30; ++i) launch(); hipEventRecord(e1); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1); ms /= 30;
        printf("%-28s grid %6d  %7.3f ms  %7.1f GB/s\n", name, grid, ms, 2.0 * bytes / ms / 1e6);
    };
    for (int grid : {1024, 2048, 4096, 8192, 16384, 65536}) {
        timeit("stride U1", grid, [&] { hipLaunchKernelGGL((k_copy<1, false>), dim3(grid), dim3(256), 0, 0, a, b, n); });
        timeit("stride U4", grid, [&] { hipLaunchKernelGGL((k_copy<4, false>), dim3(grid), dim3(256), 0, 0, a, b, n); });
        timeit("stride U8", grid, [&] { hipLaunchKernelGGL((k_copy<8, false>), dim3(grid), dim3(256), 0, 0, a, b, n); });
        timeit("stride U4 nt", grid, [&] { hipLaunchKernelGGL((k_copy<4, true>), dim3(grid), dim3(256), 0, 0, a, b, n); });
        timeit("chunk U4", grid, [&] { hipLaunchKernelGGL((k_copy_chunk<4>), dim3(grid), dim3(256), 0, 0, a, b, n); });
    }
    // one thread per float4 (no loop)
    timeit("one float4 per thread", (int)(n / 256), [&] { hipLaunchKernelGGL((k_copy<1, false>), dim3((unsigned)(n / 256)), dim3(256), 0, 0, a, b, n); });
    hipMemcpyAsync(b, a, bytes, hipMemcpyDeviceToDevice, 0); hipDeviceSynchronize();
    hipEventRecord(e0); for (int i = 0; i < 30; ++i) hipMemcpyAsync(b, a, bytes, hipMemcpyDeviceToDevice, 0); hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1); ms /= 30;
    printf("%-28s             %7.3f ms  %7.1f GB/s\n", "hipMemcpyAsync D2D", ms, 2.0 * bytes / ms / 1e6);
    return 0;
}
