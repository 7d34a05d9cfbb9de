// hbm_peak.hip — what HBM3E on this MI355X sustains for the access mixes of the embedding / interaction kernels:
// pure streaming read, pure write, 1:1 copy, 7:1 read:write, and 512-byte random row gathers (spec: 8 TB/s).
// build: hipcc --offload-arch=gfx950 -O3 tools/probes/hbm_peak.hip -o tools/probes/hbm_peak
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>

__global__ __launch_bounds__(256) void k_read(const float4* __restrict__ a, size_t n, float* out) {
    float4 s = make_float4(0, 0, 0, 0);
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) { float4 v = a[i]; s.x += v.x; s.y += v.y; s.z += v.z; s.w += v.w; }
    if (s.x + s.y + s.z + s.w == 123.456f) out[0] = s.x;
}
__global__ __launch_bounds__(256) void k_write(float4* __restrict__ a, size_t n) {
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) a[i] = make_float4(1.f, 2.f, 3.f, 4.f);
}
__global__ __launch_bounds__(256) void k_copy(const float4* __restrict__ a, float4* __restrict__ b, size_t n) {
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) b[i] = a[i];
}
// 7 reads : 1 write (the interaction forward's mix): one of every 8 256-thread stripes is also written
__global__ __launch_bounds__(256) void k_mix71(const float4* __restrict__ a, float4* __restrict__ b, size_t n) {
    float4 s = make_float4(0, 0, 0, 0);
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) {
        float4 v = a[i]; s.x += v.x;
        if (((i >> 8) & 7) == 0) b[((i >> 11) << 8) + (i & 255)] = v;
    }
    if (s.x == 123.456f) b[0] = s;
}
// random 512-byte rows: a half-wave (32 lanes x 16 B) reads one row; rows picked by a multiplicative hash
__global__ __launch_bounds__(256) void k_gather512(const float4* __restrict__ a, size_t rows, size_t nrows_read, float* out) {
    float4 s = make_float4(0, 0, 0, 0);
    const size_t g0 = ((size_t)blockIdx.x * 256 + threadIdx.x) >> 5;
    const int l = threadIdx.x & 31;
    for (size_t g = g0; g < nrows_read; g += ((size_t)gridDim.x * 256) >> 5) {
        const size_t r = ((g * 0x9E3779B97F4A7C15ull) >> 20) % rows;
        float4 v = a[r * 32 + l]; s.x += v.x; s.y += v.y;
    }
    if (s.x + s.y == 123.456f) out[0] = s.x;
}

int main() {
    const size_t bytes = 4ull << 30, n = bytes / 16;
    float4 *a, *b; float* out;
    hipMalloc(&a, bytes); hipMalloc(&b, bytes); hipMalloc(&out, 64);
    hipMemset(a, 0, bytes); hipMemset(b, 0, bytes);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    auto timeit = [&](const char* name, double moved, auto launch) {
        launch(); hipDeviceSynchronize();
        hipEventRecord(e0); for (int i = 0; i < 5; ++i) launch(); hipEventRecord(e1); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1); ms /= 5;
        printf("%-36s %8.3f ms  %7.1f GB/s\n", name, ms, moved / ms / 1e6);
    };
    for (int grid : {2048, 8192, 32768}) {
        printf("-- grid %d x 256\n", grid);
        timeit("streaming read", (double)bytes, [&] { hipLaunchKernelGGL(k_read, dim3(grid), dim3(256), 0, 0, a, n, out); });
        timeit("streaming write", (double)bytes, [&] { hipLaunchKernelGGL(k_write, dim3(grid), dim3(256), 0, 0, b, n); });
        timeit("copy (read + write counted)", 2.0 * bytes, [&] { hipLaunchKernelGGL(k_copy, dim3(grid), dim3(256), 0, 0, a, b, n); });
        timeit("7:1 read:write", bytes * 1.125, [&] { hipLaunchKernelGGL(k_mix71, dim3(grid), dim3(256), 0, 0, a, b, n); });
        timeit("random 512-B row gather (4M rows)", 4194304.0 * 512, [&] { hipLaunchKernelGGL(k_gather512, dim3(grid), dim3(256), 0, 0, a, bytes / 512, (size_t)4194304, out); });
    }
    return 0;
}
