// tr_read_probe.hip — what ds_read_b64_tr_b16 delivers: LDS holds element i at byte 2i; lane l passes the address of elements 4l..4l+3.
// Prints, per lane, the four values it received (= which (lane, element) of the plain ds_read_b64 they came from).
// hipcc --offload-arch=gfx950 -O3 -w tools/probes/tr_read_probe.hip -o /tmp/tr_read_probe && /tmp/tr_read_probe
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef short v4s __attribute__((ext_vector_type(4)));
__global__ void k(short* out) {
    __shared__ short lds[1024];
    for (int i = threadIdx.x; i < 1024; i += 64) lds[i] = (short)i;
    __syncthreads();
    v4s v = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) v4s*)(lds + 4 * threadIdx.x));
    for (int j = 0; j < 4; ++j) out[threadIdx.x * 4 + j] = v[j];
}
int main() {
    short* d; hipMalloc(&d, 512); hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, d); short h[256]; hipMemcpy(h, d, 512, hipMemcpyDeviceToHost);
    for (int l = 0; l < 64; ++l) printf("lane %2d: %4d %4d %4d %4d   (src lane,elem: %d.%d %d.%d %d.%d %d.%d)\n", l, h[4*l], h[4*l+1], h[4*l+2], h[4*l+3],
        h[4*l]/4, h[4*l]%4, h[4*l+1]/4, h[4*l+1]%4, h[4*l+2]/4, h[4*l+2]%4, h[4*l+3]/4, h[4*l+3]%4);
    return 0;
}
