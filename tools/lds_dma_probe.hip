// Where does global_load_lds_dwordx4 put a lane's 16 bytes?  One wave copies 4 chunks of 1 KB (lane l reads src + 1 KB * i + 16 l)
// with the LDS base of chunk i in M0; the kernel then dumps LDS.  Expected: LDS == source, i.e. lane l's 16 bytes land at
// base + 16 l.   hipcc --offload-arch=gfx950 -O3 tools/lds_dma_probe.hip -o tools/_build/lds_dma_probe && tools/_build/lds_dma_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
__global__ void probe(const float* __restrict__ src, float* __restrict__ out) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const int lane = threadIdx.x & 63;
    typedef __attribute__((address_space(1))) const void gptr;
    typedef __attribute__((address_space(3))) void lptr;
    for (int i = threadIdx.x; i < 1024; i += 64) smem[i] = -1.f;
    __syncthreads();
#pragma unroll
    for (int i = 0; i < 4; ++i)
        __builtin_amdgcn_global_load_lds((gptr*)(src + 256 * i + lane * 4), (lptr*)(smem + 256 * i), 16, 0, 0);
    __builtin_amdgcn_s_waitcnt(0x0f70);  // vmcnt(0)
    __syncthreads();
    for (int i = threadIdx.x; i < 1024; i += 64) out[i] = smem[i];
}
int main() {
    std::vector<float> h(1024), o(1024);
    for (int i = 0; i < 1024; ++i) h[i] = (float)i;
    float *d, *r;
    if (hipMalloc(&d, 4096) != hipSuccess || hipMalloc(&r, 4096) != hipSuccess) return 1;
    hipMemcpy(d, h.data(), 4096, hipMemcpyHostToDevice);
    hipLaunchKernelGGL(probe, dim3(1), dim3(64), 4096, 0, d, r);
    if (hipDeviceSynchronize() != hipSuccess) { printf("kernel failed\n"); return 1; }
    hipMemcpy(o.data(), r, 4096, hipMemcpyDeviceToHost);
    int bad = 0;
    for (int i = 0; i < 1024; ++i) bad += o[i] != h[i];
    printf("mismatches %d of 1024; first words: %g %g %g %g | %g %g %g %g ; word 256: %g\n", bad, o[0], o[1], o[2], o[3], o[4], o[5], o[6], o[7], o[256]);
    return bad != 0;
}
