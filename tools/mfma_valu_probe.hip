// Do VALU instructions of the SAME wave issue in the shadow of its own matrix instructions?  One wave per SIMD; per iteration one
// v_mfma_f32_32x32x16_bf16 (dependent accumulator chain, or two independent chains) followed by N independent v_fma_f32.
// Prints cycles per iteration.  hipcc --offload-arch=gfx950 -O3 tools/mfma_valu_probe.hip -o tools/_build/mfma_valu_probe
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
template <int N, int CHAINS>
__global__ __launch_bounds__(256) void k(float* out, unsigned long long* cyc, int iters) {
    f32x16 acc[2];
    for (int r = 0; r < 16; ++r) acc[0][r] = acc[1][r] = 0.f;
    bf16x8 a, b;
    for (int j = 0; j < 8; ++j) { a[j] = (__bf16)(float)(threadIdx.x + j); b[j] = (__bf16)(float)(j + 1); }
    float v[8];
    for (int j = 0; j < 8; ++j) v[j] = threadIdx.x * 0.001f + j;
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            acc[u % CHAINS] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, acc[u % CHAINS], 0, 0, 0);
#pragma unroll
            for (int j = 0; j < N; ++j) v[j & 7] = __builtin_fmaf(v[j & 7], 1.0001f, 0.5f);
            __builtin_amdgcn_sched_barrier(0);
        }
    }
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();
    float s = 0.f;
    for (int r = 0; r < 16; ++r) s += acc[0][r] + acc[1][r];
    for (int j = 0; j < 8; ++j) s += v[j];
    out[blockIdx.x * 256 + threadIdx.x] = s;
    if (threadIdx.x == 0 && blockIdx.x == 0) cyc[0] = t1 - t0;
}
template <int N, int CHAINS>
void run(float* out, unsigned long long* cyc) {
    const int iters = 2000;
    hipLaunchKernelGGL((k<N, CHAINS>), dim3(256), dim3(256), 0, 0, out, cyc, iters);
    hipDeviceSynchronize();
    unsigned long long c = 0;
    hipMemcpy(&c, cyc, 8, hipMemcpyDeviceToHost);
    printf("chains %d, %2d VALU per matrix instruction: %.1f cycles per (matrix instruction + VALU group)\n", CHAINS, N, (double)c / (iters * 8.0));
}
int main() {
    float* out; unsigned long long* cyc;
    if (hipMalloc(&out, 256 * 256 * 4) != hipSuccess || hipMalloc(&cyc, 8) != hipSuccess) return 1;
    run<0, 1>(out, cyc); run<4, 1>(out, cyc); run<8, 1>(out, cyc); run<12, 1>(out, cyc); run<16, 1>(out, cyc);
    run<0, 2>(out, cyc); run<4, 2>(out, cyc); run<8, 2>(out, cyc); run<16, 2>(out, cyc);
    return 0;
}
