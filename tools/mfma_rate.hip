// One workgroup on one CU: cycles per f32-input MFMA as a function of the instruction shape, the number of waves and the
// number of independent accumulator chains per wave.  hipcc --offload-arch=gfx950 -O3 -o tools/_build/mfma_rate tools/mfma_rate.hip
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

template <int SHAPE, int CHAINS>
__global__ void k(float* out, unsigned long long* cyc, int iters) {
    const float a = threadIdx.x * 0.001f, b = 1.0f + threadIdx.x * 0.002f;
    unsigned long long t0 = 0, t1 = 0;
    if (SHAPE == 16) {
        f32x4 acc[CHAINS];
        for (int c = 0; c < CHAINS; ++c) acc[c] = f32x4{0.f, 0.f, 0.f, 0.f};
        __syncthreads();
        t0 = __builtin_amdgcn_s_memtime();
        for (int i = 0; i < iters; ++i)
#pragma unroll
            for (int c = 0; c < CHAINS; ++c) acc[c] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, acc[c], 0, 0, 0);
        float s = 0.f;
        for (int c = 0; c < CHAINS; ++c) s += acc[c][0] + acc[c][3];
        asm volatile("" ::"v"(s));
        t1 = __builtin_amdgcn_s_memtime();
        out[blockIdx.x * blockDim.x + threadIdx.x] = s;
    } else {
        f32x16 acc[CHAINS];
        for (int c = 0; c < CHAINS; ++c)
            for (int r = 0; r < 16; ++r) acc[c][r] = 0.f;
        __syncthreads();
        t0 = __builtin_amdgcn_s_memtime();
        for (int i = 0; i < iters; ++i)
#pragma unroll
            for (int c = 0; c < CHAINS; ++c) acc[c] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc[c], 0, 0, 0);
        float s = 0.f;
        for (int c = 0; c < CHAINS; ++c) s += acc[c][0] + acc[c][15];
        asm volatile("" ::"v"(s));
        t1 = __builtin_amdgcn_s_memtime();
        out[blockIdx.x * blockDim.x + threadIdx.x] = s;
    }
    if ((threadIdx.x & 63) == 0) cyc[threadIdx.x >> 6] = t1 - t0;
}

template <int SHAPE, int CHAINS>
void run(int threads, float* out, unsigned long long* cyc) {
    const int iters = 2000;
    k<SHAPE, CHAINS><<<1, threads>>>(out, cyc, iters);
    k<SHAPE, CHAINS><<<1, threads>>>(out, cyc, iters);
    hipDeviceSynchronize();
    unsigned long long h[16];
    hipMemcpy(h, cyc, sizeof h, hipMemcpyDeviceToHost);
    unsigned long long mx = 0;
    for (int w = 0; w < threads / 64; ++w) mx = h[w] > mx ? h[w] : mx;
    const double per_wave = (double)mx / (iters * CHAINS);
    const double macs = SHAPE == 16 ? 1024.0 : 2048.0;
    printf("shape %2d  chains %d  waves %2d : %6.1f cyc per MFMA per wave, %6.1f MAC/cyc/CU\n", SHAPE, CHAINS, threads / 64,
           per_wave, macs * (threads / 64) / per_wave);
}

int main() {
    float* out;
    unsigned long long* cyc;
    hipMalloc(&out, 1024 * 4);
    hipMalloc(&cyc, 16 * 8);
    for (int threads : {64, 256, 512, 576, 1024}) {
        run<16, 1>(threads, out, cyc);
        run<16, 2>(threads, out, cyc);
        run<16, 4>(threads, out, cyc);
        run<32, 1>(threads, out, cyc);
        run<32, 2>(threads, out, cyc);
    }
    return 0;
}
