// Probe of v_mfma_f32_4x4x1_16B_f32 with A broadcast (cbsz = 4, abid = b): operand / result lane maps and issue rate.
// hipcc --offload-arch=gfx950 -O3 -o tools/mfma4_probe tools/mfma4_probe.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef float f32x4 __attribute__((ext_vector_type(4)));

template <int B>
__device__ __forceinline__ f32x4 mm(float a, float b, f32x4 c) { return __builtin_amdgcn_mfma_f32_4x4x1f32(a, b, c, 4, B, 0); }

__global__ void k_map(float* out) {  // out[abid][lane][4]
    const int l = threadIdx.x;
    const float a = 1000.f * (l % 4) + (l / 4);   // block b = l / 4 holds rows i = l % 4: value 1000 i + b
    const float b = 1.f + 0.001f * l;             // distinguishes the lane (= output column)
    f32x4 z = {0, 0, 0, 0};
#define ONE(B) { f32x4 d = mm<B>(a, b, z); for (int r = 0; r < 4; ++r) out[(B * 64 + l) * 4 + r] = d[r]; }
    ONE(0) ONE(1) ONE(2) ONE(3) ONE(4) ONE(5) ONE(6) ONE(7) ONE(8) ONE(9) ONE(10) ONE(11) ONE(12) ONE(13) ONE(14) ONE(15)
#undef ONE
}

// 128 matrix instructions per "layer" (k = 0 .. 127), four accumulators in rotation, `reps` layers; one or two waves per SIMD
__global__ __launch_bounds__(512) void k_rate(float* out, unsigned long long* cyc, int reps) {
    const int l = threadIdx.x & 63;
    float w[128];
#pragma unroll
    for (int k = 0; k < 128; ++k) w[k] = 1.f + 0.0001f * (k + l);
    float a[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) a[j] = 0.5f + 0.01f * (j + l);
    f32x4 acc[4] = {{0, 0, 0, 0}, {0, 0, 0, 0}, {0, 0, 0, 0}, {0, 0, 0, 0}};
    __syncthreads();
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for (int rep = 0; rep < reps; ++rep) {
#pragma unroll
        for (int j = 0; j < 8; ++j) {
#define Q(B) acc[B & 3] = mm<B>(a[j], w[16 * j + B], acc[B & 3]);
            Q(0) Q(1) Q(2) Q(3) Q(4) Q(5) Q(6) Q(7) Q(8) Q(9) Q(10) Q(11) Q(12) Q(13) Q(14) Q(15)
#undef Q
        }
#pragma unroll
        for (int j = 0; j < 8; ++j) a[j] += acc[j & 3][0] * 1e-30f;  // dependency between the layers
    }
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();
    f32x4 s = acc[0] + acc[1] + acc[2] + acc[3];
    out[threadIdx.x] = s[0] + s[1] + s[2] + s[3];
    if (threadIdx.x == 0) cyc[0] = t1 - t0;
}

int main() {
    float* d;
    unsigned long long* c;
    hipMalloc(&d, 16 * 64 * 4 * 4);
    hipMalloc(&c, 8);
    hipLaunchKernelGGL(k_map, dim3(1), dim3(64), 0, 0, d);
    std::vector<float> h(16 * 64 * 4);
    hipMemcpy(h.data(), d, h.size() * 4, hipMemcpyDeviceToHost);
    int bad = 0;
    for (int B = 0; B < 16; ++B)
        for (int l = 0; l < 64; ++l)
            for (int r = 0; r < 4; ++r) {
                const float want = (1000.f * r + B) * (1.f + 0.001f * l);  // D[r][lane] = A(block B, row r) * B(lane)
                const float got = h[(B * 64 + l) * 4 + r];
                if (fabsf(got - want) > 1e-3f * fabsf(want) + 1e-6f) {
                    if (bad < 8) printf("abid %d lane %d reg %d: got %g want %g\n", B, l, r, got, want);
                    ++bad;
                }
            }
    printf("lane map (cbsz 4, abid b: D[reg r][lane l] = A[lane 4 b + r] * B[lane l]): %s (%d mismatches)\n", bad ? "DIFFERENT" : "confirmed", bad);
    for (int threads : {64, 128, 256, 512}) {
        const int reps = 2000;
        hipLaunchKernelGGL(k_rate, dim3(1), dim3(threads), 0, 0, d, c, reps);
        unsigned long long cy = 0;
        hipMemcpy(&cy, c, 8, hipMemcpyDeviceToHost);
        // s_memtime counts at a fixed 100 MHz on gfx9? report raw ticks per layer and let the reader compare the rows
        printf("%d waves in the workgroup: %.1f ticks per 128-instruction layer\n", threads / 64, (double)cy / reps);
    }
    return 0;
}
