// Ceiling finder for the wide-layer access pattern: in-place read-modify-write of 1 or 3 float arrays.
// hipcc --offload-arch=gfx950 -O3 -o tools/stream_bench tools/stream_bench.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef float f32x4 __attribute__((ext_vector_type(4)));

// A: grid-stride, one float4 per lane per iteration, NARR arrays updated in place
template <int NARR, int NT>
__global__ __launch_bounds__(256) void k_stride(f32x4* a, f32x4* b, f32x4* c, size_t n4) {
    size_t i = (size_t)blockIdx.x * 256 + threadIdx.x, st = (size_t)gridDim.x * 256;
    for (; i < n4; i += st) {
        f32x4 x = NT ? __builtin_nontemporal_load(a + i) : a[i];
        f32x4 y = {0, 0, 0, 0}, z = {0, 0, 0, 0};
        if (NARR == 3) { y = NT ? __builtin_nontemporal_load(b + i) : b[i]; z = NT ? __builtin_nontemporal_load(c + i) : c[i]; }
        x = x * 1.0001f + y * 0.001f;
        if (NARR == 3) { y = y * 0.9f + z; z = z * 0.999f + x; }
        if (NT) { __builtin_nontemporal_store(x, a + i); if (NARR == 3) { __builtin_nontemporal_store(y, b + i); __builtin_nontemporal_store(z, c + i); } }
        else { a[i] = x; if (NARR == 3) { b[i] = y; c[i] = z; } }
    }
}

// B: each wave owns contiguous chunks of CH KB (per array), UNR float4 loads per lane in flight per array
template <int NARR, int UNR, int NT>
__global__ __launch_bounds__(256) void k_chunk(f32x4* a, f32x4* b, f32x4* c, size_t nchunks /* of UNR*64 float4 */) {
    const int lane = threadIdx.x & 63;
    const size_t gw = (size_t)blockIdx.x * 4 + (threadIdx.x >> 6), nw = (size_t)gridDim.x * 4;
    const size_t per = (nchunks + nw - 1) / nw;
    size_t c0 = gw * per, c1 = c0 + per;
    if (c1 > nchunks) c1 = nchunks;
    for (size_t ch = c0; ch < c1; ++ch) {
        f32x4 x[UNR], y[UNR], z[UNR];
        f32x4* pa = a + ch * UNR * 64 + lane;
        f32x4* pb = b + ch * UNR * 64 + lane;
        f32x4* pc = c + ch * UNR * 64 + lane;
#pragma unroll
        for (int u = 0; u < UNR; ++u) {
            x[u] = NT ? __builtin_nontemporal_load(pa + u * 64) : pa[u * 64];
            if (NARR == 3) { y[u] = NT ? __builtin_nontemporal_load(pb + u * 64) : pb[u * 64]; z[u] = NT ? __builtin_nontemporal_load(pc + u * 64) : pc[u * 64]; }
        }
#pragma unroll
        for (int u = 0; u < UNR; ++u) {
            x[u] = x[u] * 1.0001f;
            if (NARR == 3) { x[u] += y[u] * 0.001f; y[u] = y[u] * 0.9f + z[u]; z[u] = z[u] * 0.999f + x[u]; }
            if (NT) { __builtin_nontemporal_store(x[u], pa + u * 64); if (NARR == 3) { __builtin_nontemporal_store(y[u], pb + u * 64); __builtin_nontemporal_store(z[u], pc + u * 64); } }
            else { pa[u * 64] = x[u]; if (NARR == 3) { pb[u * 64] = y[u]; pc[u * 64] = z[u]; } }
        }
    }
}

// B2: chunked, out-of-place, chunk assignment either contiguous per wave (RR=0) or round-robin over waves (RR=1)
template <int UNR, int RR>
__global__ __launch_bounds__(256) void k_chunk_oop(const f32x4* a, const f32x4* b, const f32x4* c, f32x4* a2, f32x4* b2, f32x4* c2,
                                                   size_t nchunks) {
    const int lane = threadIdx.x & 63;
    const size_t gw = (size_t)blockIdx.x * 4 + (threadIdx.x >> 6), nw = (size_t)gridDim.x * 4;
    const size_t per = (nchunks + nw - 1) / nw;
    for (size_t it = 0; it < per; ++it) {
        const size_t ch = RR ? it * nw + gw : gw * per + it;
        if (ch >= nchunks) break;
        f32x4 x[UNR], y[UNR], z[UNR];
        const size_t o = ch * UNR * 64 + lane;
#pragma unroll
        for (int u = 0; u < UNR; ++u) {
            x[u] = __builtin_nontemporal_load(a + o + u * 64);
            y[u] = __builtin_nontemporal_load(b + o + u * 64);
            z[u] = __builtin_nontemporal_load(c + o + u * 64);
        }
#pragma unroll
        for (int u = 0; u < UNR; ++u) {
            x[u] = x[u] * 1.0001f + y[u] * 0.001f; y[u] = y[u] * 0.9f + z[u]; z[u] = z[u] * 0.999f + x[u];
            __builtin_nontemporal_store(x[u], a2 + o + u * 64);
            __builtin_nontemporal_store(y[u], b2 + o + u * 64);
            __builtin_nontemporal_store(z[u], c2 + o + u * 64);
        }
    }
}

// B3: like B2 (round-robin) but the three arrays are interleaved chunk-wise in ONE buffer: [x chunk | y chunk | z chunk]...
template <int UNR>
__global__ __launch_bounds__(256) void k_chunk_oop_il(const f32x4* a, f32x4* a2, size_t nchunks) {
    const int lane = threadIdx.x & 63;
    const size_t gw = (size_t)blockIdx.x * 4 + (threadIdx.x >> 6), nw = (size_t)gridDim.x * 4;
    for (size_t ch = gw; ch < nchunks; ch += nw) {
        f32x4 x[UNR], y[UNR], z[UNR];
        const size_t o = ch * 3 * UNR * 64 + lane;
#pragma unroll
        for (int u = 0; u < UNR; ++u) {
            x[u] = __builtin_nontemporal_load(a + o + u * 64);
            y[u] = __builtin_nontemporal_load(a + o + (UNR + u) * 64);
            z[u] = __builtin_nontemporal_load(a + o + (2 * UNR + u) * 64);
        }
#pragma unroll
        for (int u = 0; u < UNR; ++u) {
            x[u] = x[u] * 1.0001f + y[u] * 0.001f; y[u] = y[u] * 0.9f + z[u]; z[u] = z[u] * 0.999f + x[u];
            __builtin_nontemporal_store(x[u], a2 + o + u * 64);
            __builtin_nontemporal_store(y[u], a2 + o + (UNR + u) * 64);
            __builtin_nontemporal_store(z[u], a2 + o + (2 * UNR + u) * 64);
        }
    }
}

// C: out-of-place (ping-pong) variant of A, and a read-only reduction
template <int NT>
__global__ __launch_bounds__(256) void k_oop(const f32x4* a, const f32x4* b, const f32x4* c, f32x4* a2, f32x4* b2, f32x4* c2, size_t n4) {
    size_t i = (size_t)blockIdx.x * 256 + threadIdx.x, st = (size_t)gridDim.x * 256;
    for (; i < n4; i += st) {
        f32x4 x = NT ? __builtin_nontemporal_load(a + i) : a[i];
        f32x4 y = NT ? __builtin_nontemporal_load(b + i) : b[i];
        f32x4 z = NT ? __builtin_nontemporal_load(c + i) : c[i];
        x = x * 1.0001f + y * 0.001f; y = y * 0.9f + z; z = z * 0.999f + x;
        if (NT) { __builtin_nontemporal_store(x, a2 + i); __builtin_nontemporal_store(y, b2 + i); __builtin_nontemporal_store(z, c2 + i); }
        else { a2[i] = x; b2[i] = y; c2[i] = z; }
    }
}
template <int NT>
__global__ __launch_bounds__(256) void k_read(const f32x4* a, size_t n4, float* out) {
    size_t i = (size_t)blockIdx.x * 256 + threadIdx.x, st = (size_t)gridDim.x * 256;
    f32x4 acc = {0, 0, 0, 0};
    for (; i + 3 * st < n4; i += 4 * st) {
        f32x4 x0 = NT ? __builtin_nontemporal_load(a + i) : a[i];
        f32x4 x1 = NT ? __builtin_nontemporal_load(a + i + st) : a[i + st];
        f32x4 x2 = NT ? __builtin_nontemporal_load(a + i + 2 * st) : a[i + 2 * st];
        f32x4 x3 = NT ? __builtin_nontemporal_load(a + i + 3 * st) : a[i + 3 * st];
        acc += x0 + x1 + x2 + x3;
    }
    if (acc[0] + acc[1] + acc[2] + acc[3] == 123.456f) out[0] = 1.f;
}

// D: plain copy (1 read stream + 1 write stream), the guide's reference point (float4 copy: 6.29 TB/s measured there)
template <int NT>
__global__ __launch_bounds__(256) void k_copy(const f32x4* a, f32x4* a2, size_t n4) {
    size_t i = (size_t)blockIdx.x * 256 + threadIdx.x, st = (size_t)gridDim.x * 256;
    for (; i < n4; i += st) {
        const f32x4 x = NT ? __builtin_nontemporal_load(a + i) : a[i];
        if (NT) __builtin_nontemporal_store(x, a2 + i);
        else a2[i] = x;
    }
}
// D2: copy with 4 independent float4 per lane in flight (grid-stride over 4-wide groups)
template <int NT>
__global__ __launch_bounds__(256) void k_copy4(const f32x4* a, f32x4* a2, size_t n4) {
    size_t i = (size_t)blockIdx.x * 256 + threadIdx.x, st = (size_t)gridDim.x * 256;
    for (; i + 3 * st < n4; i += 4 * st) {
        f32x4 x[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) x[u] = NT ? __builtin_nontemporal_load(a + i + u * st) : a[i + u * st];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            if (NT) __builtin_nontemporal_store(x[u], a2 + i + u * st);
            else a2[i + u * st] = x[u];
        }
    }
}

// D3: copy, each wave N KB contiguous per iteration (N loads, then N stores), chunks round-robin over the waves
template <int N>
__global__ __launch_bounds__(256) void k_copyN(const f32x4* a, f32x4* a2, size_t nchunks) {
    const int lane = threadIdx.x & 63;
    const size_t gw = (size_t)blockIdx.x * 4 + (threadIdx.x >> 6), nw = (size_t)gridDim.x * 4;
    for (size_t ch = gw; ch < nchunks; ch += nw) {
        f32x4 x[N];
        const size_t o = ch * N * 64 + lane;
#pragma unroll
        for (int u = 0; u < N; ++u) x[u] = __builtin_nontemporal_load(a + o + u * 64);
#pragma unroll
        for (int u = 0; u < N; ++u) __builtin_nontemporal_store(x[u], a2 + o + u * 64);
    }
}
// D4: as D3 with N = 3, but the three loads of a wave are 1 KB pieces of three different thirds of a 3 N2-KB tile (the
// W | m | v shape of the wide kernels: tile = [x N2 KB | y N2 KB | z N2 KB], wave w of the tile's N2 waves takes KB w of each)
template <int N2>
__global__ __launch_bounds__(256) void k_copy_tile(const f32x4* a, f32x4* a2, size_t ntiles) {
    const int lane = threadIdx.x & 63;
    const size_t gw = (size_t)blockIdx.x * 4 + (threadIdx.x >> 6), nw = (size_t)gridDim.x * 4;
    for (size_t u = gw; u < ntiles * N2; u += nw) {
        const size_t t = u / N2, w = u % N2;
        const size_t o = (t * 3 * N2 + w) * 64 + lane;
        const f32x4 x = __builtin_nontemporal_load(a + o), y = __builtin_nontemporal_load(a + o + N2 * 64),
                    z = __builtin_nontemporal_load(a + o + 2 * N2 * 64);
        __builtin_nontemporal_store(x, a2 + o);
        __builtin_nontemporal_store(y, a2 + o + N2 * 64);
        __builtin_nontemporal_store(z, a2 + o + 2 * N2 * 64);
    }
}

template <typename F>
float timeit(F f, int reps = 10) {
    hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
    f(); hipDeviceSynchronize();
    hipEventRecord(a);
    for (int i = 0; i < reps; ++i) f();
    hipEventRecord(b); hipEventSynchronize(b);
    float ms; hipEventElapsedTime(&ms, a, b);
    return ms / reps;
}

#include <algorithm>
#include <functional>
#include <string>
int main() {
    const size_t n = 128ull * 1000 * 1000;  // floats per array (512 MB), like C4's wide kernels
    float *a, *b, *c, *a2, *b2, *c2, *o;
    hipMalloc(&a, n * 4); hipMalloc(&b, n * 4); hipMalloc(&c, n * 4);
    hipMalloc(&a2, n * 4); hipMalloc(&b2, n * 4); hipMalloc(&c2, n * 4); hipMalloc(&o, 64);
    for (float* p : {a, b, c, a2, b2, c2}) hipMemset(p, 0, n * 4);
    f32x4 *A = (f32x4*)a, *B = (f32x4*)b, *C = (f32x4*)c, *A2 = (f32x4*)a2, *B2 = (f32x4*)b2, *C2 = (f32x4*)c2;
    const size_t n4 = n / 4;
    struct V { std::string name; std::function<void()> f; int narr; std::vector<float> ms; };
    std::vector<V> vs;
    for (int g : {512, 1024, 2048}) {
        vs.push_back({"inplace stride 3arr g" + std::to_string(g), [=] { hipLaunchKernelGGL((k_stride<3, 0>), dim3(g), dim3(256), 0, 0, A, B, C, n4); }, 3, {}});
        vs.push_back({"inplace stride 3arr nt g" + std::to_string(g), [=] { hipLaunchKernelGGL((k_stride<3, 1>), dim3(g), dim3(256), 0, 0, A, B, C, n4); }, 3, {}});
        vs.push_back({"oop stride 3arr g" + std::to_string(g), [=] { hipLaunchKernelGGL((k_oop<0>), dim3(g), dim3(256), 0, 0, A, B, C, A2, B2, C2, n4); }, 3, {}});
        vs.push_back({"oop stride 3arr nt g" + std::to_string(g), [=] { hipLaunchKernelGGL((k_oop<1>), dim3(g), dim3(256), 0, 0, A, B, C, A2, B2, C2, n4); }, 3, {}});
    }
    float *il, *il2;
    hipMalloc(&il, 3 * n * 4); hipMalloc(&il2, 3 * n * 4); hipMemset(il, 0, 3 * n * 4); hipMemset(il2, 0, 3 * n * 4);
    f32x4 *IL = (f32x4*)il, *IL2 = (f32x4*)il2;
    for (int g : {512, 768, 1024}) {
        vs.push_back({"oop interleaved chunk16 nt rr g" + std::to_string(g), [=] { hipLaunchKernelGGL((k_chunk_oop_il<16>), dim3(g), dim3(256), 0, 0, IL, IL2, n4 / (16 * 64)); }, 3, {}});
        vs.push_back({"oop interleaved chunk4 nt rr g" + std::to_string(g), [=] { hipLaunchKernelGGL((k_chunk_oop_il<4>), dim3(g), dim3(256), 0, 0, IL, IL2, n4 / (4 * 64)); }, 3, {}});
    }
    for (int g : {512, 1024, 2048}) {
        vs.push_back({"oop interleaved chunk1 nt rr g" + std::to_string(g), [=] { hipLaunchKernelGGL((k_chunk_oop_il<1>), dim3(g), dim3(256), 0, 0, IL, IL2, n4 / (1 * 64)); }, 3, {}});
        vs.push_back({"oop interleaved chunk2 nt rr g" + std::to_string(g), [=] { hipLaunchKernelGGL((k_chunk_oop_il<2>), dim3(g), dim3(256), 0, 0, IL, IL2, n4 / (2 * 64)); }, 3, {}});
    }
    for (int g : {768, 1280, 1536}) {
        vs.push_back({"copy 1.5GB->1.5GB nt g" + std::to_string(g), [=] { hipLaunchKernelGGL((k_copy<1>), dim3(g), dim3(256), 0, 0, IL, IL2, 3 * n4); }, -1, {}});
    }
    for (int g : {512, 768, 1024}) {
        vs.push_back({"inplace chunk16 contiguous g" + std::to_string(g), [=] { hipLaunchKernelGGL((k_chunk<3, 16, 0>), dim3(g), dim3(256), 0, 0, A, B, C, n4 / (16 * 64)); }, 3, {}});
        vs.push_back({"inplace chunk16 nt contiguous g" + std::to_string(g), [=] { hipLaunchKernelGGL((k_chunk<3, 16, 1>), dim3(g), dim3(256), 0, 0, A, B, C, n4 / (16 * 64)); }, 3, {}});
        vs.push_back({"oop chunk16 nt contiguous g" + std::to_string(g), [=] { hipLaunchKernelGGL((k_chunk_oop<16, 0>), dim3(g), dim3(256), 0, 0, A, B, C, A2, B2, C2, n4 / (16 * 64)); }, 3, {}});
        vs.push_back({"oop chunk16 nt round-robin g" + std::to_string(g), [=] { hipLaunchKernelGGL((k_chunk_oop<16, 1>), dim3(g), dim3(256), 0, 0, A, B, C, A2, B2, C2, n4 / (16 * 64)); }, 3, {}});
        vs.push_back({"oop chunk4 nt round-robin g" + std::to_string(g), [=] { hipLaunchKernelGGL((k_chunk_oop<4, 1>), dim3(g), dim3(256), 0, 0, A, B, C, A2, B2, C2, n4 / (4 * 64)); }, 3, {}});
    }
    // plain copies of the 1.5 GB interleaved buffer into the other one: 3 GB moved, the traffic of one wide training kernel at C4
    for (int g : {512, 1024, 2048, 4096}) {
        vs.push_back({"copy 1.5GB->1.5GB g" + std::to_string(g), [=] { hipLaunchKernelGGL((k_copy<0>), dim3(g), dim3(256), 0, 0, IL, IL2, 3 * n4); }, -1, {}});
        vs.push_back({"copy 1.5GB->1.5GB nt g" + std::to_string(g), [=] { hipLaunchKernelGGL((k_copy<1>), dim3(g), dim3(256), 0, 0, IL, IL2, 3 * n4); }, -1, {}});
        vs.push_back({"copy x4 1.5GB->1.5GB nt g" + std::to_string(g), [=] { hipLaunchKernelGGL((k_copy4<1>), dim3(g), dim3(256), 0, 0, IL, IL2, 3 * n4); }, -1, {}});
    }
    for (int g : {512, 1024}) {
        vs.push_back({"copyN N=1 nt g" + std::to_string(g), [=] { hipLaunchKernelGGL((k_copyN<1>), dim3(g), dim3(256), 0, 0, IL, IL2, 3 * n4 / 64); }, -1, {}});
        vs.push_back({"copyN N=2 nt g" + std::to_string(g), [=] { hipLaunchKernelGGL((k_copyN<2>), dim3(g), dim3(256), 0, 0, IL, IL2, 3 * n4 / 128); }, -1, {}});
        vs.push_back({"copyN N=3 nt g" + std::to_string(g), [=] { hipLaunchKernelGGL((k_copyN<3>), dim3(g), dim3(256), 0, 0, IL, IL2, 3 * n4 / 192); }, -1, {}});
        vs.push_back({"copyN N=4 nt g" + std::to_string(g), [=] { hipLaunchKernelGGL((k_copyN<4>), dim3(g), dim3(256), 0, 0, IL, IL2, 3 * n4 / 256); }, -1, {}});
        vs.push_back({"copyN N=12 nt g" + std::to_string(g), [=] { hipLaunchKernelGGL((k_copyN<12>), dim3(g), dim3(256), 0, 0, IL, IL2, 3 * n4 / 768); }, -1, {}});
        vs.push_back({"copy_tile 3x4KB nt g" + std::to_string(g), [=] { hipLaunchKernelGGL((k_copy_tile<4>), dim3(g), dim3(256), 0, 0, IL, IL2, 3 * n4 / (12 * 64)); }, -1, {}});
        vs.push_back({"copy_tile 3x16KB nt g" + std::to_string(g), [=] { hipLaunchKernelGGL((k_copy_tile<16>), dim3(g), dim3(256), 0, 0, IL, IL2, 3 * n4 / (48 * 64)); }, -1, {}});
    }
    vs.push_back({"hipMemcpyAsync D2D 1.5GB", [=] { hipMemcpyAsync(il2, il, 3 * n * 4, hipMemcpyDeviceToDevice, 0); }, -1, {}});
    for (int g : {512, 1024}) {
        vs.push_back({"read-only g" + std::to_string(g), [=] { hipLaunchKernelGGL((k_read<0>), dim3(g), dim3(256), 0, 0, A, n4, o); }, 0, {}});
        vs.push_back({"read-only nt g" + std::to_string(g), [=] { hipLaunchKernelGGL((k_read<1>), dim3(g), dim3(256), 0, 0, A, n4, o); }, 0, {}});
    }
    for (int round = 0; round < 7; ++round)
        for (auto& v : vs) v.ms.push_back(timeit(v.f, 3));
    for (auto& v : vs) {
        std::sort(v.ms.begin(), v.ms.end());
        const double bytes = v.narr < 0 ? 6.0 * n * 4 : v.narr ? v.narr * 2.0 * n * 4 : 1.0 * n * 4;
        printf("%-40s median %7.1f us (%5.0f GB/s)  min %7.1f us (%5.0f GB/s)  max %7.1f\n", v.name.c_str(), v.ms[3] * 1e3,
               bytes / (v.ms[3] * 1e-3) / 1e9, v.ms[0] * 1e3, bytes / (v.ms[0] * 1e-3) / 1e9, v.ms[6] * 1e3);
    }
    return 0;
}
