// pvae_narrow.h -- the narrow part of the VAE (everything between the two wide layers) as ONE workgroup.
//
// For widths up to 128 the hidden layers (popvae.py:395-398, 404-406), the reparameterisation (:384-389),
// the KL term (:419-421) and their backward + Adam are a few hundred KFLOP on 64 KB matrices: launch- and
// latency-bound, not throughput-bound.  Two kernels (forward, backward) walk an op table with the batch
// activations resident in LDS, instead of ~26 small launches per step.  One workgroup of 8 waves:
//   * weights go straight from global memory (L2-resident after the first step) into MFMA B operands;
//     the next layer's operands are loaded while the current layer computes;
//   * A operands (activations / dpre) come from LDS, row stride 129 floats (conflict-free column reads);
//   * split-K partial accumulators meet in LDS and are summed in a fixed order (deterministic).
// Wider layers (width 256/512) use the per-layer kernels in pvae_kernels.h, which spread a layer over CUs.
#pragma once
#include "pvae_kernels.h"

#define PVN_LD 129
#define PVN_THREADS 512

enum { PVN_DENSE = 0, PVN_BIAS_ACT = 1, PVN_REPARAM = 2, PVN_FIRST_BIAS = 3 };

struct PvFwdOp {
    const float* W;  // [kin][nout] row-major (padded to multiples of 32)
    const float* b;  // [nout]
    float* out;      // global copy of the layer output [32][nout] (needed by the backward pass)
    int kin, nout, act, kind;
};

struct PvBwdOp {
    float *W, *mW, *vW;  // MODE_GRAD: mW = gradient dump, vW unused
    float *B, *mB, *vB;
    const float* in;     // layer input  [32][kin]
    const float* Y;      // layer output [32][nout]
    float* dpre;         // dLoss/d(pre-activation) [32][nout]: written by the chain kernel, read by pvk_narrow_adam
    const float *gW, *gB;  // MODE_TRAIN_ACC: gradients accumulated over the earlier tiles of the batch
    int kin, nout, act, kind;
};

#define PVN_MAXOPS 64
__host__ __device__ inline size_t pvn_lds_fwd() {
    return (size_t)(32 * PVN_LD + 32 * 1024 + 128) * 4 + PVN_MAXOPS * sizeof(PvFwdOp);
}
__host__ __device__ inline size_t pvn_lds_bwd() {
    return (size_t)(2 * 32 * PVN_LD + 8 * 1024) * 4 + PVN_MAXOPS * sizeof(PvBwdOp);
}

// The op table is cold in every launch (each step streams gigabytes through the caches in between), and a
// scalar load that misses to HBM costs ~1-2 us per layer.  So the whole table is copied into LDS with one
// round trip at kernel start; a descriptor is then read from LDS and made wave-uniform.
template <typename T>
__device__ __forceinline__ void pvn_stage_ops(const T* __restrict__ ops, int nops, int* ops_s, int tid) {
    const int n = nops * (int)(sizeof(T) / 4);
    for (int e = tid; e < n; e += PVN_THREADS) ops_s[e] = reinterpret_cast<const int*>(ops)[e];
}
// L2 warm-up role.  The narrow kernels are launched with 8 * (nops + 1) workgroups; workgroup 0 does the work.
// Workgroups 8, 16, 24, ... are (by the dispatcher's round-robin placement) on the same XCD as workgroup 0 and
// share its L2: workgroup 8*(p+1) pulls op p's weight matrix through that L2 and exits, so that the worker's
// operand loads -- 64 KB per layer, cold after the gigabytes the wide kernels streamed -- hit L2 instead of
// paying several HBM round trips per layer.  Purely a hint: placement affects speed only, never results.
__device__ __forceinline__ void pvn_warm_l2(const float* p, int nfloats, int tid) {
    f32x4 acc = {0.f, 0.f, 0.f, 0.f};
    const f32x4* q = reinterpret_cast<const f32x4*>(p);
    for (int e = tid; e < (nfloats >> 2); e += PVN_THREADS) acc += q[e];
    asm volatile("" ::"v"(acc));
}

template <typename T>
__device__ __forceinline__ T pvn_get_op(const int* ops_s, int i) {
    constexpr int ND = sizeof(T) / 4;
    int d[ND];
#pragma unroll
    for (int k = 0; k < ND; ++k) d[k] = __builtin_amdgcn_readfirstlane(ops_s[i * ND + k]);
    T t;
    __builtin_memcpy(&t, d, sizeof(T));
    return t;
}

#ifdef PVN_TRACE
__device__ unsigned long long pvn_trace_buf[2048];
#define PVN_T(slot) do { if (tid == 0) { pvn_trace_buf[2*(slot)] = wall_clock64(); pvn_trace_buf[2*(slot)+1] = clock64(); } } while (0)
#else
#define PVN_T(slot) do {} while (0)
#endif

// (row b, column cc) of a 32x32 accumulator block -> index of [reg][lane] storage
__device__ __forceinline__ int pvn_frag_index(int b, int cc) {
    const int h = (b >> 2) & 1, r = (b & 3) + 4 * (b >> 3);
    return r * 64 + h * 32 + cc;
}

// Element ownership for [32][n] tiles (no integer division): thread t owns rows (t>>5) and (t>>5)+16,
// columns (t&31) + 32*j.  32 consecutive threads touch 128 contiguous bytes / 32 distinct LDS banks.

// 32 MFMA B operands of a lane as two SSA vectors (a float[32] passed by reference ends up in scratch)
struct PvW32 {
    f32x16 lo, hi;
};
#define PVN_W(w, s) ((s) < 16 ? (w).lo[(s) & 15] : (w).hi[(s) & 15])
#define PVN_WSET(w, s, val)                     \
    do {                                        \
        if ((s) < 16) (w).lo[(s) & 15] = (val); \
        else (w).hi[(s) & 15] = (val);          \
    } while (0)

// ---------------------------------------------------------------------------------------------
// forward.  One CU pulls ~64 KB of weights per layer through its own L1, which costs as much as the layer's
// MFMAs; so the loads are 16 B per lane (1 KB per wave instruction) and issued one layer ahead.
//   wave w owns the k-slice [w*K/8, (w+1)*K/8) of the contraction and ALL NB = nout/32 column blocks;
//   block t holds the columns {NB*c + t : c < 32}, so that one NB-wide vector load per lane and k-step feeds
//   NB MFMAs.  The 8 k-slice partials are summed through LDS in wave order.
// ops[0] = BIAS_ACT (h1 = elu(P + b1)), hidden layers, heads, REPARAM (+KL), decoder layers; nops may stop
// after REPARAM (encoder.predict).
// ---------------------------------------------------------------------------------------------
template <int NB>
__device__ __forceinline__ PvW32 pvn_fwd_load(PvW32 w, const float* W, int N, int k0, int steps, int c, int h) {
    // NOTE: no per-load guards.  A load under a (even wave-uniform) runtime condition makes hipcc branch around it
    // and wait vmcnt(0) per element -- 8 serial memory round trips; steps beyond `steps` re-load the last row.
#pragma unroll
    for (int s = 0; s < 8; ++s) {
        {
            const int sc = s < steps ? s : steps - 1;
            const float* src = W + (k0 + 2 * sc + h) * N + NB * c;
            if (NB == 4) {
                const f32x4 v = *reinterpret_cast<const f32x4*>(src);
#pragma unroll
                for (int t = 0; t < 4; ++t) PVN_WSET(w, s * 4 + t, v[t]);
            } else if (NB == 2) {
                const float2 v = *reinterpret_cast<const float2*>(src);
                PVN_WSET(w, s * 4, v.x);
                PVN_WSET(w, s * 4 + 1, v.y);
            } else {
#pragma unroll
                for (int t = 0; t < NB; ++t) PVN_WSET(w, s * 4 + t, src[t]);
            }
        }
    }
    return w;
}

template <int NB>
__device__ __forceinline__ void pvn_fwd_dense(const PvFwdOp& op, const PvW32& w, float* act, float* red,
                                              const float* bias_s, int tid) {
    const int wave = tid >> 6, lane = tid & 63, c = lane & 31, h = lane >> 5;
    const int row0 = tid >> 5, col0 = tid & 31;
    const int kw = op.kin >> 3, steps = kw >> 1, k0 = wave * kw;
    f32x16 acc[NB];
#pragma unroll
    for (int t = 0; t < NB; ++t)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[t][r] = 0.f;
    float a[8];
#pragma unroll
    for (int s = 0; s < 8; ++s) a[s] = act[c * PVN_LD + ((k0 + 2 * s + h) & 127)];  // s >= steps: unused
#pragma unroll
    for (int s = 0; s < 8; ++s)
        if (s < steps) {
#pragma unroll
            for (int t = 0; t < NB; ++t)
                acc[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[s], PVN_W(w, s * 4 + t), acc[t], 0, 0, 0);
        }
#pragma unroll
    for (int t = 0; t < NB; ++t)
#pragma unroll
        for (int r = 0; r < 16; ++r) red[((wave * NB + t) * 16 + r) * 64 + lane] = acc[t][r];
    __syncthreads();  // all reads of `act` and all partials are done
#pragma unroll
    for (int rr = 0; rr < 2; ++rr) {
        const int b = row0 + 16 * rr;
        const int fi = pvn_frag_index(b, col0);
#pragma unroll
        for (int t = 0; t < NB; ++t) {
            float x = red[t * 1024 + fi];
#pragma unroll
            for (int wv = 1; wv < 8; ++wv) x += red[(wv * NB + t) * 1024 + fi];
            const int n = NB * col0 + t;
            x += bias_s[n];
            const float y = op.act == PV_ACT_ELU ? pv_elu(x) : x;
            act[b * PVN_LD + n] = y;
            op.out[b * op.nout + n] = y;
        }
    }
}

__global__ __launch_bounds__(PVN_THREADS) void pvk_narrow_fwd(const PvFwdOp* __restrict__ ops, int op0, int nops,
                                                              const float* __restrict__ P,
                                                              const float* __restrict__ eps, int K, int Kz,
                                                              float* __restrict__ z_out, float* __restrict__ kl_out) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    float* act = smem;                   // [32][PVN_LD]
    float* red = smem + 32 * PVN_LD;     // [8 waves][NB][16][64]
    float* bias_s = red + 32 * 1024;     // [128]
    int* ops_s = reinterpret_cast<int*>(bias_s + 128);
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63, c = lane & 31, h = lane >> 5;
    const int row0 = tid >> 5, col0 = tid & 31;
    if (blockIdx.x != 0) {
        const int p = (int)(blockIdx.x >> 3) - 1;
        if ((blockIdx.x & 7) == 0 && p < nops) {
            const PvFwdOp op = ops[p];
            if (op.kind == PVN_DENSE) pvn_warm_l2(op.W, op.kin * op.nout, tid);
        }
        return;
    }
    pvn_stage_ops(ops, nops, ops_s, tid);
    __syncthreads();

    PvW32 wnext;
    float bnext = 0.f;
#pragma unroll
    for (int s = 0; s < 16; ++s) wnext.lo[s] = wnext.hi[s] = 0.f;
    auto prefetch = [&](int i) {
        if (i >= nops) return;
        const PvFwdOp op = pvn_get_op<PvFwdOp>(ops_s, i);
        bnext = op.b[tid < op.nout ? tid : op.nout - 1];  // op.b is never null (REPARAM carries a dummy)
        if (op.kind != PVN_DENSE) return;
        const int kw = op.kin >> 3, steps = kw >> 1, k0 = wave * kw;
        switch (op.nout >> 5) {
            case 1: wnext = pvn_fwd_load<1>(wnext, op.W, op.nout, k0, steps, c, h); break;
            case 2: wnext = pvn_fwd_load<2>(wnext, op.W, op.nout, k0, steps, c, h); break;
            case 3: wnext = pvn_fwd_load<3>(wnext, op.W, op.nout, k0, steps, c, h); break;
            default: wnext = pvn_fwd_load<4>(wnext, op.W, op.nout, k0, steps, c, h); break;
        }
    };
    if (op0 > 0) {  // decoder.predict: start behind REPARAM with z (already in z_out) as the activation
        if (tid < 32)
            for (int k = 0; k < Kz; ++k) act[tid * PVN_LD + k] = z_out[tid * Kz + k];
        __syncthreads();
    }
    prefetch(op0);
    for (int i = op0; i < nops; ++i) {
        const PvFwdOp op = pvn_get_op<PvFwdOp>(ops_s, i);
        PVN_T(i * 8 + 0);
        const PvW32 wcur = wnext;
#ifdef PVN_TRACE
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        PVN_T(i * 8 + 1);
#endif
        if (tid < 128) bias_s[tid] = bnext;  // read after this op's first barrier at the earliest
        prefetch(i + 1);
        PVN_T(i * 8 + 5);
        const int N = op.nout;
        if (op.kind == PVN_BIAS_ACT) {
            float pv[2][4];
#pragma unroll
            for (int rr = 0; rr < 2; ++rr)
#pragma unroll
                for (int j = 0; j < 4; ++j)
                    pv[rr][j] = P[(row0 + 16 * rr) * N + (j * 32 < N ? j : 0) * 32 + col0];  // clamped, never guarded
            __syncthreads();
#pragma unroll
            for (int rr = 0; rr < 2; ++rr)
#pragma unroll
                for (int j = 0; j < 4; ++j)
                    if (j * 32 < N) {
                        const int b = row0 + 16 * rr, n = j * 32 + col0;
                        const float x = pv[rr][j] + bias_s[n];
                        const float y = op.act == PV_ACT_ELU ? pv_elu(x) : x;
                        act[b * PVN_LD + n] = y;
                        op.out[b * N + n] = y;
                    }
        } else if (op.kind == PVN_REPARAM) {
            // act holds the heads output: cols [0,K) = z_mean, [K,2K) = z_log_var; replaced by z (padded to Kz)
            if (tid < 32) {
                const int b = tid;
                float mu[16], sv[16];
#pragma unroll
                for (int k = 0; k < 16; ++k) {
                    mu[k] = k < K ? act[b * PVN_LD + k] : 0.f;
                    sv[k] = k < K ? act[b * PVN_LD + K + k] : 0.f;
                }
                float acc = 0.f;
                for (int k = 0; k < Kz; ++k) {
                    act[b * PVN_LD + k] = 0.f;
                    z_out[b * Kz + k] = 0.f;
                }
#pragma unroll
                for (int k = 0; k < 16; ++k)
                    if (k < K) {
                        const float es = expf(sv[k]);
                        const float zz = mu[k] + es * eps[b * K + k];  // exp(s), not exp(s/2): popvae.py:389
                        acc += 1.f + sv[k] - mu[k] * mu[k] - es;
                        act[b * PVN_LD + k] = zz;
                        z_out[b * Kz + k] = zz;
                    }
                kl_out[b] = -0.5f * acc;
            }
        } else {
            PVN_T(i * 8 + 2);
            switch (N >> 5) {
                case 1: pvn_fwd_dense<1>(op, wcur, act, red, bias_s, tid); break;
                case 2: pvn_fwd_dense<2>(op, wcur, act, red, bias_s, tid); break;
                case 3: pvn_fwd_dense<3>(op, wcur, act, red, bias_s, tid); break;
                default: pvn_fwd_dense<4>(op, wcur, act, red, bias_s, tid); break;
            }
        }
        __syncthreads();
    }
}

// ---------------------------------------------------------------------------------------------
// backward, part 1 -- the dependency chain on one CU: loss finalisation, then per layer (reverse order)
//   dpre = dY * act'(Y)  (saved to global for part 2),  dIn = dpre * W^T  -> dY of the next op.
// Ops: decoder layers D..1, REPARAM (dz -> dmu, ds), heads, encoder layers D..2, FIRST_BIAS (dpre1 + b1 Adam).
// dY of the first op = xch[0 .. 32*W0): the all-reduced decoder feedback; xch[32*W0] = summed BCE.
// The weights are only READ here; part 2 (pvk_narrow_adam, one wave per 32x32 weight block, spread over the
// chip) forms dW = in^T * dpre and applies Adam.  Splitting it this way keeps the 6 x 64 KB per layer of
// W/m/v traffic off the single CU that walks the chain.
// ---------------------------------------------------------------------------------------------
template <int STEPS>
__device__ __forceinline__ PvW32 pvn_load_contig(PvW32 w, const float* src) {
#pragma unroll
    for (int q = 0; q < STEPS / 4; ++q) {
        const f32x4 v = reinterpret_cast<const f32x4*>(src)[q];
#pragma unroll
        for (int e = 0; e < 4; ++e) PVN_WSET(w, 4 * q + e, v[e]);
    }
    return w;
}
template <int STEPS>
__device__ __forceinline__ f32x16 pvn_mma(f32x16 acc, const float* a, const PvW32& w) {
    float av[STEPS];
#pragma unroll
    for (int s = 0; s < STEPS; ++s) av[s] = a[s];
#pragma unroll
    for (int s = 0; s < STEPS; ++s) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(av[s], PVN_W(w, s), acc, 0, 0, 0);
    return acc;
}
#define PVN_SWITCH(steps, ...)                                   \
    switch ((steps) >> 3) {                                      \
        case 1: { constexpr int S_ = 8; __VA_ARGS__; } break;    \
        case 2: { constexpr int S_ = 16; __VA_ARGS__; } break;   \
        case 3: { constexpr int S_ = 24; __VA_ARGS__; } break;   \
        default: { constexpr int S_ = 32; __VA_ARGS__; } break;  \
    }

template <int MODE>
__global__ __launch_bounds__(PVN_THREADS) void pvk_narrow_bwd(const PvBwdOp* __restrict__ ops, int nops,
                                                              const float* __restrict__ xch, int W0,
                                                              const float* __restrict__ kl,
                                                              const float* __restrict__ eps,
                                                              const float* __restrict__ heads, int Kh, int K,
                                                              const PvStep* __restrict__ st, PvAdam adam,
                                                              float* __restrict__ out3, int update) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    float* dY_s = smem;
    float* dp_s = smem + 32 * PVN_LD;
    float* red = smem + 2 * 32 * PVN_LD;  // [8 waves][16][64]
    int* ops_s = reinterpret_cast<int*>(red + 8 * 1024);
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63, c = lane & 31, h = lane >> 5;
    const int row0 = tid >> 5, col0 = tid & 31;
    if (blockIdx.x != 0) {
        const int p = (int)(blockIdx.x >> 3) - 1;
        if (update && (blockIdx.x & 7) == 0 && p < nops) {
            const PvBwdOp op = ops[p];
            if (op.kind == PVN_DENSE) pvn_warm_l2(op.W, op.kin * op.nout, tid);
            if (op.kind != PVN_REPARAM && op.act == PV_ACT_ELU) pvn_warm_l2(op.Y, 32 * op.nout, tid);
        }
        return;
    }
    if (update) pvn_stage_ops(ops, nops, ops_s, tid);
    const int nrows = st->nrows;
    const float inv_b = st->inv_b, alpha = st->alpha;

    if (tid == 0) {  // loss = mean_b(recon + kl) (popvae.py:423)
        float sr = xch[32 * W0], sk = 0.f;
        for (int b = 0; b < nrows; ++b) sk += kl[b];
        out3[0] = (sr + sk) * inv_b;
        out3[1] = sr * inv_b;
        out3[2] = sk * inv_b;
    }
    if (!update) return;
    __syncthreads();

    // dIn[b][k] = sum_n dpre[b][n] W[k][n]: wave (kb, nq) owns k-block kb and half nq of the n range;
    // lane (c, h) feeds contraction indices n0 + h*half + s, s < half (a contiguous run of W row kb*32 + c).
    const int kb = wave >> 1, nq = wave & 1;
    PvW32 wnext;
    float ynext[2][4];
#pragma unroll
    for (int s = 0; s < 16; ++s) wnext.lo[s] = wnext.hi[s] = 0.f;
    auto prefetch = [&](int i) {
        if (i >= nops) return;
        const PvBwdOp op = pvn_get_op<PvBwdOp>(ops_s, i);
        if (op.kind == PVN_REPARAM) return;
        const int N = op.nout;
        if (op.act == PV_ACT_ELU) {
#pragma unroll
            for (int rr = 0; rr < 2; ++rr)
#pragma unroll
                for (int j = 0; j < 4; ++j)
                    ynext[rr][j] = op.Y[(row0 + 16 * rr) * N + (j * 32 < N ? j : 0) * 32 + col0];  // clamped
        }
        if (op.kind != PVN_DENSE || kb * 32 >= op.kin) return;
        const int chunk = N >> 1, half = chunk >> 1;
        const float* src = op.W + (kb * 32 + c) * N + nq * chunk + h * half;
        PVN_SWITCH(half, wnext = pvn_load_contig<S_>(wnext, src));
    };
    prefetch(0);
    {
        float xv[2][4];
#pragma unroll
        for (int rr = 0; rr < 2; ++rr)
#pragma unroll
            for (int j = 0; j < 4; ++j) xv[rr][j] = xch[(row0 + 16 * rr) * W0 + (j * 32 < W0 ? j : 0) * 32 + col0];
#pragma unroll
        for (int rr = 0; rr < 2; ++rr)
#pragma unroll
            for (int j = 0; j < 4; ++j)
                if (j * 32 < W0) dY_s[(row0 + 16 * rr) * PVN_LD + j * 32 + col0] = xv[rr][j];
    }
    __syncthreads();

    for (int i = 0; i < nops; ++i) {
        const PvBwdOp op = pvn_get_op<PvBwdOp>(ops_s, i);
        const int N = op.nout, Kin = op.kin;
        const PvW32 wb = wnext;
        float yv[2][4];
#pragma unroll
        for (int rr = 0; rr < 2; ++rr)
#pragma unroll
            for (int j = 0; j < 4; ++j) yv[rr][j] = ynext[rr][j];
        prefetch(i + 1);
        if (op.kind == PVN_REPARAM) {
            // dY_s = dz [32][..]; result: d heads [32][Kh] (cols [0,K) dmu, [K,2K) ds, rest 0)
            if (tid < 32) {
                const int b = tid;
                const bool valid = b < nrows;
                float dmu[16], dsv[16];
#pragma unroll
                for (int k = 0; k < 16; ++k) {
                    dmu[k] = 0.f;
                    dsv[k] = 0.f;
                    if (k < K && valid) {
                        const float mu = heads[b * Kh + k], s = heads[b * Kh + K + k];
                        const float es = expf(s), d = dY_s[b * PVN_LD + k];
                        dmu[k] = d + mu * inv_b;
                        dsv[k] = d * eps[b * K + k] * es + 0.5f * (es - 1.f) * inv_b;
                    }
                }
                for (int k = 0; k < Kh; ++k) dY_s[b * PVN_LD + k] = 0.f;  // own row only
#pragma unroll
                for (int k = 0; k < 16; ++k)
                    if (k < K) {
                        dY_s[b * PVN_LD + k] = dmu[k];
                        dY_s[b * PVN_LD + K + k] = dsv[k];
                    }
            }
            __syncthreads();
            continue;
        }
        // dpre = dY * act'(Y), rows beyond the batch zeroed; kept in LDS and written out for part 2
#pragma unroll
        for (int rr = 0; rr < 2; ++rr)
#pragma unroll
            for (int j = 0; j < 4; ++j)
                if (j * 32 < N) {
                    const int b = row0 + 16 * rr, n = j * 32 + col0;
                    float d = b < nrows ? dY_s[b * PVN_LD + n] : 0.f;
                    if (op.act == PV_ACT_ELU) d *= pv_elu_grad_from_out(yv[rr][j]);
                    dp_s[b * PVN_LD + n] = d;
                    op.dpre[b * N + n] = d;
                }
        __syncthreads();
        if (op.kind == PVN_FIRST_BIAS) {
            // dpre1 is what the wide encoder backward consumes; b1 gets its Adam step here
            if (tid < N) {
                float g = 0.f;
#pragma unroll 8
                for (int b = 0; b < 32; ++b) g += dp_s[b * PVN_LD + tid];
                if (MODE == PV_MODE_GRAD) {
                    op.mB[tid] = st->acc ? op.mB[tid] + g : g;
                } else {
                    float w_ = op.B[tid], m_ = op.mB[tid], v_ = op.vB[tid];
                    if (MODE == PV_MODE_TRAIN_ACC) g += op.gB[tid];
                    pv_adam(w_, m_, v_, g, alpha, adam);
                    op.B[tid] = w_;
                    op.mB[tid] = m_;
                    op.vB[tid] = v_;
                }
            }
            continue;
        }
        const int chunk = N >> 1, half = chunk >> 1;
        if (kb * 32 < Kin) {
            f32x16 acc;
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[r] = 0.f;
            const float* a = dp_s + c * PVN_LD + nq * chunk + h * half;
            PVN_SWITCH(half, acc = (pvn_mma<S_>(acc, a, wb)));
#pragma unroll
            for (int r = 0; r < 16; ++r) red[(wave * 16 + r) * 64 + lane] = acc[r];
        }
        __syncthreads();
        // dIn = sum of the two n-halves -> dY of the next op
#pragma unroll
        for (int rr = 0; rr < 2; ++rr) {
            const int b = row0 + 16 * rr;
            const int fi = pvn_frag_index(b, col0);
#pragma unroll
            for (int j = 0; j < 4; ++j)
                if (j * 32 < Kin) dY_s[b * PVN_LD + j * 32 + col0] = red[(j * 2) * 1024 + fi] + red[(j * 2 + 1) * 1024 + fi];
        }
        __syncthreads();
    }
}

// ---------------------------------------------------------------------------------------------
// backward, part 2: weight / bias gradients of every narrow dense layer + Adam.  grid = (max blocks, n_ops), one
// wave per 32x32 block (kb, nb) of a layer's kernel:  dW[k][n] = sum_b in[b][k] dpre[b][n]  (M = k, N = n, K = b).
// All operands come straight from global memory in 128-byte rows; the accumulator layout (lane = column n,
// registers = rows k) is also the layout of the W / m / v loads and stores.
// ---------------------------------------------------------------------------------------------
template <int MODE>
__global__ __launch_bounds__(64) void pvk_narrow_adam(const PvBwdOp* __restrict__ ops, const PvStep* __restrict__ st,
                                                      PvAdam adam) {
    const PvBwdOp op = ops[blockIdx.y];
    if (op.kind != PVN_DENSE) return;
    const int N = op.nout, Kin = op.kin;
    const int nbn = N >> 5;
    const int kb = blockIdx.x / nbn, nb = blockIdx.x - kb * nbn;
    if (kb * 32 >= Kin) return;
    const int lane = threadIdx.x, c = lane & 31, h = lane >> 5;
    const float alpha = st->alpha;
    const int voff = (kb * 32 + 4 * h) * N + nb * 32 + c;
    float wv[16], mv[16], vv[16];
    if (MODE != PV_MODE_GRAD) {
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int ro = ((r & 3) + 8 * (r >> 2)) * N;
            wv[r] = (op.W + ro)[voff];
            mv[r] = (op.mW + ro)[voff];
            vv[r] = (op.vW + ro)[voff];
        }
    }
    float a[16], b[16];
#pragma unroll
    for (int s = 0; s < 16; ++s) {
        a[s] = op.in[(2 * s + h) * Kin + kb * 32 + c];    // A[i = k][kk = b]
        b[s] = op.dpre[(2 * s + h) * N + nb * 32 + c];    // B[kk = b][j = n]
    }
    f32x16 acc;
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[r] = 0.f;
#pragma unroll
    for (int s = 0; s < 16; ++s) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a[s], b[s], acc, 0, 0, 0);
#pragma unroll
    for (int r = 0; r < 16; ++r) {
        const int ro = ((r & 3) + 8 * (r >> 2)) * N;
        if (MODE == PV_MODE_GRAD) {
            (op.mW + ro)[voff] = st->acc ? (op.mW + ro)[voff] + acc[r] : acc[r];
        } else {
            float w_ = wv[r], m_ = mv[r], v_ = vv[r];
            pv_adam(w_, m_, v_, MODE == PV_MODE_TRAIN_ACC ? acc[r] + (op.gW + ro)[voff] : acc[r], alpha, adam);
            (op.W + ro)[voff] = w_;
            (op.mW + ro)[voff] = m_;
            (op.vW + ro)[voff] = v_;
        }
    }
    if (kb == 0 && h == 0) {  // bias of these 32 columns: column sums of dpre
        const int n = nb * 32 + c;
        float g = 0.f;
#pragma unroll
        for (int s = 0; s < 32; ++s) g += op.dpre[s * N + n];
        if (MODE == PV_MODE_GRAD) {
            op.mB[n] = st->acc ? op.mB[n] + g : g;
        } else {
            float w_ = op.B[n], m_ = op.mB[n], v_ = op.vB[n];
            if (MODE == PV_MODE_TRAIN_ACC) g += op.gB[n];
            pv_adam(w_, m_, v_, g, alpha, adam);
            op.B[n] = w_;
            op.mB[n] = m_;
            op.vB[n] = v_;
        }
    }
}
