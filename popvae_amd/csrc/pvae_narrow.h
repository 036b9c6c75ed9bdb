// pvae_narrow.h -- the narrow part of the VAE (everything between the two wide layers) as ONE workgroup.
//
// For widths up to 128 the hidden layers (popvae.py:395-398, 404-406), the reparameterisation (:384-389),
// the KL term (:419-421) and their backward + Adam are a few hundred KFLOP on 64 KB matrices: launch- and
// latency-bound, not throughput-bound.  Two kernels (forward, backward) walk an op table with the batch
// activations resident in LDS, instead of ~26 small launches per step.  One workgroup of 8 waves:
//   * weights go straight from global memory (L2-resident after the warm-up role) into MFMA B operands;
//     the next layer's operands are loaded while the current layer computes;
//   * A operands (activations / dpre) come from LDS with ds_read_b128, row stride 132 floats;
//   * every wave owns 16 output columns of a layer (v_mfma_f32_16x16x4_f32): no partial sums, one barrier per layer,
//     a fixed summation order (deterministic).
// Wider layers (width 256/512) use the per-layer kernels in pvae_kernels.h, which spread a layer over CUs.
#pragma once
#include "pvae_kernels.h"

#define PVN_LD 132
// The batch's 32 rows are independent through the whole chain, so the chain runs as PVN_WORKERS workgroups on as many
// CUs, each with PVN_NT 16-row tiles: half the MFMAs per layer on each CU's matrix pipe, no communication at all.
#define PVN_NT 1
#define PVN_WORKERS (2 / PVN_NT)
#define PVN_ROWS (16 * PVN_NT)
#define PVN_THREADS 512

enum { PVN_DENSE = 0, PVN_BIAS_ACT = 1, PVN_REPARAM = 2, PVN_FIRST_BIAS = 3 };

struct PvFwdOp {
    const float* W;  // operand-order copy of the kernel (pvn_wf_index below); the Keras-layout original is PvBwdOp::W
    const float* b;  // [nout]
    float* out;      // global copy of the layer output [32][nout] (needed by the backward pass)
    int kin, nout, act, kind;
};

struct PvBwdOp {
    // -- read by the chain kernel for every op (the first PVN_BWD_LEAN dwords) --
    float* Wb;           // operand-order copy of the kernel for dIn = dpre * W^T (pvn_wb_index below)
    const float* Y;      // layer output [32][nout]
    float* dpre;         // dLoss/d(pre-activation) [32][nout]: written by the chain kernel, read by pvk_narrow_adam
    int kin, nout, act, kind;
    // -- pvk_narrow_adam / FIRST_BIAS only --
    float *mW, *vW;      // MODE_GRAD: mW = gradient dump, vW unused
    float *B, *mB, *vB;
    const float* in;     // layer input  [32][kin]
    const float *gW, *gB;  // MODE_TRAIN_ACC: gradients accumulated over the earlier tiles of the batch
    float* W;            // [kin][nout] row-major (padded to multiples of 32): the Keras-layout kernel
    float* Wf;           // operand-order copy for the forward chain
};
#define PVN_BWD_LEAN 10

// Operand-order copies of a narrow kernel W[kin][nout].  The chain kernels feed v_mfma_f32_16x16x4_f32 straight from
// global memory: wave w owns 16 output columns, lane (n = lane & 15, q = lane >> 4) the contraction quarter q, and a
// "chunk" c is the 4 consecutive contraction steps 4c .. 4c + 3.  Read from the Keras layout, one wave instruction
// touches 4 (forward) or 64 (backward) different cache lines, and the CU's address unit -- not the matrix pipe --
// bounds a layer.  In these copies a wave's chunk is 1 KB contiguous (lane-major, 16 B per lane):
//   forward   Wf[((w * kin/16 + c) * 64 + lane) * 4 + e]  = W[q * kin/4 + 4c + e][16w + n]
//   backward  Wb[((w * nout/16 + c) * 64 + lane) * 4 + e] = W[16w + n][q * nout/4 + 4c + e]
// pvk_narrow_adam keeps them in step with W; pvk_narrow_pack rebuilds them after any other write to W.
__host__ __device__ inline int pvn_wf_index(int k, int col, int kin) {
    const int ks = kin >> 2, q = k / ks, rem = k - q * ks;
    return ((((col >> 4) * (ks >> 2) + (rem >> 2)) * 64) + q * 16 + (col & 15)) * 4 + (rem & 3);
}
__host__ __device__ inline int pvn_wb_index(int k, int col, int nout) {
    const int ks = nout >> 2, q = col / ks, rem = col - q * ks;
    return ((((k >> 4) * (ks >> 2) + (rem >> 2)) * 64) + q * 16 + (k & 15)) * 4 + (rem & 3);
}

// Pointers that come out of an op descriptor have lost their address space: the compiler would use FLAT loads and
// stores for them, which count on the LDS counter as well -- every wait for an LDS read would then also wait for the
// prefetched weights.  These casts put them back into the global address space.
typedef __attribute__((address_space(1))) float pvn_gf;
typedef __attribute__((address_space(1))) f32x4 pvn_gf4;
__device__ __forceinline__ const pvn_gf* pvn_g(const float* p) { return (const pvn_gf*)p; }
__device__ __forceinline__ pvn_gf* pvn_g(float* p) { return (pvn_gf*)p; }
__device__ __forceinline__ const pvn_gf4* pvn_g4(const float* p) { return (const pvn_gf4*)p; }

#define PVN_MAXOPS 64
__host__ __device__ inline size_t pvn_lds_fwd() {
    return (size_t)(2 * 32 * PVN_LD) * 4 + PVN_MAXOPS * sizeof(PvFwdOp);
}
__host__ __device__ inline size_t pvn_lds_bwd() {
    return (size_t)(2 * 32 * PVN_LD) * 4 + PVN_MAXOPS * sizeof(PvBwdOp);
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

// first ND dwords of descriptor i only (the rest of T stays zero)
template <typename T, int ND>
__device__ __forceinline__ T pvn_get_op_lean(const int* ops_s, int i) {
    constexpr int NT = sizeof(T) / 4;
    int d[NT];
#pragma unroll
    for (int k = 0; k < NT; ++k) d[k] = k < ND ? __builtin_amdgcn_readfirstlane(ops_s[i * NT + k]) : 0;
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
// One dense layer on one CU is bound by the fp32 matrix pipe: 32 x 128 x 128 MACs at 128 MAC/cycle/CU = 4096
// cycles (1.7 us) -- so everything else (operand loads, reductions, barriers) has to stay off the critical path:
//   * the OUTPUT columns are split over the 8 waves (16 columns each, v_mfma_f32_16x16x4_f32, two 16-row tiles):
//     no cross-wave reduction, no partial sums through LDS, ONE barrier per layer;
//   * the contraction index is permuted so that lane group q = lane >> 4 owns the contiguous quarter
//     [q * KS, (q + 1) * KS) of it (KS = kin / 4): the A operands of a layer are 2 x KS/4 ds_read_b128 per lane from
//     the LDS-resident activations (row stride 132 floats: 16 consecutive lanes cover all 64 banks);
//   * the next layer's B operands (<= 32 VGPRs per lane) are loaded from global memory (L2-resident after the
//     warm-up role below) while the current layer's MFMAs run; activations ping-pong between two LDS buffers.
// Accumulator layout of 16x16x4: lane (n = lane & 15, q = lane >> 4) holds rows 4q + r, r < 4, of column n.
// ---------------------------------------------------------------------------------------------
// The layer's MFMAs with the NEXT op's 32 B operands loaded in between (8 chunks of 4 spread over the KS/4
// iterations): issued in one burst ahead of the MFMAs, 8 waves x 32 loads queue up in the CU's one address unit for
// ~1 us before the first MFMA can issue.  The chunk index is clamped to the next op's own operand count (re-loading
// its last chunk) so that no load sits under a condition.  nbase / nch are wave-uniform (SGPRs): scalar address math.
// chunk c of this wave's next B operands: 1 KB per wave instruction (c is clamped to the op's own chunk count, so that no
// load sits under a condition; a wave with nothing to prefetch passes nch = 1 and any valid 1 KB)
__device__ __forceinline__ void pvn_load_chunk(PvW32& wn, int c, const float* nbase, int nch, int lane) {
    const f32x4 v = pvn_g4(nbase + 256 * min(c, nch - 1))[lane];
#pragma unroll
    for (int e = 0; e < 4; ++e) PVN_WSET(wn, 4 * c + e, v[e]);
}
template <int KS>
__device__ __forceinline__ void pvn_mma(f32x4 (&acc)[2], const float* a0, const PvW32& w, PvW32& wn, const float* nbase,
                                        int nch, int lane) {
    // acc[0], acc[1]: the two row tiles (PVN_NT == 2), or the even-k / odd-k partial sums of the one tile (PVN_NT == 1:
    // two chains keep dependent MFMAs 2 apart; the caller adds them)
    constexpr int NI = KS / 4, LI = NI - NI / 4;  // the last quarter stays free of loads: the drain follows right behind
    f32x4 x0[NI], x1[PVN_NT == 2 ? NI : 1];
#pragma unroll
    for (int q = 0; q < NI; ++q) {
        x0[q] = *reinterpret_cast<const f32x4*>(a0 + 4 * q);
        if (PVN_NT == 2) x1[q] = *reinterpret_cast<const f32x4*>(a0 + 16 * PVN_LD + 4 * q);
    }
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int q = 0; q < NI; ++q) {
        if (q < LI) {
#pragma unroll
            for (int c = q * 8 / LI; c < (q + 1) * 8 / LI; ++c) pvn_load_chunk(wn, c, nbase, nch, lane);
        }
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            if (PVN_NT == 2) {
                acc[0] = __builtin_amdgcn_mfma_f32_16x16x4f32(x0[q][e], PVN_W(w, 4 * q + e), acc[0], 0, 0, 0);
                acc[1] = __builtin_amdgcn_mfma_f32_16x16x4f32(x1[PVN_NT == 2 ? q : 0][e], PVN_W(w, 4 * q + e), acc[1], 0, 0, 0);
            } else {
                acc[e & 1] = __builtin_amdgcn_mfma_f32_16x16x4f32(x0[q][e], PVN_W(w, 4 * q + e), acc[e & 1], 0, 0, 0);
            }
        }
        __builtin_amdgcn_sched_barrier(0);
    }
    if (PVN_NT == 1) acc[0] += acc[1];
}
// the same 8 chunks without a layer to hide behind (first op, ops without MFMAs)
__device__ __forceinline__ void pvn_prefetch_plain(PvW32& wn, const float* nbase, int nch, int lane) {
#pragma unroll
    for (int c = 0; c < 8; ++c) pvn_load_chunk(wn, c, nbase, nch, lane);
}
// Wait-count discipline.  gfx9 has ONE counter for vector loads and stores, and with both kinds outstanding the compiler
// has to wait for everything at the first use of a loaded register; __syncthreads() adds a full drain per layer on top
// (its workgroup fence covers global memory).  Either would put ~1 us of store-acknowledge / load latency between
// two layers.  So per op: the next op's operands are loaded DURING this op's MFMAs, the counter is drained once right
// AFTER the MFMAs (the loads have had the whole MFMA phase, the previous op's stores a whole op), only then does the
// epilogue issue this op's stores, and the barrier between ops orders LDS only.
#define PVN_DRAIN_VMEM() __builtin_amdgcn_s_waitcnt(0x0F70) /* vmcnt(0), expcnt / lgkmcnt untouched */
__device__ __forceinline__ void pvn_lds_barrier() {
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup", "local");
    __builtin_amdgcn_s_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup", "local");

}
#define PVN_SWITCH(ks, ...)                                      \
    switch ((ks) >> 3) {                                         \
        case 1: { constexpr int S_ = 8; __VA_ARGS__; } break;    \
        case 2: { constexpr int S_ = 16; __VA_ARGS__; } break;   \
        case 3: { constexpr int S_ = 24; __VA_ARGS__; } break;   \
        default: { constexpr int S_ = 32; __VA_ARGS__; } break;  \
    }

// ---------------------------------------------------------------------------------------------
// forward.  ops[0] = BIAS_ACT (h1 = elu(P + b1)), hidden layers, heads, REPARAM (+KL), decoder layers; nops may
// stop after REPARAM (encoder.predict); op0 > 0 starts behind REPARAM with z (decoder.predict).
// ---------------------------------------------------------------------------------------------
__global__ __launch_bounds__(PVN_THREADS) void pvk_narrow_fwd(const PvFwdOp* __restrict__ ops, int op0, int nops,
                                                              const float* __restrict__ P,
                                                              const float* __restrict__ eps, int K, int Kz,
                                                              float* __restrict__ z_out, float* __restrict__ kl_out,
                                                              int nworkers) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    float* cur = smem;                   // [PVN_ROWS][PVN_LD] of the [32][PVN_LD] reserved
    float* nxt = smem + 32 * PVN_LD;
    int* ops_s = reinterpret_cast<int*>(smem + 2 * 32 * PVN_LD);
    const int tid = threadIdx.x, wave = __builtin_amdgcn_readfirstlane(tid >> 6), lane = tid & 63, n = lane & 15, q = lane >> 4;
    const int n0 = wave * 16;
    const int rb = PVN_ROWS * (int)blockIdx.x;  // first batch row of this worker
    // nworkers = 2 per 32-row tile of the device pass (1 tile, or up to 4 for a batch above 32 rows in one weight pass);
    // at most 8, so that the workers are the first eight workgroups: one per die
    if ((int)blockIdx.x >= nworkers) {
        // warm-up role: block 8 (p + 1) + w shares (by the dispatcher's round-robin placement) the XCD of worker w
        const int p = (int)(blockIdx.x >> 3) - 1;
        if (blockIdx.x >= 8 && (int)(blockIdx.x & 7) < nworkers && p >= 0 && p < nops) {
            const PvFwdOp op = ops[p];
            if (op.kind == PVN_DENSE) pvn_warm_l2(op.W, op.kin * op.nout, tid);
        }
        return;
    }
    pvn_stage_ops(ops, nops, ops_s, tid);
    __syncthreads();

    PvW32 wnext;
#pragma unroll
    for (int s = 0; s < 16; ++s) wnext.lo[s] = wnext.hi[s] = 0.f;
    // where this lane's B operands of op `o` start (element s at src[s * stride], s < ks); a wave that has
    // nothing to load for `o` gets a harmless address
    const float* nbase = P;
    int nch = 1;
    auto aim = [&](const PvFwdOp& o) {
        nbase = P; nch = 1;
        if (o.kind != PVN_DENSE || n0 >= o.nout) return;
        nch = o.kin >> 4;
        nbase = o.W + wave * nch * 256;
    };
    if (op0 > 0) {  // decoder.predict: start behind REPARAM with z (already in z_out) as the activation
        if (tid < PVN_ROWS)
            for (int k = 0; k < Kz; ++k) cur[tid * PVN_LD + k] = z_out[(rb + tid) * Kz + k];
        __syncthreads();
    }
    PvFwdOp op = pvn_get_op<PvFwdOp>(ops_s, op0);
    aim(op);
    pvn_prefetch_plain(wnext, nbase, nch, lane);
    PVN_DRAIN_VMEM();
    for (int i = op0; i < nops; ++i) {
        PVN_T(i * 8 + 0);
        const PvW32 wcur = wnext;
        // this op's bias: issued here, complete at the drain behind the MFMAs (op.b is never null)
        const float bcur = pvn_g(op.b)[n0 < op.nout ? n0 + n : 0];
        const PvFwdOp nx = i + 1 < nops ? pvn_get_op<PvFwdOp>(ops_s, i + 1) : op;
        if (i + 1 < nops) aim(nx);
        else { nbase = P; nch = 1; }
        const int N = op.nout;
        PVN_T(i * 8 + 1);
        if (op.kind == PVN_REPARAM) {
            pvn_prefetch_plain(wnext, nbase, nch, lane);
            PVN_DRAIN_VMEM();
            // cur holds the heads output: cols [0,K) = z_mean, [K,2K) = z_log_var; nxt gets z (padded to Kz)
            if (tid < PVN_ROWS) {
                const int lr = tid, b = rb + tid;  // row in LDS, row of the batch
                float acc = 0.f;
                for (int k = 0; k < Kz; ++k) {  // (cur and nxt are different buffers)
                    float zz = 0.f;
                    if (k < K) {
                        const float mu = cur[lr * PVN_LD + k], sv = cur[lr * PVN_LD + K + k];
                        const float es = expf(sv);
                        zz = mu + es * eps[b * K + k];  // exp(s), not exp(s/2): popvae.py:389
                        acc += 1.f + sv - mu * mu - es;
                    }
                    nxt[lr * PVN_LD + k] = zz;
                    z_out[b * Kz + k] = zz;
                }
                kl_out[b] = -0.5f * acc;
            }
        } else if (n0 < N) {
            f32x4 acc[2];
#pragma unroll
            for (int t = 0; t < 2; ++t)
#pragma unroll
                for (int r = 0; r < 4; ++r) acc[t][r] = 0.f;
            if (op.kind == PVN_BIAS_ACT) {
#pragma unroll
                for (int t = 0; t < PVN_NT; ++t)
#pragma unroll
                    for (int r = 0; r < 4; ++r) acc[t][r] = P[(rb + 16 * t + 4 * q + r) * N + n0 + n];
                pvn_prefetch_plain(wnext, nbase, nch, lane);
                PVN_DRAIN_VMEM();
            } else {
                const int ks = op.kin >> 2;
                const float* a0 = cur + n * PVN_LD + q * ks;
                PVN_SWITCH(ks, (pvn_mma<S_>(acc, a0, wcur, wnext, nbase, nch, lane)));
                PVN_DRAIN_VMEM();
            }
#ifdef PVN_TRACE
            asm volatile("s_nop 0" ::"v"(acc[0]), "v"(acc[1]));
            PVN_T(i * 8 + 2);
#endif
#pragma unroll
            for (int t = 0; t < PVN_NT; ++t)
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const int lr = 16 * t + 4 * q + r;
                    const float x = acc[t][r] + bcur;
                    const float y = op.act == PV_ACT_ELU ? pv_elu(x) : x;
                    nxt[lr * PVN_LD + n0 + n] = y;
                    pvn_g(op.out)[(rb + lr) * N + n0 + n] = y;
                }
        } else {
            pvn_prefetch_plain(wnext, nbase, nch, lane);
            PVN_DRAIN_VMEM();
        }
        PVN_T(i * 8 + 3);
        pvn_lds_barrier();  // nxt complete, every read of cur done
        PVN_T(i * 8 + 4);
        float* t_ = cur;
        cur = nxt;
        nxt = t_;
        op = nx;
    }
}

// ---------------------------------------------------------------------------------------------
// backward, part 1 -- the dependency chain on one CU: loss finalisation, then per layer (reverse order)
//   dpre = dY * act'(Y)  (saved to global for part 2),  dIn = dpre * W^T  -> dY of the next op.
// Ops: decoder layers D..1, REPARAM (dz -> dmu, ds), heads, encoder layers D..2, FIRST_BIAS (dpre1 + b1 Adam).
// dY of the first op = xch[0 .. 32*W0): the all-reduced decoder feedback; xch[32*W0] = summed BCE.
// Same structure as the forward: wave w owns the 16 columns [16w, 16w + 16) of dIn, and the epilogue of op i turns
// its tile of dIn straight into the NEXT op's dpre (times act'(Y), rows beyond the batch zeroed), in LDS for the
// chain and in global memory for part 2.  The weights are only READ here; part 2 (pvk_narrow_adam, one wave per
// 32x32 weight block, spread over the chip) forms dW = in^T * dpre and applies Adam.
// ---------------------------------------------------------------------------------------------
template <int MODE>
__global__ __launch_bounds__(PVN_THREADS) void pvk_narrow_bwd(const PvBwdOp* __restrict__ ops, int nops,
                                                              const float* __restrict__ xch, int W0,
                                                              const float* __restrict__ kl,
                                                              const float* __restrict__ eps,
                                                              const float* __restrict__ heads, int Kh, int K,
                                                              const PvStep* __restrict__ st, PvAdam adam,
                                                              float* __restrict__ out3, int update, int nworkers) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    float* cur = smem;
    float* nxt = smem + 32 * PVN_LD;
    int* ops_s = reinterpret_cast<int*>(smem + 2 * 32 * PVN_LD);
    const int tid = threadIdx.x, wave = __builtin_amdgcn_readfirstlane(tid >> 6), lane = tid & 63, n = lane & 15, q = lane >> 4;
    const int n0 = wave * 16;
    const int rb = PVN_ROWS * (int)blockIdx.x;  // first batch row of this worker
    if ((int)blockIdx.x >= nworkers) {
        const int p = (int)(blockIdx.x >> 3) - 1;
        if (update && blockIdx.x >= 8 && (int)(blockIdx.x & 7) < nworkers && p >= 0 && p < nops) {
            const PvBwdOp op = ops[p];
            if (op.kind == PVN_DENSE) pvn_warm_l2(op.Wb, op.kin * op.nout, tid);
            if (op.kind != PVN_REPARAM && op.act == PV_ACT_ELU) pvn_warm_l2(op.Y, 32 * op.nout, tid);
        }
        return;
    }
    if (update) pvn_stage_ops(ops, nops, ops_s, tid);
    const int nrows = st->nrows;
    const float inv_b = st->inv_b;

    if (blockIdx.x == 0 && tid == 0) {  // loss = mean_b(recon + kl) (popvae.py:423)
        float sr = xch[16 * nworkers * W0], sk = 0.f;  // the recon sum sits behind the rows of the pass
        for (int b = 0; b < nrows; ++b) sk += kl[b];
        out3[0] = (sr + sk) * inv_b;
        out3[1] = sr * inv_b;
        out3[2] = sk * inv_b;
    }
    if (!update) return;
    __syncthreads();

    PvW32 wnext;
#pragma unroll
    for (int s = 0; s < 16; ++s) wnext.lo[s] = wnext.hi[s] = 0.f;
    // B operands of op o: lane (k = n, q) of wave w needs the run W[16w + k][q * ks + s], s < ks = nout / 4
    const float* nbase = xch;
    int nch = 1;
    auto aim = [&](const PvBwdOp& o) {
        nbase = xch; nch = 1;
        if (o.kind != PVN_DENSE || n0 >= o.kin) return;
        nch = o.nout >> 4;
        nbase = o.Wb + wave * nch * 256;
    };
    PvBwdOp op = pvn_get_op_lean<PvBwdOp, PVN_BWD_LEAN>(ops_s, 0);
    aim(op);
    pvn_prefetch_plain(wnext, nbase, nch, lane);
    PVN_DRAIN_VMEM();
    {   // dpre of the first op from the exchanged decoder feedback
        if (n0 < W0) {
            float xv[PVN_NT][4], yv[PVN_NT][4];
#pragma unroll
            for (int t = 0; t < PVN_NT; ++t)
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const int b = rb + 16 * t + 4 * q + r;
                    xv[t][r] = xch[b * W0 + n0 + n];
                    yv[t][r] = pvn_g(op.Y)[b * W0 + n0 + n];
                }
#pragma unroll
            for (int t = 0; t < PVN_NT; ++t)
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const int lr = 16 * t + 4 * q + r, b = rb + lr;
                    float d = b < nrows ? xv[t][r] : 0.f;
                    if (op.act == PV_ACT_ELU) d *= pv_elu_grad_from_out(yv[t][r]);
                    cur[lr * PVN_LD + n0 + n] = d;
                    pvn_g(op.dpre)[b * W0 + n0 + n] = d;
                }
        }
    }
    __syncthreads();

    for (int i = 0; i < nops; ++i) {
        PVN_T(128 + i * 8 + 0);
        const PvW32 wb = wnext;
        // dpre1 is in global memory by now (the previous op's epilogue), for the wide encoder backward and for
        // pvk_narrow_adam, which also sums it over the batch for b1 (the rows are spread over the workers here)
        if (op.kind == PVN_FIRST_BIAS) break;
        const PvBwdOp nx = pvn_get_op_lean<PvBwdOp, PVN_BWD_LEAN>(ops_s, i + 1);  // the table always ends with FIRST_BIAS
        aim(nx);
        if (op.kind == PVN_REPARAM) {
            pvn_prefetch_plain(wnext, nbase, nch, lane);
            PVN_DRAIN_VMEM();
            // cur = dz [32][Kz]; result: dpre of the heads op [32][Kh] (cols [0,K) dmu, [K,2K) ds, rest 0)
            if (tid < PVN_ROWS) {
                const int lr = tid, b = rb + tid;  // row in LDS, row of the batch
                const bool valid = b < nrows;
                for (int k = 2 * K; k < Kh; ++k) {
                    nxt[lr * PVN_LD + k] = 0.f;
                    pvn_g(nx.dpre)[b * Kh + k] = 0.f;
                }
                for (int k = 0; k < K; ++k) {  // (cur and nxt are different buffers)
                    float dmu = 0.f, dsv = 0.f;
                    if (valid) {
                        const float mu = heads[b * Kh + k], s = heads[b * Kh + K + k];
                        const float es = expf(s), d = cur[lr * PVN_LD + k];
                        dmu = d + mu * inv_b;
                        dsv = d * eps[b * K + k] * es + 0.5f * (es - 1.f) * inv_b;
                    }
                    nxt[lr * PVN_LD + k] = dmu;
                    nxt[lr * PVN_LD + K + k] = dsv;
                    pvn_g(nx.dpre)[b * Kh + k] = dmu;
                    pvn_g(nx.dpre)[b * Kh + K + k] = dsv;
                }
            }
        } else if (n0 < op.kin) {
            const int Kin = op.kin, ks = op.nout >> 2;
            const bool to_dpre = nx.kind != PVN_REPARAM;  // REPARAM consumes the raw dz
            float yv[PVN_NT][4];
            if (to_dpre && nx.act == PV_ACT_ELU) {
#pragma unroll
                for (int t = 0; t < PVN_NT; ++t)
#pragma unroll
                    for (int r = 0; r < 4; ++r) yv[t][r] = pvn_g(nx.Y)[(rb + 16 * t + 4 * q + r) * Kin + n0 + n];
            }
            f32x4 acc[2];
#pragma unroll
            for (int t = 0; t < 2; ++t)
#pragma unroll
                for (int r = 0; r < 4; ++r) acc[t][r] = 0.f;
            const float* a0 = cur + n * PVN_LD + q * ks;
            PVN_T(128 + i * 8 + 1);
            PVN_SWITCH(ks, (pvn_mma<S_>(acc, a0, wb, wnext, nbase, nch, lane)));
#ifdef PVN_TRACE
            asm volatile("s_nop 0" ::"v"(acc[0]), "v"(acc[1]));
            PVN_T(128 + i * 8 + 5);
#endif
            PVN_DRAIN_VMEM();
            PVN_T(128 + i * 8 + 2);
#pragma unroll
            for (int t = 0; t < PVN_NT; ++t)
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const int lr = 16 * t + 4 * q + r, b = rb + lr;
                    float d = acc[t][r];
                    if (to_dpre) {
                        if (nx.act == PV_ACT_ELU) d *= pv_elu_grad_from_out(yv[t][r]);
                        if (b >= nrows) d = 0.f;
                        pvn_g(nx.dpre)[b * Kin + n0 + n] = d;
                    }
                    nxt[lr * PVN_LD + n0 + n] = d;
                }
        } else {
            pvn_prefetch_plain(wnext, nbase, nch, lane);
            PVN_DRAIN_VMEM();
        }
        PVN_T(128 + i * 8 + 3);
        pvn_lds_barrier();
        PVN_T(128 + i * 8 + 4);
        float* t_ = cur;
        cur = nxt;
        nxt = t_;
        op = nx;
    }
}

// ---------------------------------------------------------------------------------------------
// backward, part 2: weight / bias gradients of every narrow dense layer + Adam.  grid = (max blocks, n_ops), one
// wave per 32x32 block (kb, nb) of a layer's kernel:  dW[k][n] = sum_b in[b][k] dpre[b][n]  (M = k, N = n, K = b).
// All operands come straight from global memory in 128-byte rows; the accumulator layout (lane = column n,
// registers = rows k) is also the layout of the W / m / v loads and stores.
// ---------------------------------------------------------------------------------------------
// One unit of that work = one wave: (bx, by) = (32x32 block index within the layer, op index); as a device function so
// that the wide encoder backward can carry the units on its first workgroups (pvk_enc_bwd) instead of a launch of its own.
// ntiles: 32-row tiles of the device pass (the batch sum runs over all of them)
template <int MODE>
__device__ __forceinline__ void pvn_adam_unit(const PvBwdOp* __restrict__ ops, int bx, int by, int lane_,
                                              const PvStep* __restrict__ st, PvAdam adam, int ntiles) {
    const PvBwdOp op = ops[by];
    if (op.kind == PVN_FIRST_BIAS) {
        // bias of the first (wide) layer: column sums of dpre1 over the batch, 64 columns per block.  kin != 0 marks
        // the fused narrow stack; the per-layer path (widths above 128) has pvk_bias_bwd do this
        const int n = bx * 64 + lane_;
        if (!op.kin || n >= op.nout) return;
        float g = 0.f;
#pragma unroll 8
        for (int b = 0; b < 32 * ntiles; ++b) g += op.dpre[b * op.nout + n];
        if (MODE == PV_MODE_GRAD) {
            op.mB[n] = st->acc ? op.mB[n] + g : g;
        } else {
            float w_ = op.B[n], m_ = op.mB[n], v_ = op.vB[n];
            if (MODE == PV_MODE_TRAIN_ACC) g += op.gB[n];
            pv_adam(w_, m_, v_, g, st->alpha, adam);
            op.B[n] = w_;
            op.mB[n] = m_;
            op.vB[n] = v_;
        }
        return;
    }
    if (op.kind != PVN_DENSE) return;
    const int N = op.nout, Kin = op.kin;
    const int nbn = N >> 5;
    const int kb = bx / nbn, nb = bx - kb * nbn;
    if (kb * 32 >= Kin) return;
    const int lane = lane_, c = lane & 31, h = lane >> 5;
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
    f32x16 acc;
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[r] = 0.f;
    for (int t = 0; t < ntiles; ++t) {
        float a[16], b[16];
#pragma unroll
        for (int s = 0; s < 16; ++s) {
            a[s] = op.in[(32 * t + 2 * s + h) * Kin + kb * 32 + c];    // A[i = k][kk = b]
            b[s] = op.dpre[(32 * t + 2 * s + h) * N + nb * 32 + c];    // B[kk = b][j = n]
        }
#pragma unroll
        for (int s = 0; s < 16; ++s) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a[s], b[s], acc, 0, 0, 0);
    }
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
            if (op.Wf) {  // the chain kernels' operand-order copies
                const int k = kb * 32 + 4 * h + (r & 3) + 8 * (r >> 2), col = nb * 32 + c;
                op.Wf[pvn_wf_index(k, col, Kin)] = w_;
                op.Wb[pvn_wb_index(k, col, N)] = w_;
            }
        }
    }
    if (kb == 0 && h == 0) {  // bias of these 32 columns: column sums of dpre
        const int n = nb * 32 + c;
        float g = 0.f;
#pragma unroll 8
        for (int s = 0; s < 32 * ntiles; ++s) g += op.dpre[s * N + n];
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

template <int MODE>
__global__ __launch_bounds__(64) void pvk_narrow_adam(const PvBwdOp* __restrict__ ops, const PvStep* __restrict__ st,
                                                      PvAdam adam, int ntiles) {
    pvn_adam_unit<MODE>(ops, blockIdx.x, blockIdx.y, threadIdx.x, st, adam, ntiles);
}

// Per-layer path (widths above 128): the narrow layers' dW + Adam units and, as one more row of the grid, the wide first
// layer's dpre / bias update (pvk_bias_bwd) in ONE launch -- the two are independent and each was latency-bound.
template <int MODE>
__global__ __launch_bounds__(64) void pvk_narrow_adam_bias(const PvBwdOp* __restrict__ ops, const PvStep* __restrict__ st, PvAdam adam,
                                                           const float* __restrict__ Y, const float* __restrict__ dY, int N, int act,
                                                           float* __restrict__ dpre, float* __restrict__ bias,
                                                           float* __restrict__ mB, float* __restrict__ vB,
                                                           const float* __restrict__ gB, const float* __restrict__ recon,
                                                           const float* __restrict__ kl, float* __restrict__ out3, int adam_x) {
    // adam_x = (Wp / 32)^2 units per layer; the grid is wider when the last row needs more blocks (Wp = 32)
    if (blockIdx.y + 1 == gridDim.y) {
        // the step's loss scalars (pvk_loss_final) take the first block behind the bias blocks
        if (out3 && (int)blockIdx.x == (N + 63) / 64) pv_loss_final(recon, kl, st, out3);
        else pv_bias_bwd_block<MODE>(Y, dY, N, act, dpre, bias, mB, vB, st, adam, gB, blockIdx.x);  // blocks beyond N return
        return;
    }
    if ((int)blockIdx.x < adam_x) pvn_adam_unit<MODE>(ops, blockIdx.x, blockIdx.y, threadIdx.x, st, adam, 1);
}

// Rebuilds the operand-order copies from the Keras-layout kernels (after set_weights / init / restore).  grid = (16, n_ops),
// 256 threads; destination-ordered, so the writes are coalesced.
__global__ __launch_bounds__(256) void pvk_narrow_pack(const PvBwdOp* __restrict__ ops) {
    const PvBwdOp op = ops[blockIdx.y];
    if (op.kind != PVN_DENSE || !op.Wf) return;
    const int N = op.nout, Kin = op.kin, total = N * Kin;
    for (int d = blockIdx.x * 256 + threadIdx.x; d < total; d += gridDim.x * 256) {
        const int e = d & 3, lane = (d >> 2) & 63, rest = d >> 8, n = lane & 15, q = lane >> 4;
        {   // forward copy: rest = w * (Kin / 16) + c
            const int nch = Kin >> 4, w = rest / nch, c = rest - w * nch;
            op.Wf[d] = op.W[(q * (Kin >> 2) + 4 * c + e) * N + 16 * w + n];
        }
        {   // backward copy: rest = w * (N / 16) + c
            const int nch = N >> 4, w = rest / nch, c = rest - w * nch;
            op.Wb[d] = op.W[(16 * w + n) * N + q * (N >> 2) + 4 * c + e];
        }
    }
}
