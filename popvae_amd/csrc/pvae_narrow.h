// pvae_narrow.h -- the narrow part of the VAE (everything between the two wide layers) as chains of 4-row workers.
//
// For widths up to 128 the hidden layers (popvae.py:395-398, 404-406), the reparameterisation (:384-389),
// the KL term (:419-421) and their backward + Adam are a few hundred KFLOP on 64 KB matrices: launch- and
// latency-bound, not throughput-bound.  Two kernels (forward, backward) walk an op table with the activations
// resident in LDS, instead of ~26 small launches per step.
//
// Round 3: the batch's rows never meet inside the chain, so it is cut into workers of FOUR rows -- eight workgroups for a
// 32-row batch, one per die -- and a layer is done with v_mfma_f32_4x4x1_16B_f32 in its A-broadcast form (cbsz = 4):
// D[4 rows][64 columns] += A[4 rows][k] * B[k][64 columns], one contraction step per instruction, 8 cycles each.  With 16-row
// tiles (v_mfma_f32_16x16x4_f32, two workers, rounds 1-2) a layer was 2048 cycles of each CU's matrix pipe plus ~1.2 us of
// barrier / epilogue among eight waves; with four rows it is 1024 cycles (tools/mfma4_probe.hip: 1113 measured) on TWO
// consumer waves -- wave h owns output columns [64 h, 64 h + 64):
//   * a worker is four waves.  Waves 2 / 3 are LOADERS: while layer i is computed they copy layer i + 1's kernel (64 KB, 1 KB
//     chunks of an operand-order copy, below) from L2 into the free slot of a two-slot LDS ring -- with LDS-direct loads
//     (global_load_lds_dwordx4), nothing passes through registers -- and meet the consumers at the one barrier per layer.  Waves 0 / 1 are CONSUMERS: they read their B
//     operands from the ring slot with ds_read_b128 two 16-step groups ahead of the matrix instructions that use them, and no
//     vector-memory load sits in their instruction stream at all.  (First version of this round: each compute wave loaded the
//     next layer's column into 128 registers itself between its matrix instructions -- loads and matrix instructions issue in
//     order, so a wave stalled on the memory pipe ~1.1 us per layer whatever the interleaving; 23 -> 20 / 17 us per chain with
//     the split.  A loader that requests two layers ahead into registers was measured slower: its burst of LDS writes lands
//     where the consumers start reading.)
//   * activations live in LDS as [k][4 rows]: the A operand of 16 contraction steps is ONE conflict-free ds_read_b32 per lane
//     (lane 4 b + i holds row i of step 16 j + b; instruction b of the group broadcasts block b: abid = b), and a lane's four
//     results -- rows 0..3 of its column -- go back as one ds_write_b128;
//   * a fixed summation order (deterministic), the same on every rank of an SNP-sharded model.
// Wider layers (width 256/512) use the per-layer kernels in pvae_kernels.h, which spread a layer over CUs.
#pragma once
#include "pvae_kernels.h"

#define PVQ_ROWS 4                 // batch rows per worker
#define PVQ_THREADS 256            // four waves: 0 / 1 compute (wave h owns output columns [64 h, 64 h + 64)), 2 / 3 load kernels
#define PVQ_WORKERS_PER_TILE 8     // workers per 32-row tile of a device pass
#define PVQ_LDA (128 * PVQ_ROWS)   // floats of one activation buffer in LDS: [k < 128][row < 4]

enum { PVN_DENSE = 0, PVN_BIAS_ACT = 1, PVN_REPARAM = 2, PVN_FIRST_BIAS = 3 };

struct PvFwdOp {
    const float* W;  // operand-order copy of the kernel (pvn_wf_index below); the Keras-layout original is PvBwdOp::W
    const float* b;  // [nout]
    float* out;      // global copy of the layer output [rows][nout] (needed by the backward pass)
    int kin, nout, act, kind;
};

struct PvBwdOp {
    // -- read by the chain kernel for every op (the first PVN_BWD_LEAN dwords) --
    float* Wb;           // operand-order copy of the kernel for dIn = dpre * W^T (pvn_wb_index below)
    const float* Y;      // layer output [rows][nout]
    float* dpre;         // dLoss/d(pre-activation) [rows][nout]: written by the chain kernel, read by pvk_narrow_adam
    int kin, nout, act, kind;
    // -- pvk_narrow_adam / FIRST_BIAS only --
    float *mW, *vW;      // MODE_GRAD: mW = gradient dump, vW unused
    float *B, *mB, *vB;
    const float* in;     // layer input  [rows][kin]
    const float *gW, *gB;  // MODE_TRAIN_ACC: gradients accumulated over the earlier tiles of the batch
    float* W;            // [kin][nout] row-major (padded to multiples of 32): the Keras-layout kernel
    float* Wf;           // operand-order copy for the forward chain
};
#define PVN_BWD_LEAN 10

// Operand-order copies of a narrow kernel W[kin][nout].  Lane l of consumer wave h uses ONE output column, 64 h + l, for every
// contraction step, and a "chunk" c is the 4 consecutive contraction steps 4c .. 4c + 3 of the 64 columns of one wave.  In these
// copies a chunk is 1 KB contiguous (lane-major, 16 B per lane), so a loader wave moves it with one global_load_dwordx4 and one
// ds_write_b128, and the consumer gets four steps of its column with one ds_read_b128 from the same offsets:
//   forward   (out = in * W,     column = output unit n)  Wf[((h * kin/4  + c) * 64 + l) * 4 + e] = W[4c + e][64 h + l]
//   backward  (dIn = dpre * W^T, column = input unit k)   Wb[((h * nout/4 + c) * 64 + l) * 4 + e] = W[64 h + l][4c + e]
// Columns beyond the layer (64 h + l >= nout resp. kin) hold zeros.  pvk_narrow_adam keeps the copies in step with W;
// pvk_narrow_pack rebuilds them after any other write to W.
__host__ __device__ inline int pvn_wf_index(int k, int col, int kin) {
    return ((((col >> 6) * (kin >> 2) + (k >> 2)) * 64) + (col & 63)) * 4 + (k & 3);
}
__host__ __device__ inline int pvn_wb_index(int k, int col, int nout) {
    return ((((k >> 6) * (nout >> 2) + (col >> 2)) * 64) + (k & 63)) * 4 + (col & 3);
}
// floats of the two copies
__host__ __device__ inline size_t pvn_wf_floats(int kin, int nout) { return (size_t)((nout + 63) / 64) * 64 * kin; }
__host__ __device__ inline size_t pvn_wb_floats(int kin, int nout) { return (size_t)((kin + 63) / 64) * 64 * nout; }

// Pointers that come out of an op descriptor have lost their address space: the compiler would use FLAT loads and
// stores for them, which count on the LDS counter as well -- every wait for an LDS read would then also wait for the
// prefetched weights.  These casts put them back into the global address space.
typedef __attribute__((address_space(1))) float pvn_gf;
typedef __attribute__((address_space(1))) f32x4 pvn_gf4;
__device__ __forceinline__ const pvn_gf* pvn_g(const float* p) { return (const pvn_gf*)p; }
__device__ __forceinline__ pvn_gf* pvn_g(float* p) { return (pvn_gf*)p; }
__device__ __forceinline__ const pvn_gf4* pvn_g4(const float* p) { return (const pvn_gf4*)p; }

#define PVN_MAXOPS 64
// LDS of a worker: two activation buffers, the two-slot ring of layer kernels (64 KB each), the op table
__host__ __device__ inline size_t pvn_lds_fwd() { return (size_t)(2 * PVQ_LDA + 2 * 128 * 128) * 4 + PVN_MAXOPS * sizeof(PvFwdOp); }
// (backward: + two slots of [4 rows][128] floats for the Y rows an op's epilogue multiplies with)
__host__ __device__ inline size_t pvn_lds_bwd() {
    return (size_t)(2 * PVQ_LDA + 2 * 128 * 128 + 2 * PVQ_LDA) * 4 + PVN_MAXOPS * sizeof(PvBwdOp);
}

// The op table is cold in every launch (each step streams gigabytes through the caches in between), and a
// scalar load that misses to HBM costs ~1-2 us per layer.  So the whole table is copied into LDS with one
// round trip at kernel start; a descriptor is then read from LDS and made wave-uniform.
template <typename T>
__device__ __forceinline__ void pvn_stage_ops(const T* __restrict__ ops, int nops, int* ops_s, int tid) {
    const int n = nops * (int)(sizeof(T) / 4);
    for (int e = tid; e < n; e += PVQ_THREADS) ops_s[e] = reinterpret_cast<const int*>(ops)[e];
}
// L2 warm-up role.  The chain kernels are launched with nworkers + 8 * nops workgroups; the first nworkers (a multiple of
// 8: one per die and 32-row tile) do the work.  Workgroup nworkers + 8 p + x lands (by the dispatcher's round-robin
// placement) on die x: it pulls op p's weight matrix through that die's L2 and exits, so that the workers' operand
// loads -- 64 KB per layer, cold after the gigabytes the wide kernels streamed -- hit L2 instead of paying several HBM
// round trips per layer.  Purely a hint: placement affects speed only, never results.
__device__ __forceinline__ void pvn_warm_l2(const float* p, int nfloats, int tid) {
    f32x4 acc = {0.f, 0.f, 0.f, 0.f};
    const f32x4* q = reinterpret_cast<const f32x4*>(p);
    for (int e = tid; e < (nfloats >> 2); e += PVQ_THREADS) acc += q[e];
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

// ---- loader / consumer split ------------------------------------------------------------------------------------------
// A CU draws ~60-70 GB/s from its die's L2, so the 64 KB of a layer's kernel take ~1 us to arrive whatever the matrix pipe
// does (MI355X_MICROARCH.md "Indexed rows": 66-73 GB/s per CU from L2; measured here: a chain whose waves loaded their own B
// operands between their matrix instructions spent 1.1 us per layer in that phase with two register sets, and no less with
// three).  What can be had is that NOTHING ELSE adds to it: waves 2 and 3 of a worker do nothing but move kernels from global
// memory into a two-slot ring in LDS (op i + 1's while op i computes), waves 0 and 1 compute -- B operands by ds_read_b128
// from the ring, one two-instruction-group ahead -- and the four meet at one barrier per op.
#define PVQ_SLOT_FLOATS (128 * 128)  // one kernel copy: up to 64 chunks of 1 KB

// loader wave lw (0 / 1) copies chunks lw, lw + 2, ... of a kernel copy.  The register-staged form (PVQ_DMA 0: all loads first,
// then the LDS writes) is kept for comparison: 20.2 / 22.9 us per chain against 19.0 / 22.2 with LDS-direct loads.
// (Requesting op i + 2's kernel into registers an op ahead and writing it at the top of op i + 1 was built and measured slower,
// 1.8 us per op against 1.3: the 32 ds_write_b128 of the two loaders -- 13 cycles each through the VGPR-to-LDS path -- then
// land in one burst exactly where the consumers start reading their operands.)
#ifndef PVQ_DMA
#define PVQ_DMA 1
#endif
#if PVQ_DMA
// LDS-direct loads (global_load_lds_dwordx4: lane l's 16 bytes land at the LDS base in M0 + 16 l -- tools/lds_dma_probe.hip):
// a chunk goes from L2 into the ring without passing through registers, and the LDS writes trickle in as the data arrives
// instead of following the last load as a burst of 32 ds_write_b128 per wave (~0.25 us of a 1.3 us op).
__device__ __forceinline__ void pvq_fill_op(const float* __restrict__ src, int nchunks, float* slot, int lw, int lane) {
    typedef __attribute__((address_space(1))) const void gptr;
    typedef __attribute__((address_space(3))) void lptr;
    const float* s = src + 256 * lw + 4 * lane;
    float* d = slot + 256 * lw;
    const int n = nchunks >> 1;  // chunks per loader: contraction lengths and widths are multiples of 32, so a multiple of 4
#pragma unroll 4
    for (int i = 0; i < n; ++i) __builtin_amdgcn_global_load_lds((gptr*)(s + 512 * i), (lptr*)(d + 512 * i), 16, 0, 0);
}
// ... and the loaders' wait for everything they requested (vmcnt(0): the chunks are in LDS), once per op, before the barrier
__device__ __forceinline__ void pvq_fill_wait() { __builtin_amdgcn_s_waitcnt(0x0f70); }
#else
template <int N>
__device__ __forceinline__ void pvq_fill(const float* __restrict__ src, float* slot, int lw, int lane) {
    f32x4 v[N];
#pragma unroll
    for (int i = 0; i < N; ++i) v[i] = pvn_g4(src + 256 * (2 * i + lw))[lane];
#pragma unroll
    for (int i = 0; i < N; ++i) reinterpret_cast<f32x4*>(slot)[(2 * i + lw) * 64 + lane] = v[i];
}
__device__ __forceinline__ void pvq_fill_op(const float* __restrict__ src, int nchunks, float* slot, int lw, int lane) {
    switch (nchunks >> 1) {  // chunks per loader: 4, 8, 12, 16, 24 or 32 (contraction lengths and widths are multiples of 32)
        case 4: pvq_fill<4>(src, slot, lw, lane); break;
        case 8: pvq_fill<8>(src, slot, lw, lane); break;
        case 12: pvq_fill<12>(src, slot, lw, lane); break;
        case 16: pvq_fill<16>(src, slot, lw, lane); break;
        case 24: pvq_fill<24>(src, slot, lw, lane); break;
        default: pvq_fill<32>(src, slot, lw, lane); break;
    }
}
__device__ __forceinline__ void pvq_fill_wait() {}
#endif
// nfloats contiguous floats (a worker's four rows of a layer output Y, a multiple of 128) into an LDS area, split over the two
// loader waves like a kernel copy; lanes beyond the end take no part
__device__ __forceinline__ void pvq_fill_rows(const float* __restrict__ src, int nfloats, float* dst, int lw, int lane) {
#if PVQ_DMA
    typedef __attribute__((address_space(1))) const void gptr;
    typedef __attribute__((address_space(3))) void lptr;
    for (int c = lw; c * 256 < nfloats; c += 2) {
        if (c * 256 + 4 * lane < nfloats)
            __builtin_amdgcn_global_load_lds((gptr*)(src + c * 256 + 4 * lane), (lptr*)(dst + c * 256), 16, 0, 0);
    }
#else
    for (int c = lw; c * 256 < nfloats; c += 2)
        if (c * 256 + 4 * lane < nfloats) reinterpret_cast<f32x4*>(dst)[c * 64 + lane] = pvn_g4(src + c * 256)[lane];
#endif
}
// 16 contraction steps: instruction b takes the A operand of block b (lanes 4b .. 4b + 3: rows 0..3 of step 16 j + b) for all
// 64 columns (cbsz = 4, abid = b); four accumulators in rotation keep dependent instructions four apart
template <int B>
__device__ __forceinline__ void pvq_mm16(f32x4 (&acc)[4], float a, const f32x4 (&w)[4]) {
    if constexpr (B < 16) {
        acc[B & 3] = __builtin_amdgcn_mfma_f32_4x4x1f32(a, w[B >> 2][B & 3], acc[B & 3], 4, B, 0);
        pvq_mm16<B + 1>(acc, a, w);
    }
}
// One layer's matrix instructions for this wave's 64 columns: a_lds = activation buffer + lane ([k][4 rows]: the operand of
// steps 16 j .. 16 j + 15 is the word at 64 j + lane), w_lds = this wave's half of the ring slot + lane (chunk c at 64 c),
// KC16 = contraction length / 16.
template <int KC16>
__device__ __forceinline__ f32x4 pvq_mma(const float* a_lds, const f32x4* w_lds) {
    float a[KC16];
#pragma unroll
    for (int j = 0; j < KC16; ++j) a[j] = a_lds[64 * j];
    f32x4 acc[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) acc[i] = f32x4{0.f, 0.f, 0.f, 0.f};
    // B operands two 16-step groups (eight ds_read_b128) ahead of the matrix instructions that use them, pinned by scheduling
    // barriers: left alone the compiler keeps one read in flight and every four instructions wait for LDS (2300 cycles per
    // layer against the 1100 the matrix pipe needs)
    f32x4 w[3][4];
#pragma unroll
    for (int g = 0; g < 2; ++g)
#pragma unroll
        for (int q = 0; q < 4; ++q)
            if (g < KC16) w[g][q] = w_lds[(g * 4 + q) * 64];
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int j = 0; j < KC16; ++j) {
        if (j + 2 < KC16) {
#pragma unroll
            for (int q = 0; q < 4; ++q) w[(j + 2) % 3][q] = w_lds[((j + 2) * 4 + q) * 64];
        }
        __builtin_amdgcn_sched_barrier(0);
        pvq_mm16<0>(acc, a[j], w[j % 3]);
        __builtin_amdgcn_sched_barrier(0);
    }
    return (acc[0] + acc[1]) + (acc[2] + acc[3]);
}
__device__ __forceinline__ void pvn_lds_barrier() {
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup", "local");
    __builtin_amdgcn_s_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup", "local");
}
// contraction lengths are multiples of 32 up to 128
#define PVQ_SWITCH(kc, ...)                                       \
    switch ((kc) >> 5) {                                          \
        case 1: { constexpr int C_ = 2; __VA_ARGS__; } break;     \
        case 2: { constexpr int C_ = 4; __VA_ARGS__; } break;     \
        case 3: { constexpr int C_ = 6; __VA_ARGS__; } break;     \
        default: { constexpr int C_ = 8; __VA_ARGS__; } break;    \
    }

// ---------------------------------------------------------------------------------------------
// forward.  ops[0] = BIAS_ACT (h1 = elu(P + b1)), hidden layers, heads, REPARAM (+KL), decoder layers; nops may
// stop after REPARAM (encoder.predict); op0 > 0 starts behind REPARAM with z (decoder.predict).
// Worker w = workgroup w < nworkers handles batch rows 4 w .. 4 w + 3 of the device pass.
// ---------------------------------------------------------------------------------------------
__global__ __launch_bounds__(PVQ_THREADS) void pvk_narrow_fwd(const PvFwdOp* __restrict__ ops, int op0, int nops,
                                                              const float* __restrict__ P,
                                                              const float* __restrict__ eps, int K, int Kz,
                                                              float* __restrict__ z_out, float* __restrict__ kl_out,
                                                              int nworkers) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    float* cur = smem;                   // [k][PVQ_ROWS]
    float* nxt = smem + PVQ_LDA;
    float* ring = smem + 2 * PVQ_LDA;    // two slots of PVQ_SLOT_FLOATS
    int* ops_s = reinterpret_cast<int*>(ring + 2 * PVQ_SLOT_FLOATS);
    const int tid = threadIdx.x, wave = __builtin_amdgcn_readfirstlane(tid >> 6), lane = tid & 63;
    const bool loader = wave >= 2;
    const int h = wave & 1;                     // consumer: which 64 output columns; loader: which chunks
    const int n = h * 64 + lane;                // (consumer) this lane's output column
    const int rb = PVQ_ROWS * (int)blockIdx.x;  // first batch row of this worker
    if ((int)blockIdx.x >= nworkers) {
        const int p = ((int)blockIdx.x - nworkers) >> 3;
        if (p < nops) {
            const PvFwdOp op = ops[p];
            if (op.kind == PVN_DENSE) pvn_warm_l2(op.W, (int)pvn_wf_floats(op.kin, op.nout), tid);
        }
        return;
    }
    pvn_stage_ops(ops, nops, ops_s, tid);
    __syncthreads();
    if (op0 > 0) {  // decoder.predict: start behind REPARAM with z (already in z_out) as the activation
        for (int i = tid; i < PVQ_ROWS * Kz; i += PVQ_THREADS) cur[i] = z_out[(rb + (i & 3)) * Kz + (i >> 2)];
    }
    PvFwdOp op = pvn_get_op<PvFwdOp>(ops_s, op0);
    auto chunks = [](const PvFwdOp& o) { return (int)(pvn_wf_floats(o.kin, o.nout) >> 8); };
    if (loader) {
        if (op.kind == PVN_DENSE) pvq_fill_op(op.W, chunks(op), ring, h, lane);
        pvq_fill_wait();
    }
    pvn_lds_barrier();
    for (int i = op0; i < nops; ++i) {
        const PvFwdOp nx = i + 1 < nops ? pvn_get_op<PvFwdOp>(ops_s, i + 1) : op;
        const float* wslot = ring + ((i - op0) & 1) * PVQ_SLOT_FLOATS;
        float* nslot = ring + ((i - op0 + 1) & 1) * PVQ_SLOT_FLOATS;
        const int N = op.nout;
        if (loader) {
            // op i + 1's kernel into the other slot (free since the barrier behind op i - 1)
            if (i + 1 < nops && nx.kind == PVN_DENSE) pvq_fill_op(nx.W, chunks(nx), nslot, h, lane);
            pvq_fill_wait();
            if (op.kind == PVN_REPARAM) pvn_lds_barrier();  // the consumers' sync point inside that op
        } else if (op.kind == PVN_REPARAM) {
            PVN_T(i * 8 + 0);
            // cur holds the heads output: k in [0,K) = z_mean, [K,2K) = z_log_var; nxt gets z (padded to Kz).  One (k, row)
            // pair per thread; the KL terms meet in LDS and are summed per row in k order.
            float* klt = nxt + Kz * PVQ_ROWS;  // [K][rows], behind z: the next op (contraction length Kz) never reads it
            for (int i2 = tid; i2 < PVQ_ROWS * Kz; i2 += 128) {
                const int k = i2 >> 2, r = i2 & 3, b = rb + r;
                float zz = 0.f;
                if (k < K) {
                    const float mu = cur[k * PVQ_ROWS + r], sv = cur[(K + k) * PVQ_ROWS + r];
                    const float es = expf(sv);
                    zz = mu + es * eps[b * K + k];  // exp(s), not exp(s/2): popvae.py:389
                    klt[k * PVQ_ROWS + r] = 1.f + sv - mu * mu - es;
                }
                nxt[i2] = zz;
                z_out[b * Kz + k] = zz;
            }
            pvn_lds_barrier();
            if (tid < PVQ_ROWS) {
                float acc = 0.f;
                for (int k = 0; k < K; ++k) acc += klt[k * PVQ_ROWS + tid];
                kl_out[rb + tid] = -0.5f * acc;
            }
        } else if (h * 64 < N) {
            PVN_T(i * 8 + 0);
            // this op's bias: issued here, complete when the epilogue needs it (op.b is never null).  (Requested one op ahead by
            // the consumer, or brought into LDS by the loaders with the kernel as the backward chain's Y rows are, was measured
            // slower both times: 20.1 / 19.9 against 19.0 us per chain.)
            const float bcur = pvn_g(op.b)[n < N ? n : 0];
            f32x4 acc = {0.f, 0.f, 0.f, 0.f};
            if (op.kind == PVN_BIAS_ACT) {
                if (n < N) {
#pragma unroll
                    for (int r = 0; r < PVQ_ROWS; ++r) acc[r] = P[(rb + r) * N + n];
                }
            } else {
                const f32x4* w_lds = reinterpret_cast<const f32x4*>(wslot) + h * (op.kin >> 2) * 64 + lane;
                PVQ_SWITCH(op.kin, (acc = pvq_mma<C_>(cur + lane, w_lds)));
            }
#ifdef PVN_TRACE
            asm volatile("s_nop 0" ::"v"(acc));
            PVN_T(i * 8 + 2);
#endif
            if (n < N) {
                f32x4 y;
#pragma unroll
                for (int r = 0; r < PVQ_ROWS; ++r) {
                    const float x = acc[r] + bcur;
                    y[r] = op.act == PV_ACT_ELU ? pv_elu(x) : x;
                }
                *reinterpret_cast<f32x4*>(nxt + PVQ_ROWS * n) = y;
#pragma unroll
                for (int r = 0; r < PVQ_ROWS; ++r) pvn_g(op.out)[(rb + r) * N + n] = y[r];
            }
            PVN_T(i * 8 + 3);
        }
        pvn_lds_barrier();  // nxt complete, every read of cur and of this op's slot done, the next op's slot filled
        if (!loader) PVN_T(i * 8 + 4);
        float* t_ = cur;
        cur = nxt;
        nxt = t_;
        op = nx;
    }
}

// ---------------------------------------------------------------------------------------------
// backward, part 1 -- the dependency chain: loss finalisation, then per layer (reverse order)
//   dpre = dY * act'(Y)  (saved to global for part 2),  dIn = dpre * W^T  -> dY of the next op.
// Ops: decoder layers D..1, REPARAM (dz -> dmu, ds), heads, encoder layers D..2, FIRST_BIAS (dpre1 + b1 Adam).
// dY of the first op = xch[0 .. rows*W0): the all-reduced decoder feedback; xch[rows*W0] = summed BCE.
// Same structure as the forward: lane l of consumer wave h owns column 64 h + l of dIn for the worker's four rows, and the
// epilogue of op i turns it straight into the NEXT op's dpre (times act'(Y), rows beyond the batch zeroed), in LDS for the
// chain and in global memory for part 2.  The weights are only READ here; part 2 (pvk_narrow_adam, one wave per
// 32x32 weight block, spread over the chip) forms dW = in^T * dpre and applies Adam.
// ---------------------------------------------------------------------------------------------
template <int MODE>
__global__ __launch_bounds__(PVQ_THREADS) void pvk_narrow_bwd(const PvBwdOp* __restrict__ ops, int nops,
                                                              const float* __restrict__ xch, int W0,
                                                              const float* __restrict__ kl,
                                                              const float* __restrict__ eps,
                                                              const float* __restrict__ heads, int Kh, int K,
                                                              const PvStep* __restrict__ st, PvAdam adam,
                                                              float* __restrict__ out3, int update, int nworkers) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    float* cur = smem;
    float* nxt = smem + PVQ_LDA;
    float* ring = smem + 2 * PVQ_LDA;
    float* ys = ring + 2 * PVQ_SLOT_FLOATS;  // two slots of [4 rows][<= 128]: Y of the layer whose dpre an op's epilogue forms
    int* ops_s = reinterpret_cast<int*>(ys + 2 * PVQ_LDA);
    const int tid = threadIdx.x, wave = __builtin_amdgcn_readfirstlane(tid >> 6), lane = tid & 63;
    const bool loader = wave >= 2;
    const int h = wave & 1;
    const int n = h * 64 + lane;
    const int rb = PVQ_ROWS * (int)blockIdx.x;  // first batch row of this worker
    if ((int)blockIdx.x >= nworkers) {
        const int p = ((int)blockIdx.x - nworkers) >> 3;
        if (update && p < nops) {
            const PvBwdOp op = ops[p];
            if (op.kind == PVN_DENSE) pvn_warm_l2(op.Wb, (int)pvn_wb_floats(op.kin, op.nout), tid);
            if (op.kind != PVN_REPARAM && op.act == PV_ACT_ELU) pvn_warm_l2(op.Y, PVQ_ROWS * nworkers * op.nout, tid);
        }
        return;
    }
    if (update) pvn_stage_ops(ops, nops, ops_s, tid);
    const int nrows = st->nrows;
    const float inv_b = st->inv_b;

    if (blockIdx.x == 0 && tid == 0) {  // loss = mean_b(recon + kl) (popvae.py:423)
        float sr = xch[PVQ_ROWS * nworkers * W0], sk = 0.f;  // the recon sum sits behind the rows of the pass
        for (int b = 0; b < nrows; ++b) sk += kl[b];
        out3[0] = (sr + sk) * inv_b;
        out3[1] = sr * inv_b;
        out3[2] = sk * inv_b;
    }
    if (!update) return;
    __syncthreads();

    PvBwdOp op = pvn_get_op_lean<PvBwdOp, PVN_BWD_LEAN>(ops_s, 0);
    auto chunks = [](const PvBwdOp& o) { return (int)(pvn_wb_floats(o.kin, o.nout) >> 8); };
    // The epilogue of a dense op o multiplies with act'(Y) of its successor's layer (Y = [rows][o.kin]).  The LOADERS bring the
    // worker's four rows of it into LDS one op ahead, with the kernel (the barrier that hands over the ring slot covers them): the
    // consumers have no vector-memory load left in their stream (requested by the consumers at the top of the op, the four words
    // queued behind the loaders' 64 KB and came back after the matrix instructions were done)
    auto needs_y = [](const PvBwdOp& o, const PvBwdOp& o_next) {
        return o.kind == PVN_DENSE && o_next.kind != PVN_REPARAM && o_next.act == PV_ACT_ELU;
    };
    if (loader) {
        const PvBwdOp nx0 = pvn_get_op_lean<PvBwdOp, PVN_BWD_LEAN>(ops_s, nops > 1 ? 1 : 0);
        if (needs_y(op, nx0)) pvq_fill_rows(nx0.Y + rb * op.kin, PVQ_ROWS * op.kin, ys, h, lane);
        if (op.kind == PVN_DENSE) pvq_fill_op(op.Wb, chunks(op), ring, h, lane);
        pvq_fill_wait();
    } else if (n < W0) {  // dpre of the first op from the exchanged decoder feedback
        float xv[PVQ_ROWS], yv[PVQ_ROWS];
#pragma unroll
        for (int r = 0; r < PVQ_ROWS; ++r) {
            xv[r] = xch[(rb + r) * W0 + n];
            yv[r] = pvn_g(op.Y)[(rb + r) * W0 + n];
        }
        f32x4 dv;
#pragma unroll
        for (int r = 0; r < PVQ_ROWS; ++r) {
            float d = rb + r < nrows ? xv[r] : 0.f;
            if (op.act == PV_ACT_ELU) d *= pv_elu_grad_from_out(yv[r]);
            dv[r] = d;
            pvn_g(op.dpre)[(rb + r) * W0 + n] = d;
        }
        *reinterpret_cast<f32x4*>(cur + PVQ_ROWS * n) = dv;
    }
    pvn_lds_barrier();

    // dpre1 is in global memory once the op before FIRST_BIAS has run (its epilogue), for the wide encoder backward and
    // for pvk_narrow_adam, which also sums it over the batch for b1 (the rows are spread over the workers here)
    for (int i = 0; i < nops; ++i) {
        if (op.kind == PVN_FIRST_BIAS) break;
        const PvBwdOp nx = pvn_get_op_lean<PvBwdOp, PVN_BWD_LEAN>(ops_s, i + 1);  // the table always ends with FIRST_BIAS
        const float* wslot = ring + (i & 1) * PVQ_SLOT_FLOATS;
        float* nslot = ring + ((i + 1) & 1) * PVQ_SLOT_FLOATS;
        if (loader) {
            const PvBwdOp nx2 = pvn_get_op_lean<PvBwdOp, PVN_BWD_LEAN>(ops_s, i + 2 < nops ? i + 2 : nops - 1);
            if (needs_y(nx, nx2)) pvq_fill_rows(nx2.Y + rb * nx.kin, PVQ_ROWS * nx.kin, ys + ((i + 1) & 1) * PVQ_LDA, h, lane);
            if (nx.kind == PVN_DENSE) pvq_fill_op(nx.Wb, chunks(nx), nslot, h, lane);
            pvq_fill_wait();
        } else if (op.kind == PVN_REPARAM) {
            PVN_T(128 + i * 8 + 0);
            // cur = dz [Kz][rows]; result: dpre of the heads op [Kh][rows] (k in [0,K) dmu, [K,2K) ds, rest 0)
            for (int i2 = tid; i2 < PVQ_ROWS * Kh; i2 += 128) {  // one (k, row) pair per thread
                const int k = i2 >> 2, r = i2 & 3, b = rb + r;
                if (k >= 2 * K) {
                    nxt[i2] = 0.f;
                    pvn_g(nx.dpre)[b * Kh + k] = 0.f;
                } else if (k < K) {  // (cur and nxt are different buffers)
                    float dmu = 0.f, dsv = 0.f;
                    if (b < nrows) {
                        const float mu = heads[b * Kh + k], s = heads[b * Kh + K + k];
                        const float es = expf(s), d = cur[k * PVQ_ROWS + r];
                        dmu = d + mu * inv_b;
                        dsv = d * eps[b * K + k] * es + 0.5f * (es - 1.f) * inv_b;
                    }
                    nxt[k * PVQ_ROWS + r] = dmu;
                    nxt[(K + k) * PVQ_ROWS + r] = dsv;
                    pvn_g(nx.dpre)[b * Kh + k] = dmu;
                    pvn_g(nx.dpre)[b * Kh + K + k] = dsv;
                }
            }
        } else if (h * 64 < op.kin) {
            PVN_T(128 + i * 8 + 0);
            const int Kin = op.kin;
            const bool to_dpre = nx.kind != PVN_REPARAM;  // REPARAM consumes the raw dz
            const bool mine = n < Kin;
            const float* ysl = ys + (i & 1) * PVQ_LDA;  // [4 rows][Kin], brought in by the loaders during the previous op
            f32x4 acc;
            PVN_T(128 + i * 8 + 1);
            const f32x4* w_lds = reinterpret_cast<const f32x4*>(wslot) + h * (op.nout >> 2) * 64 + lane;
            PVQ_SWITCH(op.nout, (acc = pvq_mma<C_>(cur + lane, w_lds)));
#ifdef PVN_TRACE
            asm volatile("s_nop 0" ::"v"(acc));
            PVN_T(128 + i * 8 + 5);
            PVN_T(128 + i * 8 + 2);
#endif
            if (mine) {
                f32x4 dv;
#pragma unroll
                for (int r = 0; r < PVQ_ROWS; ++r) {
                    float d = acc[r];
                    if (to_dpre) {
                        if (nx.act == PV_ACT_ELU) d *= pv_elu_grad_from_out(ysl[r * Kin + n]);
                        if (rb + r >= nrows) d = 0.f;
                        pvn_g(nx.dpre)[(rb + r) * Kin + n] = d;
                    }
                    dv[r] = d;
                }
                *reinterpret_cast<f32x4*>(nxt + PVQ_ROWS * n) = dv;
            }
            PVN_T(128 + i * 8 + 3);
        }
        pvn_lds_barrier();
        if (!loader) PVN_T(128 + i * 8 + 4);
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
// 256 threads; destination-ordered, so the writes are coalesced.  Columns beyond the layer are written as zeros.
__global__ __launch_bounds__(256) void pvk_narrow_pack(const PvBwdOp* __restrict__ ops) {
    const PvBwdOp op = ops[blockIdx.y];
    if (op.kind != PVN_DENSE || !op.Wf) return;
    const int N = op.nout, Kin = op.kin;
    const int nf = (int)pvn_wf_floats(Kin, N), nb = (int)pvn_wb_floats(Kin, N);
    for (int d = blockIdx.x * 256 + threadIdx.x; d < nf; d += gridDim.x * 256) {
        // forward copy: d = ((h * Kin/4 + c) * 64 + l) * 4 + e  <-  W[4c + e][64 h + l]
        const int e = d & 3, l = (d >> 2) & 63, rest = d >> 8, nch = Kin >> 2, h = rest / nch, c = rest - h * nch;
        const int col = 64 * h + l;
        op.Wf[d] = col < N ? op.W[(4 * c + e) * N + col] : 0.f;
    }
    for (int d = blockIdx.x * 256 + threadIdx.x; d < nb; d += gridDim.x * 256) {
        // backward copy: d = ((h * N/4 + c) * 64 + l) * 4 + e  <-  W[64 h + l][4c + e]
        const int e = d & 3, l = (d >> 2) & 63, rest = d >> 8, nch = N >> 2, h = rest / nch, c = rest - h * nch;
        const int k = 64 * h + l;
        op.Wb[d] = k < Kin ? op.W[k * N + 4 * c + e] : 0.f;
    }
}
