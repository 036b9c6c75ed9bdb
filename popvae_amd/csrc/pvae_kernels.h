// pvae_kernels.h -- hand-written gfx950 (CDNA4, MI355X) kernels for the popVAE training hot path.
//
// Everything here is fp32 and uses the exact f32-input matrix instruction v_mfma_f32_32x32x2_f32
// (bitwise a k-ordered fmaf chain), so results meet the 1e-4 fp32 parity bar against the oracle.
//
// Geometry shared by all kernels
//   * a batch tile is 32 samples (rows b = 0..31; rows >= nrows are padding and masked out);
//   * the SNP axis is cut into blocks of 32 SNPs ("blk"); widths are padded to a multiple of 32 and cut
//     into 32-wide "wb" blocks; a wave works on a *subtile* = one SNP block x NWB width blocks (NWB <= 4,
//     i.e. <= 128 units); wider layers are split into ngrp = Wp/128 groups of 128 units;
//   * MFMA 32x32x2 lane maps (lane = 32*h + c):  A[i = c][k = h],  B[k = h][j = c],
//     C/D reg r <-> row ROW(r,h) = (r&3) + 8*(r>>2) + 4*h, col c.
//   * the two wide weight matrices (and their Adam moments) are stored in HBM in *fragment-native*
//     order so that one dwordx4 load per lane per 4 accumulator registers streams a subtile with
//     perfectly coalesced 1 KiB wave accesses:
//        float index of element (blk, grp, wbl, r, lane) =
//            ((blk*ngrp + grp)*NWB + wbl)*1024 + ((r>>2)*64 + lane)*4 + (r&3)
//     encoder W1 (Keras [L][W]):  (row ROW(r,h) -> SNP blk*32+ROW, col c -> unit (grp*NWB+wbl)*32 + c)
//     decoder V  (Keras [W][L]):  (row ROW(r,h) -> unit (grp*NWB+wbl)*32 + ROW, col c -> SNP blk*32 + c)
//   * genotypes: resident matrix packed 2 bits per call, sample-major rows (code = derived count,
//     value = code/2, SNP l at bits 2*(l%4) of byte l/4 -> one little-endian u64 per (sample, blk));
//     per batch a transposed copy Xt[blk*32 + l] = u64 with sample b's code at bits 2b.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include <type_traits>

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

// -DPV_WTRACE: per-workgroup timeline of the wide kernels (100 MHz wall clock): [kernel 0 = dec, 1 = enc_bwd][workgroup][point]
#ifdef PV_WTRACE
__device__ unsigned long long pv_wtrace_buf[2 * 1024 * 6];  // points 0..3 = times, 4 = HW_ID register, 5 = XCC_ID register
#define PV_WT(k, pt) do { if (threadIdx.x == 0 && blockIdx.x < 1024) { unsigned long long* p_ = pv_wtrace_buf + ((k) * 1024 + blockIdx.x) * 6; \
    p_[pt] = wall_clock64(); if ((pt) == 0) { p_[4] = __builtin_amdgcn_s_getreg(63492); p_[5] = __builtin_amdgcn_s_getreg(63508); } } } while (0)
#else
#define PV_WT(k, pt) do {} while (0)
#endif
// -DPV_DEC3_PROF (with -DPV_WTRACE): shader-clock cycles per phase of a pvk_dec3 visit, wave 0 of the first 256 workgroups
#ifdef PV_DEC3_PROF
#define PV_D3(i) do { __builtin_amdgcn_sched_barrier(0); const unsigned long long t_ = __builtin_amdgcn_s_memtime(); \
    d3acc[i] += t_ - d3last; d3last = t_; __builtin_amdgcn_sched_barrier(0); } while (0)
#else
#define PV_D3(i) do {} while (0)
#endif
#define PV_ROW(r, h) (((r) & 3) + 8 * ((r) >> 2) + 4 * (h))
// index of element (row b, column n) of a 32 x 32 result in the [16][64] accumulator-ordered LDS tile: inverse of PV_ROW
#define PV_TILE_IDX(b, n) (((((b) & 3) + 4 * ((b) >> 3)) << 6) + ((((b) >> 2) & 1) << 5) + (n))

// Optional tails of the per-layer dense kernels (widths above 128), for the two layers that are a single workgroup:
// the heads layer's forward also does the sampling layer + KL (pvk_reparam), the first decoder layer's backward also
// turns dz into the heads' gradient (pvk_reparam_bwd).  z == NULL / dheads == NULL: no tail.
struct PvRepFwd { const float* eps; float* z; float* kl; int K, ldz; };
struct PvRepBwd { const float* heads; const float* eps; float* dheads; int ldh, K; };

enum { PV_ACT_LINEAR = 0, PV_ACT_ELU = 1 };
// TRAIN: Adam in the pass.  GRAD: gradient dump (PvStep.acc: store / add).  TRAIN_ACC: Adam on (this pass's gradient +
// the dump of the earlier tiles) -- the last 32-row tile of a batch above 32 rows.
enum { PV_MODE_TRAIN = 0, PV_MODE_GRAD = 1, PV_MODE_TRAIN_ACC = 2 };

struct PvStep {  // device-resident per-step scalars (graph-replay friendly)
    int t;        // Adam iteration (1-based after pvk_step_begin)
    float lr;     // float32 learning rate variable (ReduceLROnPlateau mutates it)
    float alpha;  // lr * sqrt(1 - b2^t) / (1 - b1^t)
    int nrows;    // valid rows of the current batch tile
    float inv_b;  // 1 / nrows
    int step_in_epoch;
    int acc;      // gradient dumps (MODE_GRAD): 0 = store, 1 = add to what is there (batches above 32 rows are
                  // processed as 32-row tiles whose gradients are accumulated before ONE Adam update)
    int bce_probs;  // K5: 0 = binary cross-entropy from the logits (what Keras does for a Sigmoid output in graph
                    // mode), 1 = from probabilities clipped to [1e-7, 1 - 1e-7] (Keras' other path).  pvae_set_bce_mode.
};

struct PvAdam {
    float b1, b2, eps;
};

// ELU(alpha = 1).  exp via v_exp_f32 (abs. error ~1e-7 on a value in (-1, 0]) instead of the ~60-instruction expm1f:
// the narrow stack runs on one CU, where VALU instruction count is wall time.
__device__ __forceinline__ float pv_elu(float x) { return x > 0.f ? x : __expf(x) - 1.f; }
// derivative of ELU expressed through its output y (alpha = 1): y > 0 ? 1 : y + 1
__device__ __forceinline__ float pv_elu_grad_from_out(float y) { return y > 0.f ? 1.f : y + 1.f; }

__device__ __forceinline__ void pv_adam(float& w, float& m, float& v, float g, float alpha, const PvAdam& A) {
    m += (g - m) * (1.f - A.b1);
    v += (g * g - v) * (1.f - A.b2);
    // v_sqrt_f32 / v_rcp_f32 (1 ulp each) instead of the ~20-instruction IEEE expansions
    w -= (m * alpha) * __builtin_amdgcn_rcpf(__builtin_amdgcn_sqrtf(v) + A.eps);
}

// code(ROW(r,h)) of a 64-bit word holding 32 2-bit codes, after pre-shifting by 8*h bits
__device__ __forceinline__ float pv_code(uint32_t lo, uint32_t hi, int r) {
    const uint32_t w = (r >> 3) ? hi : lo;
    const int sh = 2 * (r & 3) + 16 * ((r >> 2) & 1);
    return (float)((w >> sh) & 3u);
}

// One element of keras.losses.binary_crossentropy on the sigmoid output layer (popvae.py:407-409, 417-418) and its derivative
// with respect to the logit z (not yet divided by the batch size); x = genotype value in {0, 0.5, 1}.
//   probs == 0: from the logits, max(z,0) - z x + log(1 + exp(-|z|)),  d = sigmoid(z) - x;
//   probs != 0: Keras' probability form in float32, -[x log(pc + 1e-7) + (1 - x) log(1 - pc + 1e-7)] with pc = clip(p, 1e-7,
//               1 - 1e-7); the clip passes no gradient outside its range, d sigmoid / dz = p (1 - p).
// The two differ only where |z| > ~16 (there the clipped form stops learning).  The mode is uniform: the callers branch
// once around their 16-value loop (a branch per value cost the grouped decoder 12 more spilled registers).
// -DPV_BCE_LOGITS_ONLY: builds without the switch (A/B of what it costs)
#ifdef PV_BCE_LOGITS_ONLY
#define PV_BCE_MODE(st) 0
#else
#define PV_BCE_MODE(st) ((st)->bce_probs)
#endif
// exp / log through the bare v_exp_f32 / v_log_f32 (base 2, 1 ulp): every argument below is a normal number (1 + e in [1, 2],
// clipped probabilities in [1e-7, 1]), so the library forms' denormal scaling and two-constant ln 2 product (12 instructions
// per log) buy nothing; the sigmoid / BCE phase of a subtile visit is VALU time on the critical path of the T-tile kernel.
template <int PROBS>
__device__ __forceinline__ void pv_bce(float z, float x, float& bce, float& d) {
    const float kLn2 = 0.69314718056f, kLog2e = 1.44269504089f;
    const float e = __builtin_amdgcn_exp2f(-kLog2e * fabsf(z));  // exp(-|z|)
    const float s = 1.f + e;
    const float inv = __builtin_amdgcn_rcpf(s);
    const float p = z >= 0.f ? inv : e * inv;          // sigmoid(z)
    if (!PROBS) {
        bce = fmaf(__builtin_amdgcn_logf(s), kLn2, fmaf(-z, x, fmaxf(z, 0.f)));
        d = p - x;
    } else {
        const float e7 = 1e-7f;
        const float pc = fminf(fmaxf(p, e7), 1.f - e7);
        const float a1 = pc + e7, a0 = 1.f - pc + e7;
        bce = -kLn2 * (x * __builtin_amdgcn_logf(a1) + (1.f - x) * __builtin_amdgcn_logf(a0));
        const bool inside = p >= e7 && p <= 1.f - e7;
        d = inside ? ((1.f - x) * __builtin_amdgcn_rcpf(a0) - x * __builtin_amdgcn_rcpf(a1)) * (p * (1.f - p)) : 0.f;
    }
}

// One 32 x 32 accumulator fragment of logits (lane = SNP column c, registers = rows PV_ROW(r, h) of the tile starting at batch row
// row0) -> dlogits / B in place, BCE terms added to lossacc, the lane's column sum of dlogits to dbias.
//   MASK: rows beyond nrows / SNPs beyond L (lvalid) may be missing.  Without it the 16 values are straight-line code that
//         overlaps their exp / rcp / log latencies; with it the compiler builds a branch per value (~150 instructions each).
//   LOSS: this wave's BCE terms count (several waves may hold the same logits; the others skip the logarithm).
template <int PROBS, int MASK, int LOSS>
__device__ __forceinline__ void pv_sigmoid_frag(f32x16& a, float bias, uint32_t lo, uint32_t hi, int h, int row0, int nrows,
                                                bool lvalid, float inv_b, float& lossacc, float& dbias) {
#pragma unroll
    for (int r = 0; r < 16; ++r) {
        const float x = 0.5f * pv_code(lo, hi, r);
        const float z = a[r] + bias;
        float bce, dz;
        pv_bce<PROBS>(z, x, bce, dz);
        if (MASK) {
            const bool ok = lvalid && (row0 + PV_ROW(r, h) < nrows);
            if (LOSS) lossacc += ok ? bce : 0.f;
            a[r] = ok ? dz * inv_b : 0.f;
        } else {
            if (LOSS) lossacc += bce;
            a[r] = dz * inv_b;
        }
        dbias += a[r];
    }
}
// probs, full (no masks needed), loss: wave-uniform
__device__ __forceinline__ void pv_sigmoid_bce(int probs, bool full, bool loss, f32x16& a, float bias, uint32_t lo, uint32_t hi,
                                               int h, int row0, int nrows, bool lvalid, float inv_b, float& lossacc,
                                               float& dbias) {
#define PV_SB(P, M, Ls) pv_sigmoid_frag<P, M, Ls>(a, bias, lo, hi, h, row0, nrows, lvalid, inv_b, lossacc, dbias)
    if (probs) {
        if (full) { if (loss) PV_SB(1, 0, 1); else PV_SB(1, 0, 0); }
        else      { if (loss) PV_SB(1, 1, 1); else PV_SB(1, 1, 0); }
    } else {
        if (full) { if (loss) PV_SB(0, 0, 1); else PV_SB(0, 0, 0); }
        else      { if (loss) PV_SB(0, 1, 1); else PV_SB(0, 1, 0); }
    }
#undef PV_SB
}

// ---------------------------------------------------------------------------------------------
// per-step scalars
// ---------------------------------------------------------------------------------------------
__global__ void pvk_step_begin(PvStep* st, const int* __restrict__ n_total_ptr, int n_total_val, int batch, int step,
                               int train, float b1, float b2) {
    if (threadIdx.x == 0 && blockIdx.x == 0) {
        int remaining = (n_total_ptr ? *n_total_ptr : n_total_val) - step * batch;
        int nrows = remaining < batch ? remaining : batch;
        if (nrows < 0) nrows = 0;
        st->nrows = nrows;
        st->inv_b = nrows > 0 ? 1.f / (float)nrows : 0.f;
        st->step_in_epoch = step;
        st->acc = 0;
        if (train) {
            int t = st->t + 1;
            st->t = t;
            // Keras: alpha = lr * sqrt(1 - beta_2^t) / (1 - beta_1^t), evaluated in float32
            float b1p = powf(b1, (float)t), b2p = powf(b2, (float)t);
            st->alpha = st->lr * sqrtf(1.f - b2p) / (1.f - b1p);
        }
    }
}

// one 32-row tile of a larger batch: rows of this tile, 1/B of the WHOLE batch, accumulate flag
__global__ void pvk_tile_begin(PvStep* st, int nrows, float inv_b_total, int acc) {
    if (threadIdx.x == 0 && blockIdx.x == 0) {
        st->nrows = nrows;
        st->inv_b = inv_b_total;
        st->acc = acc;
    }
}

// ---------------------------------------------------------------------------------------------
// Batch gather + 2-bit transpose: Xt[l] (u64, sample b at bits 2b) from the resident packed matrix.
// One wave handles two SNP blocks (lanes 0-31 / 32-63 = sample b of block 2p / 2p+1).
// Xr (optional): the gathered words themselves, Xr[blk * 32 + b] = row b's 32 codes of block blk -- the operand order of the
// encoder forward, 256 contiguous bytes per block instead of 32 cache lines of the resident matrix (8 bytes used of each).
// ---------------------------------------------------------------------------------------------
__device__ __forceinline__ void pv_gather_block(const uint8_t* __restrict__ geno, int64_t row_bytes,
                                                const int* __restrict__ rows, int nrows, int row_offset, int nblk,
                                                uint64_t* __restrict__ Xt, uint64_t* __restrict__ Xr, int block) {
    __shared__ uint64_t sm[4][64];
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int b = lane & 31, half = lane >> 5;
    const int pair = block * 4 + wave;
    const int blk = pair * 2 + half;
    uint64_t w = 0;
    if (blk < nblk && b < nrows) {
        const int64_t row = rows[row_offset + b];
        w = *reinterpret_cast<const uint64_t*>(geno + row * row_bytes + (int64_t)blk * 8);
    }
    sm[wave][lane] = w;
    if (Xr && blk < nblk) Xr[(int64_t)blk * 32 + b] = w;
    __builtin_amdgcn_s_waitcnt(0xc07f);  // lgkmcnt(0); wave-private LDS row, no barrier needed
    __builtin_amdgcn_wave_barrier();
    // lane (l = lane&31, half) assembles SNP l of block `blk`: bits 2l..2l+1 of each sample's word
    uint64_t out = 0;
    const int sh = 2 * b;
#pragma unroll
    for (int s = 0; s < 32; ++s) {
        const uint64_t x = sm[wave][half * 32 + s];
        out |= ((x >> sh) & 3ull) << (2 * s);
    }
    if (blk < nblk) Xt[(int64_t)blk * 32 + b] = out;
}

__global__ __launch_bounds__(256) void pvk_gather_transpose(const uint8_t* __restrict__ geno, int64_t row_bytes,
                                                            const int* __restrict__ rows, const PvStep* st,
                                                            int row_offset, int nblk, uint64_t* __restrict__ Xt,
                                                            uint64_t* __restrict__ Xr) {
    // blockIdx.y = 32-row tile of the device pass: its rows follow the previous tile's, its words the previous tile's array
    const int tile = blockIdx.y;
    pv_gather_block(geno, row_bytes, rows, max(0, min(32, st->nrows - 32 * tile)), row_offset + 32 * tile, nblk,
                    Xt + (int64_t)tile * nblk * 32, Xr ? Xr + (int64_t)tile * nblk * 32 : nullptr, blockIdx.x);
}

// ---------------------------------------------------------------------------------------------
// Wide encoder forward: partial[b][u] = sum_l code(b,l) * W1[l][u] over this wave's 32 x 32 blocks.
// Block-column-major, exactly as the fused forward inside pvk_enc_bwd (pv_enc_bwd_train): the waves of a unit group
// (grp = blockIdx.x % ngrp) are numbered gw; wave gw owns block column w = gw % NWB of the subtiles slot, slot + nslots, ...
// (slot = gw / NWB) and writes its 32 x 32 fragment into the partial tile part[slot][32][Wp].  Launched on the same
// number of workgroups as that pass, both produce bit-identical partial tiles (same products in the same order), which
// pvk_reduce_parts sums in a fixed order: a step computes the same numbers whichever of the two paths fed it.
// ---------------------------------------------------------------------------------------------
// T = 32-row tiles of the device pass (1; 2 or 4 for a batch above 32 rows in ONE weight pass: the block is requested once
// and multiplied with every tile's codes).  Rows of tile t are rows[row_offset + 32 t ..]; the partial tile is [32 T][Wp].
// BF = 1 (passes over several row tiles, which the f32 matrix pipe bounds: 4 tiles x 16 x 64 cycles per block against ~200 of
// HBM time): the same products on the bf16 pipe at f32 accuracy.  Genotype codes {0, 1, 2} are exact in bf16; the weight is cut
// into three bf16 terms, w = hi + mid + lo exactly (3 x 8 significand bits, by truncation: x & 0xFFFF0000, remainder, again),
// every product code x term is exact and the accumulator is f32: 6 x v_mfma_f32_32x32x16_bf16 (32 cycles each) per block and
// tile instead of 16 x v_mfma_f32_32x32x2_f32 (64 cycles each).  The contraction slots of the K = 16 instruction are filled in
// the order the fragment-native tile already has: lane group h, element j of instruction i <-> SNP PV_ROW(8 i + j, h), for both
// operands.  A operand: four bytes of the lane's genotype word (4 codes each) through a 256-entry table in LDS.
typedef __bf16 pv_bf16x8 __attribute__((ext_vector_type(8)));
typedef uint32_t pv_u32x4 __attribute__((ext_vector_type(4)));
typedef uint32_t pv_u32x2 __attribute__((ext_vector_type(2)));
// bf16 bit patterns of the four 2-bit codes in a byte (code 3 never occurs: pvae_load_genotypes rejects it)
__device__ __forceinline__ pv_u32x2 pv_code_bf16_entry(int byte) {
    auto f = [](int c) -> uint32_t { return c == 0 ? 0u : c == 1 ? 0x3F80u : 0x4000u; };
    return pv_u32x2{f(byte & 3) | (f((byte >> 2) & 3) << 16), f((byte >> 4) & 3) | (f((byte >> 6) & 3) << 16)};
}
// high halves of two floats -> one register of two bf16 (truncation)
__device__ __forceinline__ uint32_t pv_pack_hi16(float a, float b) {
    return __builtin_amdgcn_perm(__float_as_uint(b), __float_as_uint(a), 0x07060302u);
}
// x[0..7] -> three bf16x8 operands whose sum is x exactly
__device__ __forceinline__ void pv_split3(const float (&x)[8], pv_u32x4& hi, pv_u32x4& mid, pv_u32x4& lo) {
    float r1[8], r2[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) {
        r1[i] = x[i] - __uint_as_float(__float_as_uint(x[i]) & 0xFFFF0000u);
        r2[i] = r1[i] - __uint_as_float(__float_as_uint(r1[i]) & 0xFFFF0000u);
    }
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        hi[i] = pv_pack_hi16(x[2 * i], x[2 * i + 1]);
        mid[i] = pv_pack_hi16(r1[2 * i], r1[2 * i + 1]);
        lo[i] = pv_pack_hi16(r2[2 * i], r2[2 * i + 1]);
    }
}
// pv_split3 in six stages of 4 .. 8 VALU instructions, to be placed by hand between matrix instructions (one stage hides behind
// one v_mfma_f32_32x32x16_bf16, which issues every 32 cycles): stages 0-3 cut two values each, 4 packs hi and mid, 5 packs lo
struct PvCut3 {
    float x[8], r1[8], r2[8];
};
template <int STAGE>
__device__ __forceinline__ void pv_cut_stage(PvCut3& c, pv_u32x4& hi, pv_u32x4& mid, pv_u32x4& lo) {
    if constexpr (STAGE < 4) {
#pragma unroll
        for (int i = 2 * STAGE; i < 2 * STAGE + 2; ++i) {
            c.r1[i] = c.x[i] - __uint_as_float(__float_as_uint(c.x[i]) & 0xFFFF0000u);
            c.r2[i] = c.r1[i] - __uint_as_float(__float_as_uint(c.r1[i]) & 0xFFFF0000u);
        }
    } else if constexpr (STAGE == 4) {
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            hi[i] = pv_pack_hi16(c.x[2 * i], c.x[2 * i + 1]);
            mid[i] = pv_pack_hi16(c.r1[2 * i], c.r1[2 * i + 1]);
        }
    } else {
#pragma unroll
        for (int i = 0; i < 4; ++i) lo[i] = pv_pack_hi16(c.r2[2 * i], c.r2[2 * i + 1]);
    }
}
__device__ __forceinline__ f32x16 pv_mfma_bf16(pv_u32x4 a, pv_u32x4 b, f32x16 c) {
    return __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(pv_bf16x8, a), __builtin_bit_cast(pv_bf16x8, b), c, 0, 0, 0);
}

template <int NWB, int T = 1, int BF = 0>
__global__ __launch_bounds__(256) void pvk_enc_fwd(const float* __restrict__ W1, const uint8_t* __restrict__ geno,
                                                   int64_t row_bytes, const int* __restrict__ rows,
                                                   const PvStep* st, int row_offset, int nblk, int ngrp,
                                                   float* __restrict__ part, int ts, const uint64_t* __restrict__ Xr) {
    // Xr != NULL: the batch's words as pvk_gather_transpose left them (same values as the resident matrix's, compact)
    constexpr int WG_UNITS = NWB * 32;
    __shared__ pv_u32x2 code_lut[BF ? 256 : 1];
    if (BF) {
        code_lut[threadIdx.x & 255] = pv_code_bf16_entry(threadIdx.x & 255);
        __syncthreads();
    }
    const int wave = __builtin_amdgcn_readfirstlane((int)threadIdx.x >> 6), lane = threadIdx.x & 63, c = lane & 31, h = lane >> 5;
    const int grp = blockIdx.x % ngrp, gw = (blockIdx.x / ngrp) * 4 + wave;
    const int slot = gw / NWB, w = gw - slot * NWB;
    const int nslots = (gridDim.x / ngrp) * 4 / NWB;  // the host sizes the grid so that this divides
    const int nrows = st->nrows;
    const uint8_t* xrow[T];
#pragma unroll
    for (int t = 0; t < T; ++t) xrow[t] = 32 * t + c < nrows ? geno + (int64_t)rows[row_offset + 32 * t + c] * row_bytes : nullptr;

    f32x16 acc[T];
#pragma unroll
    for (int t = 0; t < T; ++t)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[t][r] = 0.f;
    auto mma = [&](const f32x4 (&wv)[4], const uint64_t (&xw)[T]) {
#ifdef PV_EXP_NOMMA  // bound-finder builds: the kernel without its matrix instructions (NOMMA) / without its loads (NOLOAD)
#pragma unroll
        for (int t = 0; t < T; ++t) acc[t][0] += (float)(uint32_t)xw[t] + wv[0][0] + wv[1][1] + wv[2][2] + wv[3][3];
#else
        if constexpr (BF) {
            pv_u32x4 bh[2], bm[2], bl[2];
#pragma unroll
            for (int i = 0; i < 2; ++i) {
                const float x[8] = {wv[2 * i][0], wv[2 * i][1], wv[2 * i][2], wv[2 * i][3],
                                    wv[2 * i + 1][0], wv[2 * i + 1][1], wv[2 * i + 1][2], wv[2 * i + 1][3]};
                pv_split3(x, bh[i], bm[i], bl[i]);
            }
#pragma unroll
            for (int t = 0; t < T; ++t) {
                const uint64_t xs = xw[t] >> (8 * h);
                const uint32_t lo = (uint32_t)xs, hi = (uint32_t)(xs >> 32);
                // codes r = 0..3 | 4..7 | 8..11 | 12..15 of pv_code: bytes 0 and 2 of the two halves
                const pv_u32x2 c0 = code_lut[lo & 0xFF], c1 = code_lut[(lo >> 16) & 0xFF], c2 = code_lut[hi & 0xFF],
                               c3 = code_lut[(hi >> 16) & 0xFF];
                const pv_u32x4 a0 = {c0[0], c0[1], c1[0], c1[1]}, a1 = {c2[0], c2[1], c3[0], c3[1]};
                acc[t] = pv_mfma_bf16(a0, bl[0], acc[t]);
                acc[t] = pv_mfma_bf16(a1, bl[1], acc[t]);
                acc[t] = pv_mfma_bf16(a0, bm[0], acc[t]);
                acc[t] = pv_mfma_bf16(a1, bm[1], acc[t]);
                acc[t] = pv_mfma_bf16(a0, bh[0], acc[t]);
                acc[t] = pv_mfma_bf16(a1, bh[1], acc[t]);
            }
        } else {
#pragma unroll
            for (int t = 0; t < T; ++t) {
                const uint64_t xs = xw[t] >> (8 * h);
                const uint32_t lo = (uint32_t)xs, hi = (uint32_t)(xs >> 32);
#pragma unroll
                for (int r = 0; r < 16; ++r)
                    acc[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(pv_code(lo, hi, r), wv[r >> 2][r & 3], acc[t], 0, 0, 0);
            }
        }
#endif
    };
    auto request = [&](f32x4 (&wv)[4], uint64_t (&xw)[T], int blk_) {
        const f32x4* tile = reinterpret_cast<const f32x4*>(W1) + ((int64_t)blk_ * ngrp + grp) * ts + w * 256 + lane;
#ifdef PV_EXP_NOLOAD
        if (blk_ >= 4 * nslots) return;  // later groups reuse what the first one loaded
#endif
#pragma unroll
        for (int q = 0; q < 4; ++q) wv[q] = __builtin_nontemporal_load(tile + q * 64);
#pragma unroll
        for (int t = 0; t < T; ++t)
            xw[t] = Xr ? Xr[((int64_t)t * nblk + blk_) * 32 + c]
                       : xrow[t] ? *reinterpret_cast<const uint64_t*>(xrow[t] + (int64_t)blk_ * 8) : 0ull;
    };
    int blk = slot;
#ifdef PV_DEC3_PROF
    const unsigned long long ef_c0 = __builtin_amdgcn_s_memtime(), ef_r0 = __builtin_amdgcn_s_memrealtime();
#endif
    // four blocks (16 KiB per wave) requested before the first is used (a rotating four-buffer pipeline, each buffer
    // refilled as soon as it is used, measured slower: 132 vs 118 us at 1M SNPs)
#ifdef PV_EXP_NOLOAD
    f32x4 wv[4][4];
    uint64_t xw[4][T];
#endif
    for (; blk + 3 * nslots < nblk; blk += 4 * nslots) {
#ifndef PV_EXP_NOLOAD
        f32x4 wv[4][4];
        uint64_t xw[4][T];
#endif
#pragma unroll
        for (int u = 0; u < 4; ++u) request(wv[u], xw[u], blk + u * nslots);
#pragma unroll
        for (int u = 0; u < 4; ++u) mma(wv[u], xw[u]);
    }
    for (; blk < nblk; blk += nslots) {
        f32x4 wv[4];
        uint64_t xw[T];
        request(wv, xw, blk);
        mma(wv, xw);
    }
#ifdef PV_DEC3_PROF
    if (T > 1 && threadIdx.x == 0 && blockIdx.x < 1024) {  // shader cycles and 100 MHz ticks of the sweep
        asm volatile("s_nop 0" ::"v"(acc[0][0]), "v"(acc[T - 1][15]));
        pv_wtrace_buf[4096 + blockIdx.x * 2] = __builtin_amdgcn_s_memtime() - ef_c0;
        pv_wtrace_buf[4096 + blockIdx.x * 2 + 1] = __builtin_amdgcn_s_memrealtime() - ef_r0;
    }
#endif
    const int ldp = ngrp * WG_UNITS;
    float* p = part + (int64_t)slot * (32 * T * ldp) + grp * WG_UNITS + w * 32 + c;
#pragma unroll
    for (int t = 0; t < T; ++t)
#pragma unroll
        for (int r = 0; r < 16; ++r) p[(32 * t + PV_ROW(r, h)) * ldp] = acc[t][r];
}

// ---------------------------------------------------------------------------------------------
// Fixed-order reduction of per-workgroup partials: out[e] = scale * sum_p part[p*n + e], e < n (n % 16 == 0),
// and optionally sout[0] = sum_p spart[p].  grid = n / 16 blocks (PV_REDUCE_GRID); block = 4 float4 columns x 64
// slices; slice s adds partials s, s+64, ... in order (<= 8 independent loads in flight per thread at 512 partials: the
// kernel is pure latency, 64 wide blocks of 32 dependent steps each took 10 us), then the 64 slice sums are added in
// order: bitwise reproducible.
// ---------------------------------------------------------------------------------------------
#define PV_REDUCE_GRID(n) ((n) > 0 ? (n) / 16 : 1)
__device__ __forceinline__ void pv_reduce_block(const float* __restrict__ part, int nparts, int n, float scale,
                                                float* __restrict__ out, const float* __restrict__ spart,
                                                float* __restrict__ sout, int block) {
    __shared__ f32x4 sm[64][4];
    __shared__ float ss[256];
    const int tid = threadIdx.x, col = tid & 3, slice = tid >> 2;
    if (block * 16 < n) {
        const f32x4* src = reinterpret_cast<const f32x4*>(part) + block * 4 + col;
        const int n4 = n >> 2;
        f32x4 acc = {0.f, 0.f, 0.f, 0.f};
#pragma unroll 8
        for (int p = slice; p < nparts; p += 64) acc += src[(int64_t)p * n4];
        sm[slice][col] = acc;
    }
    if (block == 0 && spart) {
        float a = 0.f;
        for (int p = tid; p < nparts; p += 256) a += spart[p];
        ss[tid] = a;
    }
    __syncthreads();
    if (block * 16 < n && tid < 4) {
        f32x4 acc = sm[0][tid];
#pragma unroll 8
        for (int s2 = 1; s2 < 64; ++s2) acc += sm[s2][tid];
        reinterpret_cast<f32x4*>(out)[block * 4 + tid] = acc * scale;
    }
    if (block == 0 && spart && tid == 0) {
        float a = 0.f;
        for (int i = 0; i < 256; ++i) a += ss[i];
        sout[0] = a;
    }
}
__global__ __launch_bounds__(256) void pvk_reduce_parts(const float* __restrict__ part, int nparts, int n, float scale,
                                                        float* __restrict__ out, const float* __restrict__ spart,
                                                        float* __restrict__ sout) {
    pv_reduce_block(part, nparts, n, scale, out, spart, sout, blockIdx.x);
}

// h = act(P + bias)  (first encoder layer epilogue; after the SNP-shard all-reduce when world > 1)
__global__ void pvk_bias_act(const float* __restrict__ P, const float* __restrict__ bias, int n, int act,
                             float* __restrict__ out) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= 32 * n) return;
    float x = P[i] + bias[i % n];
    out[i] = act == PV_ACT_ELU ? pv_elu(x) : x;
}

// ---------------------------------------------------------------------------------------------
// Narrow dense layer forward: out[32][N] = act(in[32][K] * Wt[K][N] + bias).  grid = N/32 workgroups,
// 4 waves split the K range; fixed-order LDS reduce.
// ---------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void pvk_dense_fwd(const float* __restrict__ in, int K, const float* __restrict__ Wt,
                                                     const float* __restrict__ bias, int N, int act,
                                                     float* __restrict__ out, const float* __restrict__ in_bias,
                                                     float* __restrict__ in_act, PvRepFwd rp) {
    // in_bias != NULL: the input is a pre-activation, elu(in + in_bias) is formed while staging and workgroup 0 also
    // writes it to in_act (the first hidden layer: pvk_bias_act without a launch of its own)
    extern __shared__ __attribute__((aligned(16))) float smem[];
    float* in_s = smem;                  // [32][K+1]
    float* red = smem + 32 * (K + 1);    // [4][16][64]
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63, c = lane & 31, h = lane >> 5;
    const int n0 = blockIdx.x * 32;
    f32x16 acc;
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[r] = 0.f;
    const int kper = K / 4;  // this wave's share of the contraction, a multiple of 8
    // one float4 of the input tile: (bias + ELU on the way in,) scattered into the padded LDS rows
    auto put = [&](int i, f32x4 v) {
        if (in_bias) {
            const f32x4 bq = *reinterpret_cast<const f32x4*>(in_bias + i % K);
#pragma unroll
            for (int e = 0; e < 4; ++e) v[e] = pv_elu(v[e] + bq[e]);
            if (blockIdx.x == 0) *reinterpret_cast<f32x4*>(in_act + i) = v;
        }
        float* d = in_s + (i / K) * (K + 1) + (i % K);
        d[0] = v[0]; d[1] = v[1]; d[2] = v[2]; d[3] = v[3];
    };
    // Widths 512 / 256: ONE round trip per layer -- the wave's whole k range of weights (S loads) is requested first, then
    // the whole input tile (NU float4 per thread), and only then anything is waited for.  (16 steps per group and 4 staging
    // loads in flight, the first version, paid 4 + 4 dependent round trips: 11 us per layer against 6-7.)
    auto whole = [&](auto Sc, auto NUc) {
        constexpr int S = decltype(Sc)::value, NU = decltype(NUc)::value;
        float b[S];
        const int k = wave * kper;
#pragma unroll
        for (int s = 0; s < S; ++s) b[s] = Wt[(int64_t)(k + 2 * s + h) * N + n0 + c];
        f32x4 v[NU];
#pragma unroll
        for (int u = 0; u < NU; ++u) v[u] = reinterpret_cast<const f32x4*>(in)[tid + 256 * u];
#pragma unroll
        for (int u = 0; u < NU; ++u) put(4 * (tid + 256 * u), v[u]);
        __syncthreads();
#pragma unroll
        for (int s = 0; s < S; ++s) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(in_s[c * (K + 1) + k + 2 * s + h], b[s], acc, 0, 0, 0);
    };
    if (K == 512) {
        whole(std::integral_constant<int, 64>{}, std::integral_constant<int, 16>{});
    } else if (K == 256) {
        whole(std::integral_constant<int, 32>{}, std::integral_constant<int, 8>{});
    } else {
        // 16-byte loads, 4 in flight per thread (K is a multiple of 32, so a float4 never straddles a row)
        for (int i0 = tid; i0 < 8 * K; i0 += 1024) {
            f32x4 v[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) v[u] = reinterpret_cast<const f32x4*>(in)[min(i0 + 256 * u, 8 * K - 1)];
#pragma unroll
            for (int u = 0; u < 4; ++u)
                if (4 * (i0 + 256 * u) < 32 * K) put(4 * (i0 + 256 * u), v[u]);
        }
        __syncthreads();
        // 4 MFMA steps (8 k) per group, all operand loads of a group issued before its MFMAs; a load inside a
        // rolled loop is waited for every iteration (one L2 round trip per MFMA)
        auto run = [&](auto G) {
            constexpr int S = decltype(G)::value;  // MFMA steps per group
            for (int k = wave * kper; k < (wave + 1) * kper; k += 2 * S) {
                float a[S], b[S];
#pragma unroll
                for (int s = 0; s < S; ++s) {
                    a[s] = in_s[c * (K + 1) + k + 2 * s + h];
                    b[s] = Wt[(int64_t)(k + 2 * s + h) * N + n0 + c];
                }
#pragma unroll
                for (int s = 0; s < S; ++s) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a[s], b[s], acc, 0, 0, 0);
            }
        };
        if (kper % 32 == 0) run(std::integral_constant<int, 16>{});
        else run(std::integral_constant<int, 4>{});
    }
#pragma unroll
    for (int r = 0; r < 16; ++r) red[(wave * 16 + r) * 64 + lane] = acc[r];
    __syncthreads();
    for (int e = tid; e < 1024; e += 256) {
        const float s = ((red[e] + red[1024 + e]) + red[2048 + e]) + red[3072 + e];
        const int r = e >> 6, l = e & 63;
        const int b = PV_ROW(r, l >> 5), n = n0 + (l & 31);
        const float x = s + bias[n];
        const float y = act == PV_ACT_ELU ? pv_elu(x) : x;
        out[b * N + n] = y;
        if (rp.z) red[e] = y;  // slot e of the first quarter is read by this thread only
    }
    if (rp.z) {  // heads layer (N == 32, one workgroup): z = mu + exp(s) * eps and the KL term, as pvk_reparam
        __syncthreads();
        const int b = tid;
        if (b < 32) {
            float kacc = 0.f;
            for (int k = 0; k < rp.ldz; ++k) {
                float zz = 0.f;
                if (k < rp.K) {
                    const float mu = red[PV_TILE_IDX(b, k)], sv = red[PV_TILE_IDX(b, rp.K + k)];
                    const float es = expf(sv);
                    zz = mu + es * rp.eps[b * rp.K + k];  // NOTE exp(s), not exp(s/2): popvae.py:389
                    kacc += 1.f + sv - mu * mu - es;
                }
                rp.z[b * rp.ldz + k] = zz;
            }
            rp.kl[b] = -0.5f * kacc;
        }
    }
}

// ---------------------------------------------------------------------------------------------
// The same layer for contraction lengths 256 / 512 (the hidden layers at widths above 128), round 3: SIXTEEN output columns per
// workgroup and eight waves that split the contraction (v_mfma_f32_16x16x4_f32, two 16-row tiles per wave).  A layer is a few
// microseconds of which the matrix instructions were 1.75 (32 columns per workgroup, four waves: 64 dependent instructions of 64
// cycles per wave); with twice the workgroups a CU's matrix pipe has half the work (0.87 us) and draws half the weights (32 KB).
// The input tile sits in LDS with a row stride of K + 4 floats: the A operand reads of a wave (lane (row, quarter) reads word
// row * (K + 4) + k + quarter) then fall on 64 different banks, and the staging writes are whole float4s.
// grid = N / 16, 512 threads; LDS = 32 x (K + 4) + 8 x 2 x 4 x 64 floats.
// ---------------------------------------------------------------------------------------------
typedef float f32x4_mfma __attribute__((ext_vector_type(4)));
template <int K>
__global__ __launch_bounds__(512) void pvk_dense_fwd16(const float* __restrict__ in, const float* __restrict__ Wt,
                                                       const float* __restrict__ bias, int N, int act, float* __restrict__ out,
                                                       const float* __restrict__ in_bias, float* __restrict__ in_act) {
    static_assert(K == 256 || K == 512, "contraction lengths of the hidden layers at widths 256 / 512");
    constexpr int LD = K + 4, KPER = K / 8, S = KPER / 4, NU = 32 * K / 4 / 512;
    extern __shared__ __attribute__((aligned(16))) float smem[];
    float* in_s = smem;             // [32][LD]
    float* red = smem + 32 * LD;    // [8 waves][2 tiles][4][64]
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63, n = lane & 15, q = lane >> 4;
    const int n0 = blockIdx.x * 16, k0 = wave * KPER;
    // one round trip: this wave's weights (S loads), then the input tile (NU float4 per thread), then the first wait
    float b[S];
#pragma unroll
    for (int s = 0; s < S; ++s) b[s] = Wt[(int64_t)(k0 + 4 * s + q) * N + n0 + n];
    f32x4 v[NU];
#pragma unroll
    for (int u = 0; u < NU; ++u) v[u] = reinterpret_cast<const f32x4*>(in)[tid + 512 * u];
#pragma unroll
    for (int u = 0; u < NU; ++u) {
        const int i = 4 * (tid + 512 * u), row = i / K, col = i % K;
        f32x4 x = v[u];
        if (in_bias) {  // the input is a pre-activation: elu(in + in_bias), written out by workgroup 0 (pvk_dense_fwd)
            const f32x4 bq = *reinterpret_cast<const f32x4*>(in_bias + col);
#pragma unroll
            for (int e = 0; e < 4; ++e) x[e] = pv_elu(x[e] + bq[e]);
            if (blockIdx.x == 0) *reinterpret_cast<f32x4*>(in_act + i) = x;
        }
        *reinterpret_cast<f32x4*>(in_s + row * LD + col) = x;
    }
    __syncthreads();
    f32x4_mfma acc0 = {0.f, 0.f, 0.f, 0.f}, acc1 = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int s = 0; s < S; ++s) {
        const float a0 = in_s[n * LD + k0 + 4 * s + q], a1 = in_s[(16 + n) * LD + k0 + 4 * s + q];
        acc0 = __builtin_amdgcn_mfma_f32_16x16x4f32(a0, b[s], acc0, 0, 0, 0);
        acc1 = __builtin_amdgcn_mfma_f32_16x16x4f32(a1, b[s], acc1, 0, 0, 0);
    }
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        red[((wave * 2 + 0) * 4 + r) * 64 + lane] = acc0[r];
        red[((wave * 2 + 1) * 4 + r) * 64 + lane] = acc1[r];
    }
    __syncthreads();
    // 512 outputs, one per thread: element (tile t, register r, lane l) = row 16 t + 4 (l >> 4) + r, column n0 + (l & 15); the eight
    // waves' partial sums are added in wave order
    {
        const int t = tid >> 8, r = (tid >> 6) & 3, l = tid & 63;
        float sum = red[((0 * 2 + t) * 4 + r) * 64 + l];
#pragma unroll
        for (int w = 1; w < 8; ++w) sum += red[((w * 2 + t) * 4 + r) * 64 + l];
        const int row = 16 * t + 4 * (l >> 4) + r, col = n0 + (l & 15);
        const float x = sum + bias[col];
        out[row * N + col] = act == PV_ACT_ELU ? pv_elu(x) : x;
    }
}

// ---------------------------------------------------------------------------------------------
// Narrow dense layer backward, chain part (widths above 128).  grid = K/32 workgroups (one per 32 input units):
//   dpre = dY * act'(Y)  (workgroup 0 also writes it to dpre_out for pvk_narrow_adam);
//   dIn[:, k-block] = dpre * Wt[k-block,:]^T.
// The weight / bias gradients and Adam run afterwards in pvk_narrow_adam (pvae_narrow.h), one wave per 32x32
// weight block over the whole chip, for all layers in one launch.
// ---------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void pvk_dense_bwd(int K, const float* __restrict__ Wt, int N, int act,
                                                     const float* __restrict__ Y, const float* __restrict__ dY,
                                                     float* __restrict__ dIn, float* __restrict__ dpre_out,
                                                     const PvStep* st, PvRepBwd rp) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const int ld = N + 1;
    float* dp_s = smem;             // [32][N+1]
    float* w_s = smem + 32 * ld;    // [32][N+1]
    float* red = w_s + 32 * ld;     // [4][16][64]
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63, c = lane & 31, h = lane >> 5;
    const int k0 = blockIdx.x * 32;
    const int nrows = st->nrows;
    // 16-byte loads, NU x 3 in flight per thread and round (N is a multiple of 32, so a float4 never straddles a row):
    // 8 at widths 256 / 512 (one / two rounds), 4 otherwise
    auto stage = [&](auto NUc) {
        constexpr int NU = decltype(NUc)::value;
        for (int i0 = tid; i0 < 8 * N; i0 += 256 * NU) {
            f32x4 yv[NU], dv[NU], wv[NU];
#pragma unroll
            for (int u = 0; u < NU; ++u) {
                const int i4 = min(i0 + 256 * u, 8 * N - 1), i = 4 * i4;
                const int b = i / N, n = i % N;
                dv[u] = reinterpret_cast<const f32x4*>(dY)[i4];
                if (act == PV_ACT_ELU) yv[u] = reinterpret_cast<const f32x4*>(Y)[i4];
                wv[u] = *reinterpret_cast<const f32x4*>(Wt + (int64_t)(k0 + b) * N + n);
            }
#pragma unroll
            for (int u = 0; u < NU; ++u) {
                const int i = 4 * (i0 + 256 * u);
                if (i < 32 * N) {
                    const int b = i / N, n = i % N;
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        float d = b < nrows ? dv[u][e] : 0.f;
                        if (act == PV_ACT_ELU) d *= pv_elu_grad_from_out(yv[u][e]);
                        dp_s[b * ld + n + e] = d;
                        w_s[b * ld + n + e] = wv[u][e];
                        if (blockIdx.x == 0) dpre_out[i + e] = d;
                    }
                }
            }
        }
    };
    if (N % 256 == 0) stage(std::integral_constant<int, 8>{});
    else stage(std::integral_constant<int, 4>{});
    __syncthreads();
    f32x16 acc;
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[r] = 0.f;
    const int nper = N / 4;  // multiple of 8
    auto run = [&](auto G) {
        constexpr int S = decltype(G)::value;
        for (int n = wave * nper; n < (wave + 1) * nper; n += 2 * S) {
            float a[S], b[S];
#pragma unroll
            for (int s = 0; s < S; ++s) {
                a[s] = dp_s[c * ld + n + 2 * s + h];  // A[i=b][k=n]
                b[s] = w_s[c * ld + n + 2 * s + h];   // B[k=n][j=k_loc]
            }
#pragma unroll
            for (int s = 0; s < S; ++s) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a[s], b[s], acc, 0, 0, 0);
        }
    };
    if (nper % 32 == 0) run(std::integral_constant<int, 16>{});
    else run(std::integral_constant<int, 4>{});
#pragma unroll
    for (int r = 0; r < 16; ++r) red[(wave * 16 + r) * 64 + lane] = acc[r];
    __syncthreads();
    for (int e = tid; e < 1024; e += 256) {
        const float s = ((red[e] + red[1024 + e]) + red[2048 + e]) + red[3072 + e];
        const int r = e >> 6, l = e & 63;
        dIn[PV_ROW(r, l >> 5) * K + k0 + (l & 31)] = s;
        if (rp.dheads) red[e] = s;  // slot e of the first quarter is read by this thread only
    }
    if (rp.dheads) {  // first decoder layer (K == 32, one workgroup): dz -> d heads, as pvk_reparam_bwd
        __syncthreads();
        const int b = tid;
        if (b < 32) {
            const bool valid = b < nrows;
            const float ib = st->inv_b;
            for (int k = 0; k < rp.ldh; ++k) rp.dheads[b * rp.ldh + k] = 0.f;
            if (valid) {
                for (int k = 0; k < rp.K; ++k) {
                    const float mu = rp.heads[b * rp.ldh + k], sv = rp.heads[b * rp.ldh + rp.K + k];
                    const float es = expf(sv), d = red[PV_TILE_IDX(b, k)];
                    rp.dheads[b * rp.ldh + k] = d + mu * ib;
                    rp.dheads[b * rp.ldh + rp.K + k] = d * rp.eps[b * rp.K + k] * es + 0.5f * (es - 1.f) * ib;
                }
            }
        }
    }
}

// ---------------------------------------------------------------------------------------------
// The backward of the hidden layers at widths 256 / 512 (N = the layer's outputs = the contraction length here), round 3:
// SIXTEEN input units per workgroup, eight waves that split the contraction (v_mfma_f32_16x16x4_f32).  Twice the workgroups of
// pvk_dense_bwd, half the matrix work (0.87 us instead of 1.75) and half the weights (32 KB, float4 loads along the rows of W) per
// CU.  Contraction step (s, e) of a wave pairs slot q of the instruction with output unit nb + 16 s + 4 q + e, so that a lane's
// four A operands of a group are ONE ds_read_b128 and its four B operands one global float4.
// Two ways of chaining layers without a pass over Y in between:
//   in_is_dpre != 0: dY already IS dpre of this layer (the producer's epilogue below formed it): staged as it comes, 64 KB per
//                    workgroup instead of dY and Y, nothing written to dpre_out;
//   Ynext != NULL:   the epilogue turns dIn straight into the NEXT layer's dpre = dIn * act'(Ynext), rows beyond the batch zeroed
//                    (Ynext = that layer's output = this layer's input activation, [32][K]), and `dIn` is that layer's dpre buffer
//                    (what pvk_narrow_adam_bias reads) -- as the fused chain does at widths up to 128.
// grid = K / 16, 512 threads; LDS = 32 x (N + 4) + 8 x 2 x 4 x 64 floats.
// ---------------------------------------------------------------------------------------------
template <int N>
__global__ __launch_bounds__(512) void pvk_dense_bwd16(int K, const float* __restrict__ Wt, int act, const float* __restrict__ Y,
                                                       const float* __restrict__ dY, int in_is_dpre, float* __restrict__ dIn,
                                                       float* __restrict__ dpre_out, const PvStep* st,
                                                       const float* __restrict__ Ynext, int act_next) {
    static_assert(N == 256 || N == 512, "widths 256 / 512");
    constexpr int LD = N + 4, NPER = N / 8, S4 = NPER / 16, NU = 32 * N / 4 / 512;
    extern __shared__ __attribute__((aligned(16))) float smem[];
    float* dp_s = smem;            // [32][LD]
    float* red = smem + 32 * LD;   // [8 waves][2 tiles][4][64]
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63, j = lane & 15, q = lane >> 4;
    const int k0 = blockIdx.x * 16, nb = wave * NPER;
    const int nrows = st->nrows;
    f32x4 wv[S4];
#pragma unroll
    for (int s = 0; s < S4; ++s) wv[s] = *reinterpret_cast<const f32x4*>(Wt + (int64_t)(k0 + j) * N + nb + 16 * s + 4 * q);
    f32x4 dv[NU], yv[NU];
#pragma unroll
    for (int u = 0; u < NU; ++u) {
        dv[u] = reinterpret_cast<const f32x4*>(dY)[tid + 512 * u];
        if (!in_is_dpre && act == PV_ACT_ELU) yv[u] = reinterpret_cast<const f32x4*>(Y)[tid + 512 * u];
    }
#pragma unroll
    for (int u = 0; u < NU; ++u) {
        const int i = 4 * (tid + 512 * u), row = i / N, col = i % N;
        f32x4 x = dv[u];
        if (!in_is_dpre) {
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                float d = row < nrows ? x[e] : 0.f;
                if (act == PV_ACT_ELU) d *= pv_elu_grad_from_out(yv[u][e]);
                x[e] = d;
            }
            if (blockIdx.x == 0) *reinterpret_cast<f32x4*>(dpre_out + i) = x;
        }
        *reinterpret_cast<f32x4*>(dp_s + row * LD + col) = x;
    }
    __syncthreads();
    f32x4_mfma acc0 = {0.f, 0.f, 0.f, 0.f}, acc1 = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int s = 0; s < S4; ++s) {
        const f32x4 a0 = *reinterpret_cast<const f32x4*>(dp_s + j * LD + nb + 16 * s + 4 * q);
        const f32x4 a1 = *reinterpret_cast<const f32x4*>(dp_s + (16 + j) * LD + nb + 16 * s + 4 * q);
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            acc0 = __builtin_amdgcn_mfma_f32_16x16x4f32(a0[e], wv[s][e], acc0, 0, 0, 0);
            acc1 = __builtin_amdgcn_mfma_f32_16x16x4f32(a1[e], wv[s][e], acc1, 0, 0, 0);
        }
    }
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        red[((wave * 2 + 0) * 4 + r) * 64 + lane] = acc0[r];
        red[((wave * 2 + 1) * 4 + r) * 64 + lane] = acc1[r];
    }
    __syncthreads();
    {
        const int t = tid >> 8, r = (tid >> 6) & 3, l = tid & 63;
        float sum = red[((0 * 2 + t) * 4 + r) * 64 + l];
#pragma unroll
        for (int w = 1; w < 8; ++w) sum += red[((w * 2 + t) * 4 + r) * 64 + l];
        const int row = 16 * t + 4 * (l >> 4) + r, col = k0 + (l & 15);
        if (Ynext) {
            float d = row < nrows ? sum : 0.f;
            if (act_next == PV_ACT_ELU) d *= pv_elu_grad_from_out(Ynext[row * K + col]);
            sum = d;
        }
        dIn[row * K + col] = sum;
    }
}

// dpre = dY * act'(Y) (rows >= nrows zeroed) written out, plus bias gradient / Adam.  Used for the wide
// first encoder layer whose weight gradient is formed by pvk_enc_bwd.
template <int MODE>
__device__ __forceinline__ void pv_bias_bwd_block(const float* __restrict__ Y, const float* __restrict__ dY, int N, int act,
                                                  float* __restrict__ dpre, float* __restrict__ bias, float* __restrict__ mB,
                                                  float* __restrict__ vB, const PvStep* st, const PvAdam& adam,
                                                  const float* __restrict__ gB, int block) {
    const int n = block * blockDim.x + threadIdx.x;
    if (n >= N) return;
    const int nrows = st->nrows;
    float g = 0.f;
    for (int b = 0; b < 32; ++b) {
        float d = b < nrows ? dY[b * N + n] : 0.f;
        if (act == PV_ACT_ELU) d *= pv_elu_grad_from_out(Y[b * N + n]);
        dpre[b * N + n] = d;
        g += d;
    }
    if (MODE == PV_MODE_GRAD) {
        mB[n] = st->acc ? mB[n] + g : g;
    } else {
        float w = bias[n], m = mB[n], v = vB[n];
        if (MODE == PV_MODE_TRAIN_ACC) g += gB[n];
        pv_adam(w, m, v, g, st->alpha, adam);
        bias[n] = w;
        mB[n] = m;
        vB[n] = v;
    }
}

template <int MODE>
__global__ void pvk_bias_bwd(const float* __restrict__ Y, const float* __restrict__ dY, int N, int act,
                             float* __restrict__ dpre, float* __restrict__ bias, float* __restrict__ mB,
                             float* __restrict__ vB, const PvStep* st, PvAdam adam, const float* __restrict__ gB) {
    pv_bias_bwd_block<MODE>(Y, dY, N, act, dpre, bias, mB, vB, st, adam, gB, blockIdx.x);
}

// ---------------------------------------------------------------------------------------------
// Reparameterisation + KL (popvae.py:384-389, 419-421).  heads[32][ldh]: cols [0,K) = z_mean,
// [K,2K) = z_log_var.  z[32][ldz] (padding columns zero).  kl[b] = -0.5*sum(1 + s - mu^2 - e^s).
// ---------------------------------------------------------------------------------------------
__global__ void pvk_reparam(const float* __restrict__ heads, int ldh, int K, const float* __restrict__ eps,
                            float* __restrict__ z, int ldz, float* __restrict__ kl) {
    const int b = threadIdx.x;
    if (b >= 32) return;
    float acc = 0.f;
    for (int k = 0; k < ldz; ++k) {
        float zz = 0.f;
        if (k < K) {
            const float mu = heads[b * ldh + k], s = heads[b * ldh + K + k];
            const float es = expf(s);
            zz = mu + es * eps[b * K + k];  // NOTE exp(s), not exp(s/2): popvae.py:389
            acc += 1.f + s - mu * mu - es;
        }
        z[b * ldz + k] = zz;
    }
    kl[b] = -0.5f * acc;
}

// d heads from dz: dmu = dz + mu/B ; ds = dz*eps*e^s + 0.5*(e^s - 1)/B ; rows >= nrows -> 0
__global__ void pvk_reparam_bwd(const float* __restrict__ heads, int ldh, int K, const float* __restrict__ eps,
                                const float* __restrict__ dz, int ldz, float* __restrict__ dheads,
                                const PvStep* st) {
    const int b = threadIdx.x;
    if (b >= 32) return;
    const bool valid = b < st->nrows;
    const float ib = st->inv_b;
    for (int k = 0; k < ldh; ++k) dheads[b * ldh + k] = 0.f;
    if (!valid) return;
    for (int k = 0; k < K; ++k) {
        const float mu = heads[b * ldh + k], s = heads[b * ldh + K + k];
        const float es = expf(s), d = dz[b * ldz + k];
        dheads[b * ldh + k] = d + mu * ib;
        dheads[b * ldh + K + k] = d * eps[b * K + k] * es + 0.5f * (es - 1.f) * ib;
    }
}

// ---------------------------------------------------------------------------------------------
// Wide decoder, fused: logits -> sigmoid -> BCE (+ loss) -> dlogits -> dV (+Adam in place), dbias
// (+Adam), dg partials.  One wave per subtile (NGRP == 1: the whole width in one wave).
// LDS (dynamic): g_s [32][ldg] shared; per wave: vt [2][32][33] (V^T staging) and dt [32][33] (da^T).
// Outputs per workgroup: dgpart[wg][32][Wp] and losspart[wg] (summed BCE of this workgroup's SNPs).
// TRAIN: 0 = evaluation (loss only), 1 = training.
// ---------------------------------------------------------------------------------------------
template <int NWB, int TRAIN, int MODE>
__global__ __launch_bounds__(256, 2) void pvk_dec(const float* __restrict__ V, const float* __restrict__ mV,
                                                  const float* __restrict__ vV, float* __restrict__ V2,
                                                  float* __restrict__ mV2, float* __restrict__ vV2,
                                                  float* __restrict__ cb, float* __restrict__ mC, float* __restrict__ vC,
                                                  const float* __restrict__ g, const uint64_t* __restrict__ Xt,
                                                  int nblk, int L, const PvStep* st, PvAdam adam,
                                                  float* __restrict__ dgpart, float* __restrict__ losspart, int ts,
                                                  const float* __restrict__ gV, const float* __restrict__ gC) {
    constexpr int WP = NWB * 32;
    constexpr int LDG = WP + 1;
    extern __shared__ __attribute__((aligned(16))) float smem[];
    float* g_s = smem;                                   // [32][LDG]
    float* wave_s = smem + 32 * LDG + (threadIdx.x >> 6) * (3 * 32 * 33);
    float* vt = wave_s;                                  // [2][32][33]
    float* dt = wave_s + 2 * 32 * 33;                    // [32][33]
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63, c = lane & 31, h = lane >> 5;
    const int gw = blockIdx.x * 4 + wave, nwaves = gridDim.x * 4;
    const int nrows = st->nrows;
    const float inv_b = st->inv_b, alpha = st->alpha;
    const int bce_probs = PV_BCE_MODE(st);
    const bool acc_grads = MODE == PV_MODE_GRAD && st->acc != 0;

    PV_WT(0, 0);
    for (int i = tid; i < 32 * WP; i += 256) g_s[(i / WP) * LDG + (i % WP)] = g[i];
    __syncthreads();
    PV_WT(0, 1);

    f32x16 dg[NWB];
    float lossacc = 0.f;  // only the batch total enters the loss (popvae.py:423)
    if (TRAIN) {
#pragma unroll
        for (int w = 0; w < NWB; ++w)
#pragma unroll
            for (int r = 0; r < 16; ++r) dg[w][r] = 0.f;
    }

    // round-robin block assignment: at any instant the chip's waves sweep one contiguous window of HBM, which
    // sustains ~6-8 % more bandwidth than every wave streaming its own private chunk (measured, stream_bench too)
    for (int blk = gw; blk < nblk; blk += nwaves) {
        // keep the (loop-invariant) g operand reads inside the loop: hoisting them costs 128 VGPRs
        asm volatile("" ::: "memory");
        // ts = tile stride in f32x4 units: the W / m / v tiles of a subtile are interleaved in one buffer
        // ([W 16K | m 16K | v 16K] per subtile, ts = 3 * NWB * 256), so a step reads ONE window and writes ONE
        const int64_t tbase = (int64_t)blk * ts;
        const int64_t gbase = (int64_t)blk * (NWB * 256);  // gradient dumps (MODE_GRAD) are plain tiles
        const f32x4* tV = reinterpret_cast<const f32x4*>(V) + tbase;
        f32x4 vv[NWB][4];
#pragma unroll
        for (int w = 0; w < NWB; ++w)
#pragma unroll
            for (int q = 0; q < 4; ++q) vv[w][q] = __builtin_nontemporal_load(tV + (w * 4 + q) * 64 + lane);
        const int l = blk * 32 + c;
        const bool lvalid = l < L;
        const float bias = cb[l];
        uint64_t xw = Xt[l] >> (8 * h);
        const uint32_t lo = (uint32_t)xw, hi = (uint32_t)(xw >> 32);

        // (1) logits a[b][l] : M = b, N = l, K = unit.  A = g[b][unit], B = V fragment register.
        f32x16 a;
#pragma unroll
        for (int r = 0; r < 16; ++r) a[r] = 0.f;
#pragma unroll
        for (int w = 0; w < NWB; ++w)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const float ga = g_s[c * LDG + w * 32 + PV_ROW(r, h)];
                a = __builtin_amdgcn_mfma_f32_32x32x2f32(ga, vv[w][r >> 2][r & 3], a, 0, 0, 0);
            }
        // (2) sigmoid / BCE-from-logits / dlogits ; reg r <-> sample b = ROW(r,h), lane <-> SNP l
        float dbias = 0.f;
        pv_sigmoid_bce(bce_probs, (blk * 32 + 32 <= L) && nrows >= 32, true, a, bias, lo, hi, h, 0, nrows, lvalid, inv_b, lossacc,
                       dbias);
        if (TRAIN) {
            // bias: sum over the 32 samples = 16 regs + the other half-wave
            dbias += __shfl_xor(dbias, 32);
            if (h == 0 && lvalid) {
                if (MODE == PV_MODE_GRAD) {
                    mC[l] = acc_grads ? mC[l] + dbias : dbias;
                } else {
                    float w_ = bias, m_ = mC[l], v_ = vC[l];
                    if (MODE == PV_MODE_TRAIN_ACC) dbias += gC[l];
                    pv_adam(w_, m_, v_, dbias, alpha, adam);
                    cb[l] = w_;
                    mC[l] = m_;
                    vC[l] = v_;
                }
            }
            // da^T into LDS for product (3): dt[b][l_loc]
#pragma unroll
            for (int r = 0; r < 16; ++r) dt[PV_ROW(r, h) * 33 + c] = a[r];
            // (3) dg[b][unit] += sum_l da[b][l] * V[unit][l] : M = b, N = unit, K = l
            //     B operand needs V^T (unit on the lane): stage each 32x32 block through LDS.
#pragma unroll
            for (int w = 0; w < NWB; ++w) {
                float* vts = vt + (w & 1) * (32 * 33);
#pragma unroll
                for (int r = 0; r < 16; ++r) vts[PV_ROW(r, h) * 33 + c] = vv[w][r >> 2][r & 3];  // [unit_loc][l_loc]
                __builtin_amdgcn_s_waitcnt(0xc07f);
                __builtin_amdgcn_wave_barrier();
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const float av = dt[c * 33 + PV_ROW(r, h)];   // A[i=b][k=l]
                    const float bv = vts[c * 33 + PV_ROW(r, h)];  // B[k=l][j=unit_loc]
                    dg[w] = __builtin_amdgcn_mfma_f32_32x32x2f32(av, bv, dg[w], 0, 0, 0);
                }
                __builtin_amdgcn_sched_barrier(0);
            }
            // (4) dV[unit][l] = sum_b g[b][unit] * da[b][l] : M = unit, N = l, K = b ; then Adam
            const f32x4* tM = reinterpret_cast<const f32x4*>(mV) + tbase;
            const f32x4* tVv = reinterpret_cast<const f32x4*>(vV) + tbase;
            f32x4* dV2 = reinterpret_cast<f32x4*>(V2) + tbase;
            f32x4* dM2 = reinterpret_cast<f32x4*>(mV2) + (MODE == PV_MODE_GRAD ? gbase : tbase);
            f32x4* dVv2 = reinterpret_cast<f32x4*>(vV2) + tbase;
#pragma unroll
            for (int w = 0; w < NWB; ++w) {
                f32x16 dv;
#pragma unroll
                for (int r = 0; r < 16; ++r) dv[r] = 0.f;
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const float ga = g_s[PV_ROW(r, h) * LDG + w * 32 + c];  // A[i=unit_loc][k=b]
                    dv = __builtin_amdgcn_mfma_f32_32x32x2f32(ga, a[r], dv, 0, 0, 0);
                }
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    const int off = (w * 4 + q) * 64 + lane;
                    if (MODE == PV_MODE_GRAD) {
                        f32x4 gq = {dv[q * 4 + 0], dv[q * 4 + 1], dv[q * 4 + 2], dv[q * 4 + 3]};
                        if (acc_grads) gq += __builtin_nontemporal_load(dM2 + off);
                        __builtin_nontemporal_store(gq, dM2 + off);
                    } else {
                        f32x4 wq = vv[w][q], mq = __builtin_nontemporal_load(tM + off), vq = __builtin_nontemporal_load(tVv + off);
                        f32x4 ga = {0.f, 0.f, 0.f, 0.f};
                        if (MODE == PV_MODE_TRAIN_ACC) ga = __builtin_nontemporal_load(reinterpret_cast<const f32x4*>(gV) + gbase + off);
#pragma unroll
                        for (int e = 0; e < 4; ++e) {
                            float w_ = wq[e], m_ = mq[e], v_ = vq[e];
                            pv_adam(w_, m_, v_, MODE == PV_MODE_TRAIN_ACC ? dv[q * 4 + e] + ga[e] : dv[q * 4 + e], alpha, adam);
                            wq[e] = w_;
                            mq[e] = m_;
                            vq[e] = v_;
                        }
                        __builtin_nontemporal_store(wq, dV2 + off);
                        __builtin_nontemporal_store(mq, dM2 + off);
                        __builtin_nontemporal_store(vq, dVv2 + off);
                    }
                }
                __builtin_amdgcn_sched_barrier(0);
            }
        }
    }
    // ---- per-workgroup reductions (fixed order) ----
    __syncthreads();  // everyone done with g_s / wave staging
    PV_WT(0, 2);
    float* red = smem;  // reuse: [4][NWB*16][64] for dg, then [4][32] loss
    // loss: fixed-order butterfly over the wave, then over the 4 waves
    lossacc += __shfl_xor(lossacc, 1);
    lossacc += __shfl_xor(lossacc, 2);
    lossacc += __shfl_xor(lossacc, 4);
    lossacc += __shfl_xor(lossacc, 8);
    lossacc += __shfl_xor(lossacc, 16);
    lossacc += __shfl_xor(lossacc, 32);
    float* lred = TRAIN ? red + 4 * NWB * 1024 : red;  // evaluation launches bring only the g tile's LDS (lds_dec_eval)
    if (lane == 0) lred[wave] = lossacc;
    if (TRAIN) {
#pragma unroll
        for (int w = 0; w < NWB; ++w)
#pragma unroll
            for (int r = 0; r < 16; ++r) red[((wave * NWB + w) * 16 + r) * 64 + lane] = dg[w][r];
    }
    __syncthreads();
    if (tid == 0) losspart[blockIdx.x] = ((lred[0] + lred[1]) + lred[2]) + lred[3];
    if (TRAIN) {
        for (int e = tid; e < NWB * 1024; e += 256) {
            const float s = ((red[e] + red[NWB * 1024 + e]) + red[2 * NWB * 1024 + e]) + red[3 * NWB * 1024 + e];
            const int w = e >> 10, r = (e >> 6) & 15, ln = e & 63;
            dgpart[(int64_t)blockIdx.x * (32 * WP) + PV_ROW(r, ln >> 5) * WP + w * 32 + (ln & 31)] = s;
        }
    }
    PV_WT(0, 3);
}

// ---------------------------------------------------------------------------------------------
// Wide decoder for widths above 128 (NGRP = Wp / 128 groups of 128 units, NGRP in {2, 4}).
// Workgroup = NW waves; wave = (block bsub = wave / NGRP, group grp = wave % NGRP), i.e. NW / NGRP SNP blocks
// per iteration.  NW = 8 at width 512 (one workgroup per CU); NW = 4 at width 256, where two workgroups fit a CU's LDS and,
// not sharing barriers, overlap each other's loads and compute.  Each wave forms the partial logits of its group; the partials meet in LDS (the wave's own
// V^T staging area doubles as the mailbox) and every wave of a block sums them in group order, so all of
// them hold bit-identical logits.  From there on each wave treats its group as pvk_dec does: dg partial,
// dV + Adam on its own 32x128 subtile, which it still holds in registers -- no weight byte is read twice.
// ---------------------------------------------------------------------------------------------
// Sync point among the waves of ONE SNP block inside pvk_dec_grp (they share an LDS counter that only grows): arrive, then
// wait until `target` arrivals.  All waves of a workgroup are resident, so the wait always ends; the spin is bounded all the
// same (returns false when it gives up -- the caller then poisons its loss, loudly wrong instead of hung).
__device__ __forceinline__ bool pv_block_sync(int* cnt, int target, int lane) {
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");  // this wave's LDS writes are done before it is counted
    __builtin_amdgcn_wave_barrier();
    if (lane == 0) __hip_atomic_fetch_add(cnt, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
    bool ok = false;
    for (int spin = 0; spin < (1 << 22); ++spin) {
        if (__hip_atomic_load(cnt, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP) >= target) { ok = true; break; }
        __builtin_amdgcn_s_sleep(1);
    }
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
    return ok;
}

template <int NGRP, int TRAIN, int MODE, int NW>
__global__ __launch_bounds__(NW * 64, 8 / NW) void pvk_dec_grp(const float* __restrict__ V, const float* __restrict__ mV,
                                                   const float* __restrict__ vV, float* __restrict__ V2,
                                                   float* __restrict__ mV2, float* __restrict__ vV2,
                                                   float* __restrict__ cb, float* __restrict__ mC, float* __restrict__ vC,
                                                   const float* __restrict__ g, const uint64_t* __restrict__ Xt,
                                                   int nblk, int L, const PvStep* st, PvAdam adam,
                                                   float* __restrict__ dgpart, float* __restrict__ losspart, int ts,
                                                   const float* __restrict__ gV, const float* __restrict__ gC) {
    constexpr int NWB = 4;
    constexpr int WP = NGRP * 128;
    constexpr int LDG = WP + 1;
    constexpr int BPI = NW / NGRP;  // SNP blocks per workgroup iteration (NW waves per workgroup)
    extern __shared__ __attribute__((aligned(16))) float smem[];
    float* g_s = smem;  // [32][LDG]
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63, c = lane & 31, h = lane >> 5;
    float* vt = smem + 32 * LDG + wave * (2 * 32 * 33);  // [32][33]; also this wave's partial-logit mailbox
    float* dt = vt + 32 * 33;                            // [32][33]
    const int grp = wave % NGRP, bsub = wave / NGRP;
    const int bgroups = (nblk + BPI - 1) / BPI;
    const int per = (bgroups + gridDim.x - 1) / gridDim.x;
    const int nrows = st->nrows;
    const float inv_b = st->inv_b, alpha = st->alpha;
    const int bce_probs = PV_BCE_MODE(st);
    const bool acc_grads = MODE == PV_MODE_GRAD && st->acc != 0;

    // per SNP block of the workgroup: [0] partial logits posted, [1] partial logits read (both only ever grow)
    __shared__ int blk_sync[BPI][2];
    if (tid < 2 * BPI) blk_sync[tid >> 1][tid & 1] = 0;
    bool sync_ok = true;
    for (int i = tid; i < 32 * WP; i += NW * 64) g_s[(i / WP) * LDG + (i % WP)] = g[i];
    __syncthreads();
    const float* gg = g_s + grp * 128;  // this wave's 128 units

    f32x16 dg[NWB];
    float lossacc = 0.f;
    if (TRAIN) {
#pragma unroll
        for (int w = 0; w < NWB; ++w)
#pragma unroll
            for (int r = 0; r < 16; ++r) dg[w][r] = 0.f;
    }

    int nsync = 0;
    for (int it = 0; it < per; ++it) {
        asm volatile("" ::: "memory");
        const int bg = it * (int)gridDim.x + blockIdx.x;  // round-robin over workgroups
        const int blk = bg * BPI + bsub;
        const bool valid = bg < bgroups && blk < nblk;  // the same for all waves of a block: they skip the iteration together
        if (!valid) continue;
        const int64_t tbase = ((int64_t)blk * NGRP + grp) * ts;            // interleaved W|m|v tiles
        const int64_t gbase = ((int64_t)blk * NGRP + grp) * (NWB * 256);  // plain tiles (gradient dump)
        const f32x4* tV = reinterpret_cast<const f32x4*>(V) + tbase;
        f32x4 vv[NWB][4];
        f32x16 a;
#pragma unroll
        for (int r = 0; r < 16; ++r) a[r] = 0.f;
#pragma unroll
        for (int w = 0; w < NWB; ++w)
#pragma unroll
            for (int q = 0; q < 4; ++q) vv[w][q] = __builtin_nontemporal_load(tV + (w * 4 + q) * 64 + lane);
        // (1) partial logits over this group's units
#pragma unroll
        for (int w = 0; w < NWB; ++w)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const float ga = gg[c * LDG + w * 32 + PV_ROW(r, h)];
                a = __builtin_amdgcn_mfma_f32_32x32x2f32(ga, vv[w][r >> 2][r & 3], a, 0, 0, 0);
            }
#pragma unroll
        for (int r = 0; r < 16; ++r) vt[r * 64 + lane] = a[r];
        // the NGRP waves of this SNP block meet here; the other blocks of the workgroup run on (a workgroup barrier at this
        // point kept all waves of a CU in the same phase: loads and compute did not overlap)
        ++nsync;
        sync_ok &= pv_block_sync(&blk_sync[bsub][0], NGRP * nsync, lane);
        {
            const float* mb = smem + 32 * LDG + (bsub * NGRP) * (2 * 32 * 33);
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                float sum = mb[r * 64 + lane];
#pragma unroll
                for (int g2 = 1; g2 < NGRP; ++g2) sum += mb[g2 * (2 * 32 * 33) + r * 64 + lane];
                a[r] = sum;
            }
        }
        // mailboxes double as the V^T staging tiles of product (3): nobody overwrites one before all have read it
        sync_ok &= pv_block_sync(&blk_sync[bsub][1], NGRP * nsync, lane);
        const int l = blk * 32 + c;
        const bool lvalid = l < L;
        const float bias = cb[l];
        uint64_t xw = Xt[l] >> (8 * h);
        const uint32_t lo = (uint32_t)xw, hi = (uint32_t)(xw >> 32);
        // (2) sigmoid / BCE-from-logits / dlogits (every wave of the block computes the same values)
        float dbias = 0.f;
        // (not pv_sigmoid_bce: its straight-line form holds more values in flight, and this kernel is at its register limit --
        // 12 more spilled registers, +3 % at width 512)
        auto sigmoid_bce = [&](auto Pc) {
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int b = PV_ROW(r, h);
                const float x = 0.5f * pv_code(lo, hi, r);
                const float z = a[r] + bias;
                float bce, dz;
                pv_bce<decltype(Pc)::value>(z, x, bce, dz);
                const bool ok = lvalid && (b < nrows);
                lossacc += (ok && grp == 0) ? bce : 0.f;
                const float d = ok ? dz * inv_b : 0.f;
                a[r] = d;
                dbias += d;
            }
        };
        if (bce_probs) sigmoid_bce(std::integral_constant<int, 1>{});
        else sigmoid_bce(std::integral_constant<int, 0>{});
        if (TRAIN) {
            dbias += __shfl_xor(dbias, 32);
            if (grp == 0 && h == 0 && lvalid) {
                if (MODE == PV_MODE_GRAD) {
                    mC[l] = acc_grads ? mC[l] + dbias : dbias;
                } else {
                    float w_ = bias, m_ = mC[l], v_ = vC[l];
                    if (MODE == PV_MODE_TRAIN_ACC) dbias += gC[l];
                    pv_adam(w_, m_, v_, dbias, alpha, adam);
                    cb[l] = w_;
                    mC[l] = m_;
                    vC[l] = v_;
                }
            }
#pragma unroll
            for (int r = 0; r < 16; ++r) dt[PV_ROW(r, h) * 33 + c] = a[r];
            // (3) dg[b][unit] += sum_l da[b][l] * V[unit][l]
#pragma unroll
            for (int w = 0; w < NWB; ++w) {
#pragma unroll
                for (int r = 0; r < 16; ++r) vt[PV_ROW(r, h) * 33 + c] = vv[w][r >> 2][r & 3];
                __builtin_amdgcn_s_waitcnt(0xc07f);
                __builtin_amdgcn_wave_barrier();
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const float av = dt[c * 33 + PV_ROW(r, h)];
                    const float bv = vt[c * 33 + PV_ROW(r, h)];
                    dg[w] = __builtin_amdgcn_mfma_f32_32x32x2f32(av, bv, dg[w], 0, 0, 0);
                }
                __builtin_amdgcn_s_waitcnt(0xc07f);
                __builtin_amdgcn_wave_barrier();
                __builtin_amdgcn_sched_barrier(0);
            }
            // (4) dV[unit][l] = sum_b g[b][unit] * da[b][l], then Adam
            const f32x4* tM = reinterpret_cast<const f32x4*>(mV) + tbase;
            const f32x4* tVv = reinterpret_cast<const f32x4*>(vV) + tbase;
            f32x4* dV2 = reinterpret_cast<f32x4*>(V2) + tbase;
            f32x4* dM2 = reinterpret_cast<f32x4*>(mV2) + (MODE == PV_MODE_GRAD ? gbase : tbase);
            f32x4* dVv2 = reinterpret_cast<f32x4*>(vV2) + tbase;
#pragma unroll
            for (int w = 0; w < NWB; ++w) {
                f32x16 dv;
#pragma unroll
                for (int r = 0; r < 16; ++r) dv[r] = 0.f;
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const float ga = gg[PV_ROW(r, h) * LDG + w * 32 + c];
                    dv = __builtin_amdgcn_mfma_f32_32x32x2f32(ga, a[r], dv, 0, 0, 0);
                }
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    const int off = (w * 4 + q) * 64 + lane;
                    if (MODE == PV_MODE_GRAD) {
                        f32x4 gq = {dv[q * 4 + 0], dv[q * 4 + 1], dv[q * 4 + 2], dv[q * 4 + 3]};
                        if (acc_grads) gq += __builtin_nontemporal_load(dM2 + off);
                        __builtin_nontemporal_store(gq, dM2 + off);
                    } else {
                        f32x4 wq = vv[w][q], mq = __builtin_nontemporal_load(tM + off), vq = __builtin_nontemporal_load(tVv + off);
                        f32x4 ga = {0.f, 0.f, 0.f, 0.f};
                        if (MODE == PV_MODE_TRAIN_ACC) ga = __builtin_nontemporal_load(reinterpret_cast<const f32x4*>(gV) + gbase + off);
#pragma unroll
                        for (int e = 0; e < 4; ++e) {
                            float w_ = wq[e], m_ = mq[e], v_ = vq[e];
                            pv_adam(w_, m_, v_, MODE == PV_MODE_TRAIN_ACC ? dv[q * 4 + e] + ga[e] : dv[q * 4 + e], alpha, adam);
                            wq[e] = w_;
                            mq[e] = m_;
                            vq[e] = v_;
                        }
                        __builtin_nontemporal_store(wq, dV2 + off);
                        __builtin_nontemporal_store(mq, dM2 + off);
                        __builtin_nontemporal_store(vq, dVv2 + off);
                    }
                }
                __builtin_amdgcn_sched_barrier(0);
            }
        }
    }
    // ---- per-workgroup reductions (fixed order) ----
    __syncthreads();
    float* red = smem;  // [NW waves][NWB*16][64], then [NW] loss
    lossacc += __shfl_xor(lossacc, 1);
    lossacc += __shfl_xor(lossacc, 2);
    lossacc += __shfl_xor(lossacc, 4);
    lossacc += __shfl_xor(lossacc, 8);
    lossacc += __shfl_xor(lossacc, 16);
    lossacc += __shfl_xor(lossacc, 32);
    float* lred = red + NW * NWB * 1024;
    if (!sync_ok) lossacc = __builtin_nanf("");  // a sync point gave up: make the step loudly wrong
    if (lane == 0) lred[wave] = lossacc;
    if (TRAIN) {
#pragma unroll
        for (int w = 0; w < NWB; ++w)
#pragma unroll
            for (int r = 0; r < 16; ++r) red[((wave * NWB + w) * 16 + r) * 64 + lane] = dg[w][r];
    }
    __syncthreads();
    if (tid == 0) {
        float sacc = 0.f;
        for (int w2 = 0; w2 < NW; ++w2) sacc += lred[w2];
        losspart[blockIdx.x] = sacc;
    }
    if (TRAIN) {
        for (int e = tid; e < NGRP * NWB * 1024; e += NW * 64) {
            const int gq = e / (NWB * 1024), f = e % (NWB * 1024);
            float sum = red[(0 * NGRP + gq) * (NWB * 1024) + f];
#pragma unroll
            for (int b2 = 1; b2 < BPI; ++b2) sum += red[(b2 * NGRP + gq) * (NWB * 1024) + f];
            const int w = f >> 10, r = (f >> 6) & 15, ln = f & 63;
            dgpart[(int64_t)blockIdx.x * (32 * WP) + PV_ROW(r, ln >> 5) * WP + gq * 128 + w * 32 + (ln & 31)] = sum;
        }
    }
}

// ---------------------------------------------------------------------------------------------
// Wide decoder, training pass for widths up to 128 (one unit group), block-column split: the NWB waves of one SNP block
// each hold ONE 32-unit block column of the subtile (4 KiB of V | m | v each) instead of one wave holding all of it.
//   * a wave's whole working set is ~110 registers, so everything it needs is requested a phase ahead: the Adam moments of
//     this block at the top of the visit (they land during products (1)-(3)), the V block of the NEXT visit as well.  The
//     one-wave-per-subtile kernel (pvk_dec, 248 registers) issued 2 KiB of moment loads, waited, stored, 16 times per
//     subtile; a wave took 24-32 us per subtile of which ~9 us were arithmetic (tools/trace_wide.py);
//   * the unit of work of a workgroup is BPI = NW / NWB subtiles per iteration (1 at width 128): a small problem or SNP
//     shard spreads evenly over the CUs;
//   * the partial logits of the NWB waves meet in LDS (the wave's V^T staging tile doubles as its mailbox) and every wave
//     sums them in block-column order, so all hold bit-identical logits; the sigmoid / BCE / dlogits arithmetic is then
//     done by each of them (16 values per lane), which is cheaper than a second exchange;
//   * each wave owns a block column of dg: the workgroup's partial tile needs no cross-wave sum when BPI == 1.
// Waves of ONE SNP block meet at two LDS counters that only grow (pv_block_sync).  LDS: g tile + per wave 2 x [32][33].
// ---------------------------------------------------------------------------------------------
template <int NWB>
struct PvDec2 {
    static constexpr int NW = NWB == 3 ? 3 : 4;  // waves per workgroup
    static constexpr int BPI = NW / NWB;         // SNP blocks per workgroup iteration
    static constexpr int WP = NWB * 32, LDG = WP + 1;
    static constexpr size_t lds() { return (size_t)(32 * LDG + NW * 2 * 32 * 33) * 4; }
};
// PART: 0 = the whole pass.  The pass in two launches (SNP shards, where the serial narrow chain is a third of the step):
//   1 = loss + dg only: products (1)-(3), reads V, the batch words and the bias -- 4 bytes per parameter; its dg partial
//       tiles feed the backward chain at once;
//   2 = dV + Adam only: products (1), (2), (4), the full read and write of V | m | v; it needs nothing the backward chain
//       produces, so it runs on a second stream beside the chain and the reduction.
// Both do the same arithmetic in the same order as the whole pass: the step is bit-identical either way.
template <int NWB, int PART = 0>
__global__ __launch_bounds__(PvDec2<NWB>::NW * 64, 2) void pvk_dec2(const float* __restrict__ V, const float* __restrict__ mV,
                                                                    const float* __restrict__ vV, float* __restrict__ V2,
                                                                    float* __restrict__ mV2, float* __restrict__ vV2,
                                                                    float* __restrict__ cb, float* __restrict__ mC,
                                                                    float* __restrict__ vC, const float* __restrict__ g,
                                                                    const uint64_t* __restrict__ Xt, int nblk, int L,
                                                                    const PvStep* st, PvAdam adam, float* __restrict__ dgpart,
                                                                    float* __restrict__ losspart, int ts) {
    using G = PvDec2<NWB>;
    constexpr int NW = G::NW, BPI = G::BPI, WP = G::WP, LDG = G::LDG;
    extern __shared__ __attribute__((aligned(16))) float smem[];
    float* g_s = smem;  // [32][LDG]
    const int tid = threadIdx.x, wave = __builtin_amdgcn_readfirstlane(tid >> 6), lane = tid & 63, c = lane & 31, h = lane >> 5;
    float* vt = smem + 32 * LDG + wave * (2 * 32 * 33);  // [32][33]; also this wave's partial-logit mailbox ([4][64][4])
    float* dt = vt + 32 * 33;                            // [32][33]
    const int bsub = wave / NWB, w = wave - bsub * NWB;
    const float* mb0 = smem + 32 * LDG + (bsub * NWB) * (2 * 32 * 33);  // mailbox of block column 0 of this SNP block
    const int nrows = st->nrows;
    const float inv_b = st->inv_b, alpha = st->alpha;
    const int bce_probs = PV_BCE_MODE(st);
    __shared__ int blk_sync[BPI][2];  // [0] partial logits posted, [1] partial logits read (both only ever grow)
    if (tid < 2 * BPI) blk_sync[tid >> 1][tid & 1] = 0;
    bool sync_ok = true;
    int nsync = 0;
    PV_WT(0, 0);

    // visit `it` of this wave: SNP block (it * gridDim.x + blockIdx.x) * BPI + bsub, block column w
    auto block_of = [&](int it) { return (it * (int)gridDim.x + (int)blockIdx.x) * BPI + bsub; };
    auto tile = [&](int blk_) { return (int64_t)blk_ * ts + w * 256 + lane; };
    f32x4 vv[4], vn[4];
    uint64_t xw = 0, xw_n = 0;
    float bias = 0.f, bias_n = 0.f;
    int blk = block_of(0);
    if (blk < nblk) {  // the first block travels while g is staged
        const f32x4* tV = reinterpret_cast<const f32x4*>(V) + tile(blk);
#pragma unroll
        for (int q = 0; q < 4; ++q) vv[q] = __builtin_nontemporal_load(tV + q * 64);
        xw = Xt[blk * 32 + c];
        bias = cb[blk * 32 + c];
    }
    for (int i = tid; i < 32 * WP; i += NW * 64) g_s[(i / WP) * LDG + (i % WP)] = g[i];
    __syncthreads();
    PV_WT(0, 1);
    // this wave's 32 units of g, as the A operands of products (1) and (4): the same for every visit
    float g1[16], g4[16];
#pragma unroll
    for (int r = 0; r < 16; ++r) {
        g1[r] = g_s[c * LDG + w * 32 + PV_ROW(r, h)];    // A[i=b][k=unit]
        g4[r] = g_s[PV_ROW(r, h) * LDG + w * 32 + c];    // A[i=unit_loc][k=b]
    }

    f32x16 dg;
#pragma unroll
    for (int r = 0; r < 16; ++r) dg[r] = 0.f;
    float lossacc = 0.f;  // only the batch total enters the loss (popvae.py:423)

    for (int it = 0; blk < nblk; ++it) {
        const int nb = block_of(it + 1) < nblk ? block_of(it + 1) : blk;  // past the end: a harmless re-read
        const int l = blk * 32 + c;
        const bool lvalid = l < L;
        // requests of this visit: the genotype / bias words and V of the next visit, the Adam moments of this block.  The
        // bias word goes FIRST: loads return in order, so it has landed before product (1) of the next visit can start,
        // i.e. before that visit's sync points, behind which block column 0 rewrites the bias in place.
        const int64_t tb = tile(blk);
        f32x4 mq[4], vq[4];
        float cm = 0.f, cv = 0.f;
        {
            bias_n = cb[nb * 32 + c];
            xw_n = Xt[nb * 32 + c];
            if (PART != 1 && w == 0 && h == 0 && lvalid) {
                cm = mC[l];
                cv = vC[l];
            }
#ifndef PV_DEC2_VAR
#define PV_DEC2_VAR 0
#endif
            const f32x4* tVn = reinterpret_cast<const f32x4*>(V) + tile(nb);
            if (PV_DEC2_VAR != 2) {
#pragma unroll
                for (int q = 0; q < 4; ++q) vn[q] = __builtin_nontemporal_load(tVn + q * 64);
            }
            if (PV_DEC2_VAR != 1 && PART != 1) {
                const f32x4* tM = reinterpret_cast<const f32x4*>(mV) + tb;
                const f32x4* tVv = reinterpret_cast<const f32x4*>(vV) + tb;
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    mq[q] = __builtin_nontemporal_load(tM + q * 64);
                    vq[q] = __builtin_nontemporal_load(tVv + q * 64);
                }
            }
        }
        const uint64_t xs = xw >> (8 * h);
        const uint32_t lo = (uint32_t)xs, hi = (uint32_t)(xs >> 32);

        // (1) logits a[b][l] over this wave's 32 units: M = b, N = l, K = unit.  A = g[b][unit], B = V fragment register.
        f32x16 a;
#pragma unroll
        for (int r = 0; r < 16; ++r) a[r] = 0.f;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            a = __builtin_amdgcn_mfma_f32_32x32x2f32(g1[r], vv[r >> 2][r & 3], a, 0, 0, 0);
        }
        if (NWB > 1) {
            // the NWB waves of this SNP block meet; the other blocks of the workgroup run on
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const f32x4 t4 = {a[4 * q], a[4 * q + 1], a[4 * q + 2], a[4 * q + 3]};
                reinterpret_cast<f32x4*>(vt)[q * 64 + lane] = t4;
            }
            ++nsync;
            sync_ok &= pv_block_sync(&blk_sync[bsub][0], NWB * nsync, lane);
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                f32x4 sum = reinterpret_cast<const f32x4*>(mb0)[q * 64 + lane];
#pragma unroll
                for (int w2 = 1; w2 < NWB; ++w2) sum += reinterpret_cast<const f32x4*>(mb0 + w2 * (2 * 32 * 33))[q * 64 + lane];
                a[4 * q] = sum[0]; a[4 * q + 1] = sum[1]; a[4 * q + 2] = sum[2]; a[4 * q + 3] = sum[3];
            }
            // mailboxes double as the V^T staging tiles of product (3): nobody overwrites one before all have read it
            sync_ok &= pv_block_sync(&blk_sync[bsub][1], NWB * nsync, lane);
        }
        if (PV_DEC2_VAR == 1 && PART != 1) {  // experiment: the moments requested behind product (1) instead of at the top of the visit
            const f32x4* tM = reinterpret_cast<const f32x4*>(mV) + tb;
            const f32x4* tVv = reinterpret_cast<const f32x4*>(vV) + tb;
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                mq[q] = __builtin_nontemporal_load(tM + q * 64);
                vq[q] = __builtin_nontemporal_load(tVv + q * 64);
            }
        }
        // (2) sigmoid / BCE-from-logits / dlogits ; reg r <-> sample b = ROW(r,h), lane <-> SNP l (every wave of the block
        //     computes the same values)
        float dbias = 0.f;
        pv_sigmoid_bce(bce_probs, (blk * 32 + 32 <= L) && nrows >= 32, w == 0, a, bias, lo, hi, h, 0, nrows, lvalid, inv_b, lossacc,
                       dbias);
        if (PV_DEC2_VAR == 2) {  // experiment: the next visit's V requested behind the sigmoid phase
            const f32x4* tVn2 = reinterpret_cast<const f32x4*>(V) + tile(nb);
#pragma unroll
            for (int q = 0; q < 4; ++q) vn[q] = __builtin_nontemporal_load(tVn2 + q * 64);
        }
        // bias: sum over the 32 samples = 16 regs + the other half-wave; block column 0 updates it
        dbias += __shfl_xor(dbias, 32);
        if (PART != 1 && w == 0 && h == 0 && lvalid) {
            float w_ = bias;
            pv_adam(w_, cm, cv, dbias, alpha, adam);
            cb[l] = w_;
            mC[l] = cm;
            vC[l] = cv;
        }
        // (3) dg[b][unit] += sum_l da[b][l] * V[unit][l] : M = b, N = unit, K = l.  A needs da^T, B needs V^T (unit on the
        //     lane): both staged through this wave's LDS tiles.
        if (PART != 2) {
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            dt[PV_ROW(r, h) * 33 + c] = a[r];                 // [b][l_loc]
            vt[PV_ROW(r, h) * 33 + c] = vv[r >> 2][r & 3];    // [unit_loc][l_loc]
        }
        __builtin_amdgcn_s_waitcnt(0xc07f);
        __builtin_amdgcn_wave_barrier();
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const float av = dt[c * 33 + PV_ROW(r, h)];  // A[i=b][k=l]
            const float bv = vt[c * 33 + PV_ROW(r, h)];  // B[k=l][j=unit_loc]
            dg = __builtin_amdgcn_mfma_f32_32x32x2f32(av, bv, dg, 0, 0, 0);
        }
        __builtin_amdgcn_s_waitcnt(0xc07f);
        __builtin_amdgcn_wave_barrier();
        }
        // (4) dV[unit][l] = sum_b g[b][unit] * da[b][l] : M = unit, N = l, K = b ; then Adam on the block
        if (PART != 1) {
        f32x16 dv;
#pragma unroll
        for (int r = 0; r < 16; ++r) dv[r] = 0.f;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            dv = __builtin_amdgcn_mfma_f32_32x32x2f32(g4[r], a[r], dv, 0, 0, 0);
        }
        f32x4* oV = reinterpret_cast<f32x4*>(V2) + tb;
        f32x4* oM = reinterpret_cast<f32x4*>(mV2) + tb;
        f32x4* oVv = reinterpret_cast<f32x4*>(vV2) + tb;
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            f32x4 wq = vv[q], m4 = mq[q], v4 = vq[q];
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                float w_ = wq[e], m_ = m4[e], v_ = v4[e];
                pv_adam(w_, m_, v_, dv[q * 4 + e], alpha, adam);
                wq[e] = w_;
                m4[e] = m_;
                v4[e] = v_;
            }
            __builtin_nontemporal_store(wq, oV + q * 64);
            __builtin_nontemporal_store(m4, oM + q * 64);
            __builtin_nontemporal_store(v4, oVv + q * 64);
        }
        }
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int q = 0; q < 4; ++q) vv[q] = vn[q];
        xw = xw_n;
        bias = bias_n;
        blk = block_of(it + 1);
    }
    if (PART == 2) return;  // the loss and dg came from the PART 1 launch
    // ---- per-workgroup reductions (fixed order) ----
    __syncthreads();  // everyone done with g_s / the staging tiles
    PV_WT(0, 2);
    float* red = smem;  // [NW][16][64] for dg when BPI > 1, then [NW] loss
    lossacc += __shfl_xor(lossacc, 1);
    lossacc += __shfl_xor(lossacc, 2);
    lossacc += __shfl_xor(lossacc, 4);
    lossacc += __shfl_xor(lossacc, 8);
    lossacc += __shfl_xor(lossacc, 16);
    lossacc += __shfl_xor(lossacc, 32);
    if (!sync_ok) lossacc = __builtin_nanf("");  // a sync point gave up: make the step loudly wrong
    float* lred = red + NW * 1024;
    if (lane == 0) lred[wave] = lossacc;
    if (BPI > 1) {
#pragma unroll
        for (int r = 0; r < 16; ++r) red[(wave * 16 + r) * 64 + lane] = dg[r];
    }
    __syncthreads();
    if (tid == 0) {
        float sacc = 0.f;
        for (int w2 = 0; w2 < NW; ++w2) sacc += lred[w2];
        losspart[blockIdx.x] = sacc;
    }
    float* dgp = dgpart + (int64_t)blockIdx.x * (32 * WP);
    if (BPI > 1) {
        for (int e = tid; e < NWB * 1024; e += NW * 64) {
            const int wq = e >> 10, f = e & 1023;  // block column, fragment element
            float sum = red[wq * 1024 + f];
#pragma unroll
            for (int b2 = 1; b2 < BPI; ++b2) sum += red[(b2 * NWB + wq) * 1024 + f];
            const int r = f >> 6, ln = f & 63;
            dgp[PV_ROW(r, ln >> 5) * WP + wq * 32 + (ln & 31)] = sum;
        }
    } else {
#pragma unroll
        for (int r = 0; r < 16; ++r) dgp[PV_ROW(r, h) * WP + w * 32 + c] = dg[r];
    }
    PV_WT(0, 3);
}

// ---------------------------------------------------------------------------------------------
// Wide decoder, training pass for a batch of T = 2 or 4 row tiles (33 .. 128 rows) in ONE weight pass, width 128.
// At T tiles a subtile visit is ~200 T matrix instructions and 16 T sigmoids against the same 96 KB of traffic: the kernel
// is bound by the matrix pipe and the VALU, not by HBM, and what counts is that no lane does redundant arithmetic and no
// wave waits for another more than it must.  So here the waves of a workgroup own ROW TILES, not block columns (pvk_dec2):
// wave (sub, t) holds the whole subtile `sub` of V (64 registers) and does tile t's logits (full contraction, no exchange),
// sigmoid / BCE, dg -- exactly pvk_dec's per-subtile work, once per tile and none of it twice.  What the T waves of a
// subtile share goes through LDS at two sync points per visit:
//   C: V^T, which the dg product needs as its B operand, is staged once per subtile -- wave t transposes block columns
//      t, t + T, ... and every wave reads all four (each wave staging all of V^T itself: 48 more LDS writes and 8 waits);
//   A: dV = sum_t g_t^T da_t -- every wave posts its four 32 x 32 fragments, then wave t sums block columns t, t + T, ...
//      over the tiles in tile order and does Adam on them, all waves at once (a sync per column serialised the updates:
//      each sync waited for the one wave that was updating).
// A third counter, B, is arrived at after the update and waited for only before the next visit's staging (the wave's da^T
// tile aliases its post area), a whole logits + sigmoid phase later: it does not block in practice.
// One workgroup (4 waves = 4 / T subtiles x T tiles) per CU, 512-register budget: V of the next visit is requested a
// visit ahead, the moments of the columns a wave updates at the top of the visit.
// LDS: g tiles [T][32][LDG] + per wave { post [4][16][64] (da^T [32][33] aliased), V^T columns [4/T][32][33] } + bias-gradient
// slots (two sets, by visit parity).
// ---------------------------------------------------------------------------------------------
__device__ __forceinline__ void pv_block_arrive(int* cnt, int lane) {
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");  // this wave's LDS accesses are done before it is counted
    __builtin_amdgcn_wave_barrier();
    if (lane == 0) __hip_atomic_fetch_add(cnt, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
}
__device__ __forceinline__ bool pv_block_wait(int* cnt, int target) {
    bool ok = false;
    for (int spin = 0; spin < (1 << 22); ++spin) {
        if (__hip_atomic_load(cnt, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP) >= target) { ok = true; break; }
        __builtin_amdgcn_s_sleep(1);
    }
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
    return ok;
}

template <int T>
struct PvDec3 {
    static constexpr int NW = 4, SPW = NW / T, CPW = 4 / T;   // waves, subtiles per workgroup iteration, block columns a wave updates
    static constexpr int WP = 128, LDG = WP + 1;
    static constexpr int WAVE_F = 4 * 1024 + CPW * 32 * 33;  // floats of LDS per wave
    static constexpr size_t lds() {
        const size_t work = (size_t)T * 32 * LDG + (size_t)NW * WAVE_F + 2 * NW * 32;
        const size_t red = (size_t)NW * 4 * 1024 + NW;  // epilogue: dg fragments of all waves + loss
        return (work > red ? work : red) * 4;
    }
};
template <int T, int MODE = PV_MODE_TRAIN>
__global__ __launch_bounds__(256, 1) void pvk_dec3(const float* __restrict__ V, const float* __restrict__ mV,
                                                   const float* __restrict__ vV, float* __restrict__ V2, float* __restrict__ mV2,
                                                   float* __restrict__ vV2, float* __restrict__ cb, float* __restrict__ mC,
                                                   float* __restrict__ vC, const float* __restrict__ g,
                                                   const uint64_t* __restrict__ Xt, int nblk, int L, const PvStep* st, PvAdam adam,
                                                   float* __restrict__ dgpart, float* __restrict__ losspart, int ts,
                                                   const float* __restrict__ gV, const float* __restrict__ gC) {
    // MODE (chunks of a batch above 32 T rows, same conventions as pvk_dec): PV_MODE_GRAD dumps dV / dc into mV2 / mC (plain
    // tiles; PvStep.acc: store or add) and leaves V, c alone; PV_MODE_TRAIN_ACC adds the dump gV / gC to this pass's gradient
    // before Adam.  The dump traffic is requested where it is used (these passes are bound by the matrix pipe).
    using G = PvDec3<T>;
    constexpr int NW = G::NW, SPW = G::SPW, WP = G::WP, LDG = G::LDG, CPW = G::CPW;
    extern __shared__ __attribute__((aligned(16))) float smem[];
    float* g_s = smem;  // [T][32][LDG]
    const int tid = threadIdx.x, wave = __builtin_amdgcn_readfirstlane(tid >> 6), lane = tid & 63, c = lane & 31, h = lane >> 5;
    float* waves_s = smem + T * 32 * LDG;
    float* post = waves_s + wave * G::WAVE_F;  // [4][16][64] this wave's dV fragments of the four block columns
    float* dt = post;                          // [32][33]    da^T (dead before the first fragment is posted)
    float* vtc = post + 4 * 1024;              // [CPW][32][33] V^T of block columns t, t + T, ...
    float* dbias_s = waves_s + NW * G::WAVE_F;  // [2][NW][32]
    const int sub = wave / T, t = wave - sub * T;
    const float* sub_s = waves_s + (sub * T) * G::WAVE_F;  // region of tile 0's wave of this subtile
    const float* gt = g_s + t * 32 * LDG;
    const int nrows = st->nrows;
    const float inv_b = st->inv_b, alpha = st->alpha;
    const int bce_probs = PV_BCE_MODE(st);
    const bool acc_grads = MODE == PV_MODE_GRAD && st->acc != 0;
    __shared__ int sync_a[SPW], sync_b[SPW], sync_c[SPW];  // arrivals of the T waves of a subtile (only ever grow)
    if (tid < SPW) sync_a[tid] = sync_b[tid] = sync_c[tid] = 0;
    bool sync_ok = true;

    auto subtile_of = [&](int it) { return (it * (int)gridDim.x + (int)blockIdx.x) * SPW + sub; };
    f32x4 vv[4][4], vn[4][4];
    uint64_t xw = 0, xw_n = 0;
    float bias = 0.f, bias_n = 0.f;
    int blk = subtile_of(0);
    if (blk < nblk) {
        const f32x4* tV = reinterpret_cast<const f32x4*>(V) + (int64_t)blk * ts + lane;
#pragma unroll
        for (int w = 0; w < 4; ++w)
#pragma unroll
            for (int q = 0; q < 4; ++q) vv[w][q] = __builtin_nontemporal_load(tV + (w * 4 + q) * 64);
        xw = Xt[((int64_t)t * nblk + blk) * 32 + c];
        bias = cb[blk * 32 + c];
    }
    for (int i = tid; i < T * 32 * WP; i += NW * 64) g_s[(i / WP) * LDG + (i % WP)] = g[i];
    __syncthreads();

    f32x16 dg[4];
#pragma unroll
    for (int w = 0; w < 4; ++w)
#pragma unroll
        for (int r = 0; r < 16; ++r) dg[w][r] = 0.f;
    float lossacc = 0.f;
#ifdef PV_DEC3_PROF
    unsigned long long d3acc[10] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0}, d3last = __builtin_amdgcn_s_memtime();
    const unsigned long long d3rt0 = __builtin_amdgcn_s_memrealtime();  // 100 MHz: in-kernel clock = cycles / this
    int d3visits = 0;
#endif
    // (the wave's g operands are the same for every visit, but 128 more registers of them cost more in AGPR copies than
    // the LDS reads they save: 832 vs 736 us at T = 2, 1M SNPs)

    for (int it = 0; blk < nblk; ++it) {
        const int nb = subtile_of(it + 1) < nblk ? subtile_of(it + 1) : blk;  // past the end: a harmless re-read
        const int l = blk * 32 + c;
        const bool lvalid = l < L;
        const int64_t tb = (int64_t)blk * ts + lane;
        // requests of this visit (the bias word first: it has landed before the next visit's sync points, behind which
        // tile 0's wave rewrites the bias in place -- loads return in order)
        bias_n = cb[nb * 32 + c];
        xw_n = Xt[((int64_t)t * nblk + nb) * 32 + c];
        // (every lane loads the bias moments, clamped: a branch here would end the scheduling region the requests are
        // spread over.)  The 1 KB requests themselves are spread over the matrix phases: issued back to back by four waves
        // in lock step they queued at the CU's address unit for ~4000 cycles per visit, every wave stalled at issue.
        const int lc = l < L ? l : L - 1;
        float cm = 0.f, cv = 0.f;
        if (MODE != PV_MODE_GRAD) {
            cm = mC[lc];
            cv = vC[lc];
        }
        // what the update of this wave's columns reads besides V: the moments (Adam), the dump (accumulate variants)
        constexpr int PER = MODE == PV_MODE_GRAD ? 1 : MODE == PV_MODE_TRAIN_ACC ? 3 : 2;
        constexpr int NREQ = PER * 4 * CPW, RSTEP = 64 / NREQ;  // one request every RSTEP matrix instructions of the dg product
        f32x4 mq[CPW][4], vq[CPW][4], gq[CPW][4];
        const f32x4* tVn = reinterpret_cast<const f32x4*>(V) + (int64_t)nb * ts + lane;
        PV_D3(0);
        const uint64_t xs = xw >> (8 * h);
        const uint32_t lo = (uint32_t)xs, hi = (uint32_t)(xs >> 32);
        // (1) logits of tile t: M = b, N = l, K = unit (all 128)
        f32x16 a;
#pragma unroll
        for (int r = 0; r < 16; ++r) a[r] = 0.f;
#pragma unroll
        for (int w = 0; w < 4; ++w)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                if ((r & 3) == 0) vn[w][r >> 2] = __builtin_nontemporal_load(tVn + (w * 4 + (r >> 2)) * 64);
                a = __builtin_amdgcn_mfma_f32_32x32x2f32(gt[c * LDG + w * 32 + PV_ROW(r, h)], vv[w][r >> 2][r & 3], a, 0, 0, 0);
            }
#pragma unroll
        for (int i = 0; i < 16; ++i) {  // 4 operand reads, one request, 4 matrix instructions
            __builtin_amdgcn_sched_group_barrier(0x100, 4, 0);
            __builtin_amdgcn_sched_group_barrier(0x020, 1, 0);
            __builtin_amdgcn_sched_group_barrier(0x008, 4, 0);
        }
        PV_D3(1);
        // (2) sigmoid / BCE / dlogits of tile t
        float dbias = 0.f;
        // full: every row of the tile and every SNP of the block exists (wave-uniform)
        pv_sigmoid_bce(bce_probs, (blk * 32 + 32 <= L) && (32 * t + 32 <= nrows), true, a, bias, lo, hi, h, 32 * t, nrows, lvalid,
                       inv_b, lossacc, dbias);
        dbias += __shfl_xor(dbias, 32);
        float* dbias_v = dbias_s + (it & 1) * NW * 32;  // read by tile 0's wave behind sync A; rewritten two visits on
        if (h == 0) dbias_v[wave * 32 + c] = dbias;
        PV_D3(2);
        // (3) dg[b][unit] += sum_l da[b][l] * V[unit][l]: da^T and this wave's share of V^T go to LDS, then sync C
        sync_ok &= pv_block_wait(&sync_b[sub], T * it);  // the other tiles' waves have read last visit's fragments
#pragma unroll
        for (int r = 0; r < 16; ++r) dt[PV_ROW(r, h) * 33 + c] = a[r];
#pragma unroll
        for (int w = 0; w < 4; ++w) {
            if (w % T != t) continue;  // wave-uniform
#pragma unroll
            for (int r = 0; r < 16; ++r) vtc[(w / T) * (32 * 33) + PV_ROW(r, h) * 33 + c] = vv[w][r >> 2][r & 3];
        }
        PV_D3(3);
        pv_block_arrive(&sync_c[sub], lane);
        sync_ok &= pv_block_wait(&sync_c[sub], T * (it + 1));
        PV_D3(4);
#pragma unroll
        for (int w = 0; w < 4; ++w) {
            const float* vt = sub_s + (w % T) * G::WAVE_F + 4 * 1024 + (w / T) * (32 * 33);
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                if ((w * 16 + r) % RSTEP == 0 && (w * 16 + r) / RSTEP < NREQ) {  // columns t, t + T, ... of this wave
                    const int j = (w * 16 + r) / RSTEP, kq = j / PER, what = j % PER, k = kq / 4, q = kq & 3;
                    const int64_t o = tb + ((t + k * T) * 4 + q) * 64;
                    const int64_t go = (int64_t)blk * (4 * 256) + ((t + k * T) * 4 + q) * 64 + lane;  // dumps are plain tiles
                    if (MODE == PV_MODE_GRAD) gq[k][q] = __builtin_nontemporal_load(reinterpret_cast<const f32x4*>(mV2) + go);
                    else if (what == 0) mq[k][q] = __builtin_nontemporal_load(reinterpret_cast<const f32x4*>(mV) + o);
                    else if (what == 1) vq[k][q] = __builtin_nontemporal_load(reinterpret_cast<const f32x4*>(vV) + o);
                    else gq[k][q] = __builtin_nontemporal_load(reinterpret_cast<const f32x4*>(gV) + go);
                }
                dg[w] = __builtin_amdgcn_mfma_f32_32x32x2f32(dt[c * 33 + PV_ROW(r, h)], vt[c * 33 + PV_ROW(r, h)], dg[w], 0, 0, 0);
            }
        }
#pragma unroll
        for (int i = 0; i < 16; ++i) {
            int nreq = 0;  // requests that belong to matrix instructions 4 i .. 4 i + 3
#pragma unroll
            for (int x = 4 * i; x < 4 * i + 4; ++x) nreq += (x % RSTEP == 0 && x / RSTEP < NREQ) ? 1 : 0;
            __builtin_amdgcn_sched_group_barrier(0x100, 8, 0);
            if (nreq == 1) __builtin_amdgcn_sched_group_barrier(0x020, 1, 0);
            if (nreq == 2) __builtin_amdgcn_sched_group_barrier(0x020, 2, 0);
            __builtin_amdgcn_sched_group_barrier(0x008, 4, 0);
        }
        PV_D3(5);
        __builtin_amdgcn_s_waitcnt(0xc07f);  // the da^T reads are done before the fragments overwrite the tile
        __builtin_amdgcn_wave_barrier();
        // (4) dV[unit][l] = sum over the tiles of g_t^T da_t: every wave posts its four fragments, sync A, then wave t sums
        //     block columns t, t + T, ... in tile order and does Adam on them
        f32x16 dv[2];
#pragma unroll
        for (int w = 0; w <= 4; ++w) {
            if (w < 4) {
#pragma unroll
                for (int r = 0; r < 16; ++r) dv[w & 1][r] = 0.f;
#pragma unroll
                for (int r = 0; r < 16; ++r)
                    dv[w & 1] = __builtin_amdgcn_mfma_f32_32x32x2f32(gt[PV_ROW(r, h) * LDG + w * 32 + c], a[r], dv[w & 1], 0, 0, 0);
            }
            if (w > 0) {  // column w - 1's fragment, behind the matrix instructions of column w
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    const f32x16& pv = dv[(w - 1) & 1];
                    const f32x4 t4 = {pv[4 * q], pv[4 * q + 1], pv[4 * q + 2], pv[4 * q + 3]};
                    reinterpret_cast<f32x4*>(post + (w - 1) * 1024)[q * 64 + lane] = t4;
                }
            }
        }
        PV_D3(6);
        pv_block_arrive(&sync_a[sub], lane);
        sync_ok &= pv_block_wait(&sync_a[sub], T * (it + 1));
        PV_D3(7);
#pragma unroll
        for (int w = 0; w < 4; ++w) {
            if (w % T != t) continue;  // wave-uniform
            const int k = w / T;
            const float* p0 = sub_s + w * 1024;
            f32x4 frag[4][T];  // all reads first: one LDS round trip per column instead of one per fragment
#pragma unroll
            for (int q = 0; q < 4; ++q)
#pragma unroll
                for (int t2 = 0; t2 < T; ++t2) frag[q][t2] = reinterpret_cast<const f32x4*>(p0 + t2 * G::WAVE_F)[q * 64 + lane];
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                f32x4 gsum = frag[q][0];
#pragma unroll
                for (int t2 = 1; t2 < T; ++t2) gsum += frag[q][t2];
                const int off = (w * 4 + q) * 64;
                const int64_t goff = (int64_t)blk * (4 * 256) + off + lane;  // gradient dumps are plain tiles
                if (MODE == PV_MODE_GRAD) {  // (the old dump was requested whether it counts or not: no branch in the dg phase)
                    if (acc_grads) gsum += gq[k][q];
                    __builtin_nontemporal_store(gsum, reinterpret_cast<f32x4*>(mV2) + goff);
                    continue;
                }
                if (MODE == PV_MODE_TRAIN_ACC) gsum += gq[k][q];
                f32x4 wq = vv[w][q], m4 = mq[k][q], v4 = vq[k][q];
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    float w_ = wq[e], m_ = m4[e], v_ = v4[e];
                    pv_adam(w_, m_, v_, gsum[e], alpha, adam);
                    wq[e] = w_;
                    m4[e] = m_;
                    v4[e] = v_;
                }
                __builtin_nontemporal_store(wq, reinterpret_cast<f32x4*>(V2) + tb + off);
                __builtin_nontemporal_store(m4, reinterpret_cast<f32x4*>(mV2) + tb + off);
                __builtin_nontemporal_store(v4, reinterpret_cast<f32x4*>(vV2) + tb + off);
            }
        }
        if (t == 0 && h == 0 && lvalid) {  // the output bias: sum of the tiles' parts, in tile order
            float gb = dbias_v[(sub * T) * 32 + c];
#pragma unroll
            for (int t2 = 1; t2 < T; ++t2) gb += dbias_v[(sub * T + t2) * 32 + c];
            if (MODE == PV_MODE_GRAD) {
                mC[l] = acc_grads ? mC[l] + gb : gb;
            } else {
                if (MODE == PV_MODE_TRAIN_ACC) gb += gC[l];
                float w_ = bias;
                pv_adam(w_, cm, cv, gb, alpha, adam);
                cb[l] = w_;
                mC[l] = cm;
                vC[l] = cv;
            }
        }
        PV_D3(8);
        pv_block_arrive(&sync_b[sub], lane);
#pragma unroll
        for (int w = 0; w < 4; ++w)
#pragma unroll
            for (int q = 0; q < 4; ++q) vv[w][q] = vn[w][q];
        xw = xw_n;
        bias = bias_n;
        blk = subtile_of(it + 1);
        PV_D3(9);
#ifdef PV_DEC3_PROF
        ++d3visits;
#endif
    }
#ifdef PV_DEC3_PROF
    if (tid == 0 && blockIdx.x < 256) {
        for (int i = 0; i < 10; ++i) pv_wtrace_buf[blockIdx.x * 12 + i] = d3acc[i];
        pv_wtrace_buf[blockIdx.x * 12 + 10] = d3visits;
        pv_wtrace_buf[blockIdx.x * 12 + 11] = __builtin_amdgcn_s_memrealtime() - d3rt0;
    }
#endif
    // ---- per-workgroup reductions (fixed order) ----
    __syncthreads();
    float* red = smem;  // [NW][4][16][64], then [NW] loss
    lossacc += __shfl_xor(lossacc, 1);
    lossacc += __shfl_xor(lossacc, 2);
    lossacc += __shfl_xor(lossacc, 4);
    lossacc += __shfl_xor(lossacc, 8);
    lossacc += __shfl_xor(lossacc, 16);
    lossacc += __shfl_xor(lossacc, 32);
    if (!sync_ok) lossacc = __builtin_nanf("");  // a sync point gave up: make the step loudly wrong
    float* lred = red + NW * 4 * 1024;
    if (lane == 0) lred[wave] = lossacc;
#pragma unroll
    for (int w = 0; w < 4; ++w)
#pragma unroll
        for (int r = 0; r < 16; ++r) red[((wave * 4 + w) * 16 + r) * 64 + lane] = dg[w][r];
    __syncthreads();
    if (tid == 0) {
        float sacc = 0.f;
        for (int w2 = 0; w2 < NW; ++w2) sacc += lred[w2];
        losspart[blockIdx.x] = sacc;
    }
    // dgpart[workgroup][32 T][WP]: tile t2's rows = sum over the workgroup's subtiles of wave (sub2, t2)'s fragments
    float* dgp = dgpart + (int64_t)blockIdx.x * (32 * T * WP);
    for (int e = tid; e < T * 4 * 1024; e += NW * 64) {
        const int t2 = e / 4096, f = e % 4096;  // tile, (block column, fragment element)
        float sum = red[t2 * 4096 + f];
#pragma unroll
        for (int s2 = 1; s2 < SPW; ++s2) sum += red[(s2 * T + t2) * 4096 + f];
        const int w = f >> 10, r = (f >> 6) & 15, ln = f & 63;
        dgp[(32 * t2 + PV_ROW(r, ln >> 5)) * WP + w * 32 + (ln & 31)] = sum;
    }
}

// ---------------------------------------------------------------------------------------------
// pvk_dec3 at four row tiles with its three products on the bf16 matrix pipe at f32 accuracy (round 3).  At T = 4
// pvk_dec3<4> is bound by the fp32 matrix pipe (192 instructions of ~74 cycles per visit, 14 300 of 20 090 cycles).  Here:
//   (1) logits of tile t = g_t . V: six bf16 term products (pv_split3, as pvk_dec_eval_bf): V is cut in registers, the terms of
//       g come from the operand-order table in LDS ([tile][block column][instruction][term][lane] x 16 B, 24 KB per tile);
//   (3) dg += da . V^T: both operands have the SNP on the lane where the contraction wants it in registers.  Each goes through a
//       private f32 scratch in LDS once (written by rows of registers, read back eight SNPs a lane) and is cut into terms: da^T of
//       tile t stays in wave t's registers (the A operand), V^T of block column t is posted in operand order for all four waves;
//   (4) dV = sum over ALL tiles of g_t^T . da_t as ONE contraction over the 128 rows, by the wave that updates the block column:
//       wave t owns block column t of the subtile (CPW = 1 at T = 4).  Its A operands -- the three terms of g^T for its 32 units
//       and all 128 rows -- never change during the launch and live in 96 registers; the B operands are the terms of every
//       tile's da, which that tile's wave posts in operand order (the accumulator layout of product (1) IS the B layout of this
//       contraction: no transposition), 6 KB per tile.  The sum over the tiles happens inside the accumulator: no dV fragments
//       through LDS (64 KB per visit in pvk_dec3), no second sync point.
// One sync point per visit (C: da terms, V^T terms and bias parts posted), one counter that is only waited for a visit later (B).
// Per visit (tools/trace_dec3.py, 1M SNPs): 13 600 cycles at ~2.0-2.3 GHz against 20 090 for pvk_dec3<4> -- 4 600 of matrix pipe,
// the rest VALU (cutting operands ~700 instructions, sigmoid / BCE 1 600 cycles, Adam), LDS round trips and the sync point, none of
// it overlapped with the matrix instructions: one wave per SIMD on 410 registers.  Two waves per SIMD (a matrix wave and a helper
// wave per tile) would overlap them; the g table's 96 KB of LDS and the 256-register budget are what stand in the way.
// LDS: g table 96 KB + V^T terms 24 KB (each wave's 6 KB double as its transposition scratch) + da terms 24 KB + bias parts = 145 KB.
// ---------------------------------------------------------------------------------------------
struct PvDec3b {
    static constexpr int T = 4, NW = 4, WP = 128;
    static constexpr int GTAB_F = T * 4 * 2 * 3 * 64 * 4;  // floats of the g operand table
    static constexpr int RS = 36;                          // row stride of the f32 transposition scratch (16-byte rows)
    static constexpr int WAVE_F = 2 * 3 * 64 * 4;          // V^T terms of one block column; doubles as the transposition scratch
    static constexpr int POST_F = T * 2 * 3 * 64 * 4;      // da terms of the four tiles
    static constexpr size_t lds() {
        const size_t work = (size_t)GTAB_F + NW * WAVE_F + POST_F + 2 * NW * 32;
        const size_t red = (size_t)NW * 4 * 1024 + NW;
        return (work > red ? work : red) * 4;
    }
};
template <int MODE = PV_MODE_TRAIN>
__global__ __launch_bounds__(256, 1) void pvk_dec3b(const float* __restrict__ V, const float* __restrict__ mV,
                                                    const float* __restrict__ vV, float* __restrict__ V2, float* __restrict__ mV2,
                                                    float* __restrict__ vV2, float* __restrict__ cb, float* __restrict__ mC,
                                                    float* __restrict__ vC, const float* __restrict__ g,
                                                    const uint64_t* __restrict__ Xt, int nblk, int L, const PvStep* st, PvAdam adam,
                                                    float* __restrict__ dgpart, float* __restrict__ losspart, int ts,
                                                    const float* __restrict__ gV, const float* __restrict__ gC) {
    using G = PvDec3b;
    constexpr int T = G::T, NW = G::NW, WP = G::WP;
    extern __shared__ __attribute__((aligned(16))) float smem[];
    pv_u32x4* gop = reinterpret_cast<pv_u32x4*>(smem);  // [T][4 w][2 i][3 term][64 lanes]
    const int tid = threadIdx.x, wave = __builtin_amdgcn_readfirstlane(tid >> 6), lane = tid & 63, c = lane & 31, h = lane >> 5;
    const int t = wave;  // this wave's row tile, and the block column it updates
    float* waves_s = smem + G::GTAB_F;
    // this wave's 6 KB: the three terms of V^T of block column t in the B-operand order of product (3), [2 i][3 term][64 lanes]
    // x 16 B, read by all four waves; before they are written the same bytes serve as the wave's private f32 scratch
    // [32][RS] through which first da of this tile, then the V column are transposed
    float* scr = waves_s + wave * G::WAVE_F;
    pv_u32x4* vpost = reinterpret_cast<pv_u32x4*>(waves_s);  // [4 w][2 i][3 term][64 lanes]
    pv_u32x4* dpost = reinterpret_cast<pv_u32x4*>(waves_s + NW * G::WAVE_F);  // [T][2 q][3 term][64 lanes]
    float* dbias_s = waves_s + NW * G::WAVE_F + G::POST_F;                    // [2][NW][32]
    const int nrows = st->nrows;
    const float inv_b = st->inv_b, alpha = st->alpha;
    const int bce_probs = PV_BCE_MODE(st);
    const bool acc_grads = MODE == PV_MODE_GRAD && st->acc != 0;
    __shared__ int sync_b, sync_c;  // arrivals of the four waves (only ever grow)
    if (tid == 0) sync_b = sync_c = 0;
    bool sync_ok = true;

    auto subtile_of = [&](int it) { return it * (int)gridDim.x + (int)blockIdx.x; };
    f32x4 vv[4][4], vn[4][4];
    uint64_t xw = 0, xw_n = 0;
    float bias = 0.f, bias_n = 0.f;
    int blk = subtile_of(0);
    if (blk < nblk) {
        const f32x4* tV = reinterpret_cast<const f32x4*>(V) + (int64_t)blk * ts + lane;
#pragma unroll
        for (int w = 0; w < 4; ++w)
#pragma unroll
            for (int q = 0; q < 4; ++q) vv[w][q] = __builtin_nontemporal_load(tV + (w * 4 + q) * 64);
        xw = Xt[((int64_t)t * nblk + blk) * 32 + c];
        bias = cb[blk * 32 + c];
    }
    // g operand table of product (1): lane (b, kg) of instruction (w, i) holds units 32 w + ROW(8 i + j, kg), j = 0 .. 7
    for (int e = tid; e < T * 4 * 2 * 64; e += NW * 64) {
        const int ln = e & 63, i = (e >> 6) & 1, w = (e >> 7) & 3, t2 = e >> 9, cc = ln & 31, hh = ln >> 5;
        float x[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) x[j] = g[(32 * t2 + cc) * WP + w * 32 + PV_ROW(8 * i + j, hh)];
        pv_u32x4 th, tm, tl;
        pv_split3(x, th, tm, tl);
        pv_u32x4* dst = gop + (((t2 * 4 + w) * 2 + i) * 3) * 64 + ln;
        dst[0] = th;
        dst[64] = tm;
        dst[128] = tl;
    }
    // A operands of product (4), resident: lane (unit 32 t + c, kg = h) of instruction (t2, q) holds rows 32 t2 + ROW(8 q + j, h)
    pv_u32x4 gth[T][2], gtm[T][2], gtl[T][2];
#pragma unroll
    for (int t2 = 0; t2 < T; ++t2)
#pragma unroll
        for (int q = 0; q < 2; ++q) {
            float x[8];
#pragma unroll
            for (int j = 0; j < 8; ++j) x[j] = g[(32 * t2 + PV_ROW(8 * q + j, h)) * WP + 32 * t + c];
            pv_split3(x, gth[t2][q], gtm[t2][q], gtl[t2][q]);
        }
    __syncthreads();

    f32x16 dg[4];
#pragma unroll
    for (int w = 0; w < 4; ++w)
#pragma unroll
        for (int r = 0; r < 16; ++r) dg[w][r] = 0.f;
    float lossacc = 0.f;

#ifdef PV_DEC3_PROF
    unsigned long long d3acc[10] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0}, d3last = __builtin_amdgcn_s_memtime();
    const unsigned long long d3rt0 = __builtin_amdgcn_s_memrealtime();
    int d3visits = 0;
#endif
    for (int it = 0; blk < nblk; ++it) {
        const int nb = subtile_of(it + 1) < nblk ? subtile_of(it + 1) : blk;  // past the end: a harmless re-read
        const int l = blk * 32 + c;
        const bool lvalid = l < L;
        const int64_t tb = (int64_t)blk * ts + lane;
        // requests of this visit (the bias word first: loads return in order, so it has landed before the next visit's sync point,
        // behind which wave 0 rewrites the bias in place)
        bias_n = cb[nb * 32 + c];
        xw_n = Xt[((int64_t)t * nblk + nb) * 32 + c];
        const int lc = l < L ? l : L - 1;
        float cm = 0.f, cv = 0.f;
        if (MODE != PV_MODE_GRAD) {
            cm = mC[lc];
            cv = vC[lc];
        }
        f32x4 mq[4], vq[4], gq[4];
        const f32x4* tVn = reinterpret_cast<const f32x4*>(V) + (int64_t)nb * ts + lane;
        const int64_t go = (int64_t)blk * (4 * 256) + (t * 4) * 64 + lane;  // gradient dumps are plain tiles
        const uint64_t xs = xw >> (8 * h);
        const uint32_t lo = (uint32_t)xs, hi = (uint32_t)(xs >> 32);
        PV_D3(0);
        // (1) logits of tile t on the bf16 pipe; the next visit's V is requested between the instructions
        f32x16 a;
#pragma unroll
        for (int r = 0; r < 16; ++r) a[r] = 0.f;
        // The eight V values of group s + 1 are cut into their terms between the six matrix instructions of group s = (w, i), one
        // stage of pv_cut_stage behind each, held in place by scheduling barriers.  This does NOT overlap VALU and matrix work:
        // a wave's own VALU instructions do not issue in the shadow of its own matrix instruction (tools/mfma_valu_probe.hip: 33.7
        // cycles per v_mfma_f32_32x32x16_bf16 alone, + 2.2 per independent VALU instruction behind it, with one or two accumulator
        // chains), and this kernel runs one wave per SIMD.  What the staging buys is that operand reads and requests are never
        // waited for: 3900 -> 2850 cycles per visit for the phase (1536 of matrix pipe + ~350 VALU instructions).
        pv_u32x4 bh[2], bm[2], bl[2];
        PvCut3 ct;
        auto cut_load = [&](int s_) {
            const int w = s_ >> 1, i = s_ & 1;
#pragma unroll
            for (int j = 0; j < 8; ++j) ct.x[j] = vv[w][2 * i + (j >> 2)][j & 3];
        };
        cut_load(0);
        pv_cut_stage<0>(ct, bh[0], bm[0], bl[0]);
        pv_cut_stage<1>(ct, bh[0], bm[0], bl[0]);
        pv_cut_stage<2>(ct, bh[0], bm[0], bl[0]);
        pv_cut_stage<3>(ct, bh[0], bm[0], bl[0]);
        pv_cut_stage<4>(ct, bh[0], bm[0], bl[0]);
        pv_cut_stage<5>(ct, bh[0], bm[0], bl[0]);
#pragma unroll
        for (int s_ = 0; s_ < 8; ++s_) {
            const int w = s_ >> 1, i = s_ & 1, cur = s_ & 1, nxt = cur ^ 1;
            const pv_u32x4* src = gop + (((t * 4 + w) * 2 + i) * 3) * 64 + lane;
            const pv_u32x4 ah = src[0], am = src[64], al = src[128];
            vn[w][2 * i] = __builtin_nontemporal_load(tVn + (w * 4 + 2 * i) * 64);
            vn[w][2 * i + 1] = __builtin_nontemporal_load(tVn + (w * 4 + 2 * i + 1) * 64);
            if (s_ + 1 < 8) cut_load(s_ + 1);
            __builtin_amdgcn_sched_barrier(0);
            a = pv_mfma_bf16(al, bh[cur], a);
            if (s_ + 1 < 8) pv_cut_stage<0>(ct, bh[nxt], bm[nxt], bl[nxt]);
            __builtin_amdgcn_sched_barrier(0);
            a = pv_mfma_bf16(ah, bl[cur], a);
            if (s_ + 1 < 8) pv_cut_stage<1>(ct, bh[nxt], bm[nxt], bl[nxt]);
            __builtin_amdgcn_sched_barrier(0);
            a = pv_mfma_bf16(am, bm[cur], a);
            if (s_ + 1 < 8) pv_cut_stage<2>(ct, bh[nxt], bm[nxt], bl[nxt]);
            __builtin_amdgcn_sched_barrier(0);
            a = pv_mfma_bf16(am, bh[cur], a);
            if (s_ + 1 < 8) pv_cut_stage<3>(ct, bh[nxt], bm[nxt], bl[nxt]);
            __builtin_amdgcn_sched_barrier(0);
            a = pv_mfma_bf16(ah, bm[cur], a);
            if (s_ + 1 < 8) pv_cut_stage<4>(ct, bh[nxt], bm[nxt], bl[nxt]);
            __builtin_amdgcn_sched_barrier(0);
            a = pv_mfma_bf16(ah, bh[cur], a);
            if (s_ + 1 < 8) pv_cut_stage<5>(ct, bh[nxt], bm[nxt], bl[nxt]);
            __builtin_amdgcn_sched_barrier(0);
        }
        PV_D3(1);
        // (2) sigmoid / BCE / dlogits of tile t
        float dbias = 0.f;
        pv_sigmoid_bce(bce_probs, (blk * 32 + 32 <= L) && (32 * t + 32 <= nrows), true, a, bias, lo, hi, h, 32 * t, nrows, lvalid,
                       inv_b, lossacc, dbias);
        dbias += __shfl_xor(dbias, 32);
        PV_D3(2);
        // everything the other waves read of last visit's posts has been read (their arrival at B) before it is overwritten
        sync_ok &= pv_block_wait(&sync_b, T * it);
        float* dbias_v = dbias_s + (it & 1) * NW * 32;
        if (h == 0) dbias_v[wave * 32 + c] = dbias;
        // da of this tile: its three bf16 terms in the B-operand order of product (4) (lane (l, kg = h): rows ROW(8 q + j, h)),
        // da^T in f32 for product (3), and this wave's block column of V^T
#pragma unroll
        for (int q = 0; q < 2; ++q) {
            const float x[8] = {a[8 * q], a[8 * q + 1], a[8 * q + 2], a[8 * q + 3], a[8 * q + 4], a[8 * q + 5], a[8 * q + 6], a[8 * q + 7]};
            pv_u32x4 dh, dm, dl;
            pv_split3(x, dh, dm, dl);
            pv_u32x4* dst = dpost + ((t * 2 + q) * 3) * 64 + lane;
            dst[0] = dh;
            dst[64] = dm;
            dst[128] = dl;
        }
        // da^T of this tile as the A operand of product (3): lane (b, kg) of instruction i holds SNPs 16 i + 8 kg + j.  The
        // accumulator has the SNP on the lane: through the scratch, [b][l] written by rows of registers, read back 8 SNPs a lane
        pv_u32x4 ath[2], atm[2], atl[2];
        constexpr int RS = G::RS;
#pragma unroll
        for (int r = 0; r < 16; ++r) scr[PV_ROW(r, h) * RS + c] = a[r];
        __builtin_amdgcn_s_waitcnt(0xc07f);
        __builtin_amdgcn_wave_barrier();
        {
            f32x4 xa[2][2];
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
                for (int p = 0; p < 2; ++p) xa[i][p] = *reinterpret_cast<const f32x4*>(scr + c * RS + 16 * i + 8 * h + 4 * p);
            __builtin_amdgcn_s_waitcnt(0xc07f);
            __builtin_amdgcn_wave_barrier();
            // ... then V^T of block column t the same way (same scratch), cut into its terms
#pragma unroll
            for (int w = 0; w < 4; ++w) {
                if (w != t) continue;  // wave-uniform
#pragma unroll
                for (int r = 0; r < 16; ++r) scr[PV_ROW(r, h) * RS + c] = vv[w][r >> 2][r & 3];
            }
#pragma unroll
            for (int i = 0; i < 2; ++i) {
                const float x[8] = {xa[i][0][0], xa[i][0][1], xa[i][0][2], xa[i][0][3], xa[i][1][0], xa[i][1][1], xa[i][1][2], xa[i][1][3]};
                pv_split3(x, ath[i], atm[i], atl[i]);
            }
            __builtin_amdgcn_s_waitcnt(0xc07f);
            __builtin_amdgcn_wave_barrier();
            f32x4 xv[2][2];
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
                for (int p = 0; p < 2; ++p) xv[i][p] = *reinterpret_cast<const f32x4*>(scr + c * RS + 16 * i + 8 * h + 4 * p);
            __builtin_amdgcn_s_waitcnt(0xc07f);  // every read of the scratch is back before the terms overwrite it
            __builtin_amdgcn_wave_barrier();
#pragma unroll
            for (int i = 0; i < 2; ++i) {
                const float x[8] = {xv[i][0][0], xv[i][0][1], xv[i][0][2], xv[i][0][3], xv[i][1][0], xv[i][1][1], xv[i][1][2], xv[i][1][3]};
                pv_u32x4 vh, vm, vl;
                pv_split3(x, vh, vm, vl);
                pv_u32x4* dst = vpost + ((t * 2 + i) * 3) * 64 + lane;
                dst[0] = vh;
                dst[64] = vm;
                dst[128] = vl;
            }
        }
        PV_D3(3);
        pv_block_arrive(&sync_c, lane);
        // what the update of block column t reads besides V: the moments (Adam), the dump (accumulate variants)
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            if (MODE == PV_MODE_GRAD) {
                gq[q] = __builtin_nontemporal_load(reinterpret_cast<const f32x4*>(mV2) + go + q * 64);
            } else {
                mq[q] = __builtin_nontemporal_load(reinterpret_cast<const f32x4*>(mV) + tb + (t * 4 + q) * 64);
                vq[q] = __builtin_nontemporal_load(reinterpret_cast<const f32x4*>(vV) + tb + (t * 4 + q) * 64);
                if (MODE == PV_MODE_TRAIN_ACC) gq[q] = __builtin_nontemporal_load(reinterpret_cast<const f32x4*>(gV) + go + q * 64);
            }
        }
        sync_ok &= pv_block_wait(&sync_c, T * (it + 1));
        PV_D3(4);
        // (4) dV of block column t over all 128 rows on the bf16 pipe: tiles in order, one accumulator
        f32x16 dv;
#pragma unroll
        for (int r = 0; r < 16; ++r) dv[r] = 0.f;
#pragma unroll
        for (int t2 = 0; t2 < T; ++t2)
#pragma unroll
            for (int q = 0; q < 2; ++q) {
                const pv_u32x4* src = dpost + ((t2 * 2 + q) * 3) * 64 + lane;
                const pv_u32x4 dh = src[0], dm = src[64], dl = src[128];
                dv = pv_mfma_bf16(gtl[t2][q], dh, dv);
                dv = pv_mfma_bf16(gth[t2][q], dl, dv);
                dv = pv_mfma_bf16(gtm[t2][q], dm, dv);
                dv = pv_mfma_bf16(gtm[t2][q], dh, dv);
                dv = pv_mfma_bf16(gth[t2][q], dm, dv);
                dv = pv_mfma_bf16(gth[t2][q], dh, dv);
            }
        PV_D3(5);
        // (3) dg[b][unit] += sum_l da[b][l] V[unit][l] on the bf16 pipe: A = the terms of da^T (registers), B = the terms of V^T of
        //     block column w (wave w's post).  The update of block column t sits in the same basic block (the column is picked
        //     with selects, not with a branch on t), so that its stores and the waits for its moments are scheduled among the
        //     matrix instructions; the arithmetic itself does not hide behind them (see product (1)).
        f32x4 wsel[4];
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            wsel[q] = vv[0][q];
#pragma unroll
            for (int w = 1; w < 4; ++w) wsel[q] = (t == w) ? vv[w][q] : wsel[q];
        }
        f32x4 oW[4], oM[4], oV[4];
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            f32x4 gsum = {dv[4 * q], dv[4 * q + 1], dv[4 * q + 2], dv[4 * q + 3]};
            if (MODE == PV_MODE_GRAD) {
                if (acc_grads) gsum += gq[q];
                oM[q] = gsum;
                continue;
            }
            if (MODE == PV_MODE_TRAIN_ACC) gsum += gq[q];
            f32x4 wq = wsel[q], m4 = mq[q], v4 = vq[q];
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                float w_ = wq[e], m_ = m4[e], v_ = v4[e];
                pv_adam(w_, m_, v_, gsum[e], alpha, adam);
                wq[e] = w_;
                m4[e] = m_;
                v4[e] = v_;
            }
            oW[q] = wq;
            oM[q] = m4;
            oV[q] = v4;
        }
#pragma unroll
        for (int w = 0; w < 4; ++w)
#pragma unroll
            for (int i = 0; i < 2; ++i) {
                const pv_u32x4* src = vpost + ((w * 2 + i) * 3) * 64 + lane;
                const pv_u32x4 vh = src[0], vm = src[64], vl = src[128];
                dg[w] = pv_mfma_bf16(atl[i], vh, dg[w]);
                dg[w] = pv_mfma_bf16(ath[i], vl, dg[w]);
                dg[w] = pv_mfma_bf16(atm[i], vm, dg[w]);
                dg[w] = pv_mfma_bf16(atm[i], vh, dg[w]);
                dg[w] = pv_mfma_bf16(ath[i], vm, dg[w]);
                dg[w] = pv_mfma_bf16(ath[i], vh, dg[w]);
            }
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const int off = (t * 4 + q) * 64;
            if (MODE == PV_MODE_GRAD) {
                __builtin_nontemporal_store(oM[q], reinterpret_cast<f32x4*>(mV2) + go + q * 64);
            } else {
                __builtin_nontemporal_store(oW[q], reinterpret_cast<f32x4*>(V2) + tb + off);
                __builtin_nontemporal_store(oM[q], reinterpret_cast<f32x4*>(mV2) + tb + off);
                __builtin_nontemporal_store(oV[q], reinterpret_cast<f32x4*>(vV2) + tb + off);
            }
        }
#pragma unroll
        for (int s_ = 0; s_ < 8; ++s_) {
            __builtin_amdgcn_sched_group_barrier(0x100, 3, 0);
#pragma unroll
            for (int m = 0; m < 6; ++m) {
                __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
                __builtin_amdgcn_sched_group_barrier(0x002, 5, 0);
            }
        }
        PV_D3(6);
        if (t == 0 && h == 0 && lvalid) {  // the output bias: sum of the tiles' parts, in tile order
            float gb = dbias_v[c];
#pragma unroll
            for (int t2 = 1; t2 < T; ++t2) gb += dbias_v[t2 * 32 + c];
            if (MODE == PV_MODE_GRAD) {
                mC[l] = acc_grads ? mC[l] + gb : gb;
            } else {
                if (MODE == PV_MODE_TRAIN_ACC) gb += gC[l];
                float w_ = bias;
                pv_adam(w_, cm, cv, gb, alpha, adam);
                cb[l] = w_;
                mC[l] = cm;
                vC[l] = cv;
            }
        }
        PV_D3(8);
        pv_block_arrive(&sync_b, lane);
#pragma unroll
        for (int w = 0; w < 4; ++w)
#pragma unroll
            for (int q = 0; q < 4; ++q) vv[w][q] = vn[w][q];
        xw = xw_n;
        bias = bias_n;
        blk = subtile_of(it + 1);
        PV_D3(9);
#ifdef PV_DEC3_PROF
        ++d3visits;
#endif
    }
#ifdef PV_DEC3_PROF
    if (tid == 0 && blockIdx.x < 256) {
        for (int i = 0; i < 10; ++i) pv_wtrace_buf[blockIdx.x * 12 + i] = d3acc[i];
        pv_wtrace_buf[blockIdx.x * 12 + 10] = d3visits;
        pv_wtrace_buf[blockIdx.x * 12 + 11] = __builtin_amdgcn_s_memrealtime() - d3rt0;
    }
#endif
    // ---- per-workgroup reductions (fixed order) ----
    __syncthreads();
    float* red = smem;  // [NW][4][16][64], then [NW] loss
    lossacc += __shfl_xor(lossacc, 1);
    lossacc += __shfl_xor(lossacc, 2);
    lossacc += __shfl_xor(lossacc, 4);
    lossacc += __shfl_xor(lossacc, 8);
    lossacc += __shfl_xor(lossacc, 16);
    lossacc += __shfl_xor(lossacc, 32);
    if (!sync_ok) lossacc = __builtin_nanf("");  // a sync point gave up: make the step loudly wrong
    float* lred = red + NW * 4 * 1024;
    if (lane == 0) lred[wave] = lossacc;
#pragma unroll
    for (int w = 0; w < 4; ++w)
#pragma unroll
        for (int r = 0; r < 16; ++r) red[((wave * 4 + w) * 16 + r) * 64 + lane] = dg[w][r];
    __syncthreads();
    if (tid == 0) {
        float sacc = 0.f;
        for (int w2 = 0; w2 < NW; ++w2) sacc += lred[w2];
        losspart[blockIdx.x] = sacc;
    }
    // dgpart[workgroup][32 T][WP]: tile t2's rows = wave t2's fragments
    float* dgp = dgpart + (int64_t)blockIdx.x * (32 * T * WP);
    for (int e = tid; e < T * 4 * 1024; e += NW * 64) {
        const int t2 = e / 4096, f = e % 4096;  // tile, (block column, fragment element)
        const int w = f >> 10, r = (f >> 6) & 15, ln = f & 63;
        dgp[(32 * t2 + PV_ROW(r, ln >> 5)) * WP + w * 32 + (ln & 31)] = red[t2 * 4096 + f];
    }
}

// ---------------------------------------------------------------------------------------------
// Wide decoder, evaluation pass (loss only: validation batches, popvae.py:496) for T row tiles, width 128: pvk_dec<4, 0, ..>'s
// read-only sweep with the subtile held in registers while every tile's logits and BCE terms are formed -- V is read once
// per 32 T rows.  One wave per subtile, round-robin; LDS: the T g tiles.  losspart[workgroup] = summed BCE of its SNPs.
// ---------------------------------------------------------------------------------------------
template <int T, int PER_TILE = 0>
__global__ __launch_bounds__(256) void pvk_dec_eval_mt(const float* __restrict__ V, const float* __restrict__ cb,
                                                       const float* __restrict__ g, const uint64_t* __restrict__ Xt, int nblk, int L,
                                                       const PvStep* st, float* __restrict__ losspart, int ts) {
    constexpr int WP = 128, LDG = WP + 1;
    extern __shared__ __attribute__((aligned(16))) float smem[];
    float* g_s = smem;  // [32 T][LDG]
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63, c = lane & 31, h = lane >> 5;
    const int gw = blockIdx.x * 4 + wave, nwaves = gridDim.x * 4;
    const int nrows = st->nrows;
    const float inv_b = st->inv_b;
    const int bce_probs = PV_BCE_MODE(st);
    for (int i = tid; i < 32 * T * WP; i += 256) g_s[(i / WP) * LDG + (i % WP)] = g[i];
    __syncthreads();
    // PER_TILE: every row tile is a batch of its own (several validation batches of 32 rows in one pass over V): one BCE sum per
    // tile, losspart[tile][workgroup]; otherwise the tiles are one batch, losspart[workgroup]
    constexpr int NL = PER_TILE ? T : 1;
    float lossacc[NL];
#pragma unroll
    for (int t = 0; t < NL; ++t) lossacc[t] = 0.f;
    for (int blk = gw; blk < nblk; blk += nwaves) {
        asm volatile("" ::: "memory");  // keep the g operand reads inside the loop (see pvk_dec)
        const f32x4* tV = reinterpret_cast<const f32x4*>(V) + (int64_t)blk * ts + lane;
        f32x4 vv[4][4];
#pragma unroll
        for (int w = 0; w < 4; ++w)
#pragma unroll
            for (int q = 0; q < 4; ++q) vv[w][q] = __builtin_nontemporal_load(tV + (w * 4 + q) * 64);
        const int l = blk * 32 + c;
        const bool lvalid = l < L;
        const float bias = cb[lvalid ? l : L - 1];
#pragma unroll
        for (int t = 0; t < T; ++t) {
            if (32 * t >= nrows) break;  // wave-uniform: tiles beyond the batch
            const uint64_t xs = Xt[((int64_t)t * nblk + blk) * 32 + c] >> (8 * h);
            const uint32_t lo = (uint32_t)xs, hi = (uint32_t)(xs >> 32);
            const float* gt = g_s + t * 32 * LDG;
            f32x16 a;
#pragma unroll
            for (int r = 0; r < 16; ++r) a[r] = 0.f;
#pragma unroll
            for (int w = 0; w < 4; ++w)
#pragma unroll
                for (int r = 0; r < 16; ++r)
                    a = __builtin_amdgcn_mfma_f32_32x32x2f32(gt[c * LDG + w * 32 + PV_ROW(r, h)], vv[w][r >> 2][r & 3], a, 0, 0, 0);
            float dbias = 0.f;
            pv_sigmoid_bce(bce_probs, (blk * 32 + 32 <= L) && (32 * t + 32 <= nrows), true, a, bias, lo, hi, h, 32 * t, nrows, lvalid,
                           inv_b, lossacc[PER_TILE ? t : 0], dbias);
        }
    }
    __syncthreads();
#pragma unroll
    for (int t = 0; t < NL; ++t) {
        float v = lossacc[t];
        v += __shfl_xor(v, 1);
        v += __shfl_xor(v, 2);
        v += __shfl_xor(v, 4);
        v += __shfl_xor(v, 8);
        v += __shfl_xor(v, 16);
        v += __shfl_xor(v, 32);
        if (lane == 0) smem[t * 4 + wave] = v;
    }
    __syncthreads();
    if (tid < NL) losspart[tid * gridDim.x + blockIdx.x] = ((smem[tid * 4] + smem[tid * 4 + 1]) + smem[tid * 4 + 2]) + smem[tid * 4 + 3];
}

// ---------------------------------------------------------------------------------------------
// The same evaluation pass on the bf16 matrix pipe at f32 accuracy.  logits = g . V has two general operands: each is cut into
// three bf16 terms (x = hi + mid + lo exactly, pv_split3) and the six products whose weight is at least 2^-16 of the result are
// formed -- hi.hi, hi.mid, mid.hi, hi.lo, lo.hi, mid.mid; the three dropped ones are below 2^-24, the rounding of an f32 product
// -- every one exact in the f32 accumulator's input: 12 x v_mfma_f32_32x32x16_bf16 (32 cycles each) per 32 x 32 x 32 block product
// instead of 16 x v_mfma_f32_32x32x2_f32 (64 cycles each), 2.7 x less time on the matrix pipe, which bounds this pass (343 us
// at 1M SNPs and four tiles against ~110 us of HBM time).  V is split in registers once per subtile (for all tiles); g is split
// once per launch into an operand-order table in LDS (24 KB per tile: [tile][block column][instruction][term][lane] x 16 B, one
// conflict-free ds_read_b128 per operand), shared by the EIGHT waves of a workgroup (one workgroup per CU).
// ---------------------------------------------------------------------------------------------
template <int T>
struct PvDecEvalBf {
    static constexpr int NW = 8;
    static constexpr size_t lds() { return (size_t)T * 4 * 2 * 3 * 64 * 16; }
};
template <int T, int PER_TILE = 0>
__global__ __launch_bounds__(512) void pvk_dec_eval_bf(const float* __restrict__ V, const float* __restrict__ cb,
                                                       const float* __restrict__ g, const uint64_t* __restrict__ Xt, int nblk, int L,
                                                       const PvStep* st, float* __restrict__ losspart, int ts) {
    constexpr int NW = PvDecEvalBf<T>::NW;
    extern __shared__ __attribute__((aligned(16))) float smem[];
    pv_u32x4* gop = reinterpret_cast<pv_u32x4*>(smem);  // [T][4 w][2 i][3 term][64 lanes]
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63, c = lane & 31, h = lane >> 5;
    const int gw = blockIdx.x * NW + wave, nwaves = gridDim.x * NW;
    const int nrows = st->nrows;
    const float inv_b = st->inv_b;
    const int bce_probs = PV_BCE_MODE(st);
    for (int e = tid; e < T * 4 * 2 * 64; e += NW * 64) {
        const int ln = e & 63, i = (e >> 6) & 1, w = (e >> 7) & 3, t = e >> 9, cc = ln & 31, hh = ln >> 5;
        float x[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) x[j] = g[(32 * t + cc) * 128 + w * 32 + PV_ROW(8 * i + j, hh)];
        pv_u32x4 th, tm, tl;
        pv_split3(x, th, tm, tl);
        pv_u32x4* dst = gop + (((t * 4 + w) * 2 + i) * 3) * 64 + ln;
        dst[0] = th;
        dst[64] = tm;
        dst[128] = tl;
    }
    __syncthreads();
    constexpr int NL = PER_TILE ? T : 1;
    float lossacc[NL];
#pragma unroll
    for (int t = 0; t < NL; ++t) lossacc[t] = 0.f;
    for (int blk = gw; blk < nblk; blk += nwaves) {
        const f32x4* tV = reinterpret_cast<const f32x4*>(V) + (int64_t)blk * ts + lane;
        pv_u32x4 bh[4][2], bm[4][2], bl[4][2];
        {
            f32x4 vv[4][4];
#pragma unroll
            for (int w = 0; w < 4; ++w)
#pragma unroll
                for (int q = 0; q < 4; ++q) vv[w][q] = __builtin_nontemporal_load(tV + (w * 4 + q) * 64);
#pragma unroll
            for (int w = 0; w < 4; ++w)
#pragma unroll
                for (int i = 0; i < 2; ++i) {
                    const float x[8] = {vv[w][2 * i][0], vv[w][2 * i][1], vv[w][2 * i][2], vv[w][2 * i][3],
                                        vv[w][2 * i + 1][0], vv[w][2 * i + 1][1], vv[w][2 * i + 1][2], vv[w][2 * i + 1][3]};
                    pv_split3(x, bh[w][i], bm[w][i], bl[w][i]);
                }
        }
        const int l = blk * 32 + c;
        const bool lvalid = l < L;
        const float bias = cb[lvalid ? l : L - 1];
#pragma unroll
        for (int t = 0; t < T; ++t) {
            if (32 * t >= nrows) break;  // wave-uniform: tiles beyond the batch
            const uint64_t xs = Xt[((int64_t)t * nblk + blk) * 32 + c] >> (8 * h);
            const uint32_t lo = (uint32_t)xs, hi = (uint32_t)(xs >> 32);
            f32x16 a;
#pragma unroll
            for (int r = 0; r < 16; ++r) a[r] = 0.f;
#pragma unroll
            for (int w = 0; w < 4; ++w)
#pragma unroll
                for (int i = 0; i < 2; ++i) {
                    const pv_u32x4* src = gop + (((t * 4 + w) * 2 + i) * 3) * 64 + lane;
                    const pv_u32x4 ah = src[0], am = src[64], al = src[128];
                    a = pv_mfma_bf16(al, bh[w][i], a);
                    a = pv_mfma_bf16(ah, bl[w][i], a);
                    a = pv_mfma_bf16(am, bm[w][i], a);
                    a = pv_mfma_bf16(am, bh[w][i], a);
                    a = pv_mfma_bf16(ah, bm[w][i], a);
                    a = pv_mfma_bf16(ah, bh[w][i], a);
                }
            float dbias = 0.f;
            pv_sigmoid_bce(bce_probs, (blk * 32 + 32 <= L) && (32 * t + 32 <= nrows), true, a, bias, lo, hi, h, 32 * t, nrows, lvalid,
                           inv_b, lossacc[PER_TILE ? t : 0], dbias);
        }
    }
    __syncthreads();
    float* red = smem;
#pragma unroll
    for (int t = 0; t < NL; ++t) {
        float v = lossacc[t];
        v += __shfl_xor(v, 1);
        v += __shfl_xor(v, 2);
        v += __shfl_xor(v, 4);
        v += __shfl_xor(v, 8);
        v += __shfl_xor(v, 16);
        v += __shfl_xor(v, 32);
        if (lane == 0) red[t * NW + wave] = v;
    }
    __syncthreads();
    if (tid < NL) {
        float sacc = 0.f;
        for (int w2 = 0; w2 < NW; ++w2) sacc += red[tid * NW + w2];
        losspart[tid * gridDim.x + blockIdx.x] = sacc;
    }
}

// Loss scalars of up to T validation batches that went through one pass (pvk_dec_eval_mt<T, 1>): batch t = rows 32 t .. of the pass.
//   stage & 1: recon[t] = sum over the workgroups of losspart[t][.] (fixed order) -> xrecon[t]   (then the shards' all-reduce)
//   stage & 2: out3[3 t ..] = (recon + kl, recon, kl) / n_t of batch t (popvae.py:423), n_t = its rows
__global__ __launch_bounds__(256) void pvk_loss_tiles(const float* __restrict__ losspart, int nwg, int ntiles,
                                                      const float* __restrict__ kl, const PvStep* st, float* __restrict__ xrecon,
                                                      float* __restrict__ out3, int stage) {
    __shared__ float sm[256];
    const int tid = threadIdx.x, t = tid >> 6, lane = tid & 63;  // 64 threads per tile, at most 4 tiles
    if (stage & 1) {
        float a = 0.f;
        if (t < ntiles)
            for (int p = lane; p < nwg; p += 64) a += losspart[t * nwg + p];
        sm[tid] = a;
        __syncthreads();
        if (lane == 0 && t < ntiles) {
            float sum = 0.f;
            for (int i = 0; i < 64; ++i) sum += sm[t * 64 + i];
            xrecon[t] = sum;
        }
        __syncthreads();
    }
    if ((stage & 2) && lane == 0 && t < ntiles) {
        const int n = min(32, st->nrows - 32 * t);
        if (n > 0) {
            const float sr = xrecon[t];
            float sk = 0.f;
            for (int b = 0; b < n; ++b) sk += kl[32 * t + b];
            const float inv = 1.f / (float)n;
            out3[3 * t] = (sr + sk) * inv;
            out3[3 * t + 1] = sr * inv;
            out3[3 * t + 2] = sk * inv;
        }
    }
}

// ---------------------------------------------------------------------------------------------
// decoder.predict (scripts/popvae_decoder_tests_HGDP.py:156): probs[b][l] = sigmoid(g_D[b] . V[:, l] + c[l]) for
// b < nrows.  One wave per SNP block, looping over the unit groups; read-only pass over V (HBM-bound,
// 4 B per wide parameter + 4 B per output).
// ---------------------------------------------------------------------------------------------
template <int NWB>
__global__ __launch_bounds__(256) void pvk_dec_probs(const float* __restrict__ V, const float* __restrict__ cb,
                                                     const float* __restrict__ g, int nblk, int64_t L, int ngrp, int ts,
                                                     int nrows, float* __restrict__ probs) {
    const int WP = ngrp * NWB * 32, LDG = WP + 1;
    extern __shared__ __attribute__((aligned(16))) float smem[];  // g_s [32][LDG]
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63, c = lane & 31, h = lane >> 5;
    const int gw = blockIdx.x * 4 + wave, nwaves = gridDim.x * 4;
    for (int i = tid; i < 32 * WP; i += 256) smem[(i / WP) * LDG + (i % WP)] = g[i];
    __syncthreads();
    for (int blk = gw; blk < nblk; blk += nwaves) {
        f32x16 a;
#pragma unroll
        for (int r = 0; r < 16; ++r) a[r] = 0.f;
        for (int grp = 0; grp < ngrp; ++grp) {
            const f32x4* tV = reinterpret_cast<const f32x4*>(V) + ((int64_t)blk * ngrp + grp) * ts;
            f32x4 vv[NWB][4];
#pragma unroll
            for (int w = 0; w < NWB; ++w)
#pragma unroll
                for (int q = 0; q < 4; ++q) vv[w][q] = __builtin_nontemporal_load(tV + (w * 4 + q) * 64 + lane);
#pragma unroll
            for (int w = 0; w < NWB; ++w)
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const float ga = smem[c * LDG + (grp * NWB + w) * 32 + PV_ROW(r, h)];
                    a = __builtin_amdgcn_mfma_f32_32x32x2f32(ga, vv[w][r >> 2][r & 3], a, 0, 0, 0);
                }
        }
        const int64_t l = (int64_t)blk * 32 + c;
        if (l < L) {
            const float bias = cb[l];
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int b = PV_ROW(r, h);
                const float z = a[r] + bias;
                const float e = __expf(-fabsf(z));
                const float inv = __builtin_amdgcn_rcpf(1.f + e);
                if (b < nrows) probs[(int64_t)b * L + l] = z >= 0.f ? inv : e * inv;
            }
        }
    }
}

// loss = mean_b(recon + kl) over valid rows (popvae.py:423) -> out[0..2] = loss, mean recon, mean kl
__device__ __forceinline__ void pv_loss_final(const float* __restrict__ recon, const float* __restrict__ kl, const PvStep* st,
                                              float* __restrict__ out) {
    if (threadIdx.x == 0) {
        const int n = st->nrows;
        float sr = recon[0], sk = 0.f;
        for (int b = 0; b < n; ++b) sk += kl[b];
        out[0] = (sr + sk) * st->inv_b;
        out[1] = sr * st->inv_b;
        out[2] = sk * st->inv_b;
    }
}
__global__ void pvk_loss_final(const float* __restrict__ recon, const float* __restrict__ kl, const PvStep* st,
                               float* __restrict__ out) {
    pv_loss_final(recon, kl, st, out);
}

// ---------------------------------------------------------------------------------------------
// Wide encoder backward + Adam: dW1[l][u] = 0.5 * sum_b code(b,l) * dpre1[b][u]; M = l, N = u, K = b.
// One wave per subtile; dpre1 staged once per workgroup in LDS.
// ---------------------------------------------------------------------------------------------
// FUSE = 1 (plain training pass only): the freshly updated block, still in registers, is also multiplied with the NEXT
// batch's genotype codes -- the next step's first-layer forward (pvk_enc_fwd) rides along, which saves one full read of W1
// (4 of the 52 bytes per wide parameter and step).  Partials go to fpart[slot][32][Wp], summed by pvk_reduce_parts.
struct PvBwdOp;
template <int MODE>
__device__ __forceinline__ void pvn_adam_unit(const PvBwdOp* __restrict__ ops, int bx, int by, int lane_,
                                              const PvStep* __restrict__ st, PvAdam adam, int ntiles);
// Training pass of pvk_enc_bwd (MODE_TRAIN), block-column-major: the waves of a unit group are numbered gw = 0, 1, ...;
// wave gw owns block column w = gw % NWB (32 units) of the subtiles slot, slot + nslots, ... (slot = gw / NWB), so
//   * the unit of work is a 32 x 32 block (12 KiB of W | m | v per wave) instead of a whole subtile: a workgroup's share of a
//     small problem or SNP shard is even to within one block per wave.  Per-CU evenness is what counts: a CU moves about the
//     same bytes per microsecond with 4 or 8 resident waves, and with subtile-sized units 100 k SNPs left some CUs twice the
//     work of others (per-workgroup timeline, tools/trace_wide.py: the kernel ended 25 us after its median workgroup);
//   * the fused forward needs ONE 32 x 32 accumulator per wave instead of NWB of them (48 registers less at width 128), and the
//     waves' forward fragments are disjoint blocks of the partial tile fpart[slot][32][Wp]: no cross-wave reduction;
//   * the sweep is software-pipelined one block deep: the next block (and the genotype words that go with it) is requested
//     before the current one is touched, so a wave always has 12 KiB of loads in flight and never waits for a request it has
//     just issued; waits are for the OLDER block only (gfx9 counts loads and stores on one in-order counter).
// The narrow layers' weight-gradient + Adam units (adam_units > 0) take the first ceil(adam_units / 4) workgroups of the
// launch, one unit per wave, and exit: carried by sweep waves they delayed those by 10-30 us (their dependent loads crawl
// while the chip streams), which was the tail of the kernel at small shapes.
// T = 32-row tiles of the device pass: dpre is [32 T][ldp], Xt holds T word arrays of nblk * 32 entries one after the other, and
// dW sums over the tiles before the ONE Adam update (a batch above 32 rows in one weight pass; the fused forward needs T == 1).
template <int NWB, int FUSE, int T, int MODE = PV_MODE_TRAIN>
__device__ __forceinline__ void pv_enc_bwd_train(const float* __restrict__ W1, const float* __restrict__ mW,
                                                 const float* __restrict__ vW, float* __restrict__ W1o, float* __restrict__ mWo,
                                                 float* __restrict__ vWo, const float* __restrict__ dpre, int ldp,
                                                 const uint64_t* __restrict__ Xt, int nblk, int ngrp, const PvStep* st, PvAdam adam,
                                                 const uint8_t* __restrict__ geno, int64_t row_bytes, const int* __restrict__ rows,
                                                 int next_off, int next_n, float* __restrict__ fpart, int ts,
                                                 const PvBwdOp* __restrict__ adam_ops, int adam_units, float* smem,
                                                 const float* __restrict__ gW = nullptr, const uint64_t* __restrict__ Xn = nullptr) {
    // Xn (FUSE): the next batch's words, gathered by the step head (pv_gather_raw_block); NULL: read from the resident matrix
    // MODE (chunks of a batch above 32 T rows): PV_MODE_GRAD stores / adds (PvStep.acc) 0.5 dW into the dump mWo (plain tiles) and
    // reads no weights at all; PV_MODE_TRAIN_ACC adds the dump gW to this pass's gradient before Adam.
    constexpr int WG_UNITS = NWB * 32;
    static_assert(T == 1 || FUSE == 0, "the fused forward rides on one-tile passes only");
    static_assert(MODE == PV_MODE_TRAIN || FUSE == 0, "the fused forward rides on plain training passes only");
    const bool acc_grads = MODE == PV_MODE_GRAD && st->acc != 0;
    const int tid = threadIdx.x, wave = __builtin_amdgcn_readfirstlane(tid >> 6), lane = tid & 63, c = lane & 31, h = lane >> 5;
    int bx = blockIdx.x, nwg = gridDim.x;
    if (adam_ops) {
        const int awgs = (adam_units + 3) >> 2;
        if (bx < awgs) {
            const int unit = bx * 4 + wave;
            if (unit < adam_units) pvn_adam_unit<PV_MODE_TRAIN>(adam_ops, unit & 15, unit >> 4, lane, st, adam, T);
            return;
        }
        bx -= awgs;
        nwg -= awgs;
    }
    // (all wave-uniform: the tile addresses live in SGPRs and a lane only holds its 16-byte offset)
    const int grp = bx % ngrp, gw = (bx / ngrp) * 4 + wave;
    const int slot = gw / NWB, w = gw - slot * NWB;
    const int nslots = (nwg / ngrp) * 4 / NWB;  // the host sizes the grid so that this divides
    const float alpha = st->alpha;
    struct Block { f32x4 w[4], m[4], v[4], g[4]; };  // (members a MODE does not touch cost nothing)
    auto request = [&](Block& bk, int blk_) {
        const int64_t o = ((int64_t)blk_ * ngrp + grp) * ts + w * 256 + lane;
        const int64_t go = ((int64_t)blk_ * ngrp + grp) * (NWB * 256) + w * 256 + lane;  // gradient dumps are plain tiles
        const f32x4* tW = reinterpret_cast<const f32x4*>(W1) + o;
        const f32x4* tM = reinterpret_cast<const f32x4*>(mW) + o;
        const f32x4* tV = reinterpret_cast<const f32x4*>(vW) + o;
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            if (MODE == PV_MODE_GRAD) {
                if (acc_grads) bk.g[q] = __builtin_nontemporal_load(reinterpret_cast<const f32x4*>(mWo) + go + q * 64);
                continue;
            }
            bk.w[q] = __builtin_nontemporal_load(tW + q * 64);
            bk.m[q] = __builtin_nontemporal_load(tM + q * 64);
            bk.v[q] = __builtin_nontemporal_load(tV + q * 64);
            if (MODE == PV_MODE_TRAIN_ACC) bk.g[q] = __builtin_nontemporal_load(reinterpret_cast<const f32x4*>(gW) + go + q * 64);
        }
    };
    const uint8_t* nrow = nullptr;  // FUSE: lane c's sample of the next batch
    if (FUSE && !Xn && c < next_n) nrow = geno + (int64_t)rows[next_off + c] * row_bytes;
    auto words = [&](int blk_, uint64_t (&xw_)[T], uint64_t& xn_) {
#pragma unroll
        for (int t = 0; t < T; ++t) xw_[t] = Xt[((int64_t)t * nblk + blk_) * 32 + c];
        if (FUSE) xn_ = Xn ? Xn[(int64_t)blk_ * 32 + c] : nrow ? *reinterpret_cast<const uint64_t*>(nrow + (int64_t)blk_ * 8) : 0ull;
    };
    Block bk[2];
    uint64_t xw[T] = {}, xn = 0;
    int blk = slot;
    if (blk < nblk) {  // the first block travels while dpre is staged
        request(bk[0], blk);
        words(blk, xw, xn);
    }
    for (int i = tid; i < 32 * T * ldp; i += 256) {
        const int b = i / ldp, u = i % ldp;
        smem[((u / WG_UNITS) * (32 * T) + b) * WG_UNITS + (u % WG_UNITS)] = dpre[i];
    }
    __syncthreads();
    PV_WT(1, 1);
    const float* dp = smem + grp * (32 * T) * WG_UNITS + w * 32;
    f32x16 facc;
#pragma unroll
    for (int r = 0; r < 16; ++r) facc[r] = 0.f;
    auto block = [&](Block& cur, Block& nxt) {
        const int nb = blk + nslots < nblk ? blk + nslots : blk;  // past the end: a harmless re-read of this block
        request(nxt, nb);
        uint64_t xw2[T], xn2 = 0;
        words(nb, xw2, xn2);
        const uint64_t ns = xn >> (8 * h);
        const uint32_t nlo = (uint32_t)ns, nhi = (uint32_t)(ns >> 32);
        const int64_t o = ((int64_t)blk * ngrp + grp) * ts + w * 256 + lane;
        f32x4* oW = reinterpret_cast<f32x4*>(W1o) + o;
        f32x4* oM = reinterpret_cast<f32x4*>(mWo) + o;
        f32x4* oV = reinterpret_cast<f32x4*>(vWo) + o;
        f32x16 dw;
#pragma unroll
        for (int r = 0; r < 16; ++r) dw[r] = 0.f;
#pragma unroll
        for (int t = 0; t < T; ++t) {
            const uint64_t xs = xw[t] >> (8 * h);
            const uint32_t lo = (uint32_t)xs, hi = (uint32_t)(xs >> 32);
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const float bv = dp[(32 * t + PV_ROW(r, h)) * WG_UNITS + c];  // B[k=b][j=unit]
                dw = __builtin_amdgcn_mfma_f32_32x32x2f32(pv_code(lo, hi, r), bv, dw, 0, 0, 0);
            }
        }
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            if (MODE == PV_MODE_GRAD) {
                f32x4 gq = {0.5f * dw[q * 4 + 0], 0.5f * dw[q * 4 + 1], 0.5f * dw[q * 4 + 2], 0.5f * dw[q * 4 + 3]};
                if (acc_grads) gq += cur.g[q];
                __builtin_nontemporal_store(gq, reinterpret_cast<f32x4*>(mWo) + ((int64_t)blk * ngrp + grp) * (NWB * 256) + w * 256 +
                                                    lane + q * 64);
                continue;
            }
            f32x4 wq = cur.w[q], mq = cur.m[q], vq = cur.v[q];
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                float w_ = wq[e], m_ = mq[e], v_ = vq[e];
                pv_adam(w_, m_, v_, MODE == PV_MODE_TRAIN_ACC ? 0.5f * dw[q * 4 + e] + cur.g[q][e] : 0.5f * dw[q * 4 + e], alpha, adam);
                wq[e] = w_;
                mq[e] = m_;
                vq[e] = v_;
            }
            __builtin_nontemporal_store(wq, oW + q * 64);
            __builtin_nontemporal_store(mq, oM + q * 64);
            __builtin_nontemporal_store(vq, oV + q * 64);
            if (FUSE) {
#pragma unroll
                for (int e = 0; e < 4; ++e)
                    facc = __builtin_amdgcn_mfma_f32_32x32x2f32(pv_code(nlo, nhi, q * 4 + e), wq[e], facc, 0, 0, 0);
            }
        }
        __builtin_amdgcn_sched_barrier(0);  // the next block's requests stay ahead of this block's work, one block deep
#pragma unroll
        for (int t = 0; t < T; ++t) xw[t] = xw2[t];
        xn = xn2;
        blk += nslots;
    };
    while (blk < nblk) {
        block(bk[0], bk[1]);
        if (!(blk < nblk)) break;
        block(bk[1], bk[0]);
    }
    PV_WT(1, 2);
    if (FUSE) {
        // this wave's block of the partial tile fpart[slot][32][Wp] (waves without work contribute their zeros)
        const int ldf = ngrp * WG_UNITS;
        float* p = fpart + (int64_t)slot * (32 * ldf) + grp * WG_UNITS + w * 32 + c;
#pragma unroll
        for (int r = 0; r < 16; ++r) p[PV_ROW(r, h) * ldf] = facc[r];
    }
}

template <int NWB, int MODE, int FUSE, int T = 1>
__global__ __launch_bounds__(256, 2) void pvk_enc_bwd(const float* __restrict__ W1, const float* __restrict__ mW,
                                                   const float* __restrict__ vW, float* __restrict__ W1o,
                                                   float* __restrict__ mWo, float* __restrict__ vWo,
                                                   const float* __restrict__ dpre, int ldp, const uint64_t* __restrict__ Xt,
                                                   int nblk, int ngrp, const PvStep* st, PvAdam adam,
                                                   const uint8_t* __restrict__ geno, int64_t row_bytes,
                                                   const int* __restrict__ rows, int next_off, int next_n,
                                                   float* __restrict__ fpart, int ts, const float* __restrict__ gW,
                                                   const PvBwdOp* __restrict__ adam_ops, int adam_units,
                                                   const uint64_t* __restrict__ Xn = nullptr) {
    constexpr int WG_UNITS = NWB * 32;
    extern __shared__ __attribute__((aligned(16))) float smem[];  // [ngrp][32][WG_UNITS]
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63, c = lane & 31, h = lane >> 5;
    PV_WT(1, 0);
    if constexpr (MODE == PV_MODE_TRAIN || T > 1) {  // several row tiles: the block-column sweep in every mode
        pv_enc_bwd_train<NWB, FUSE, T, MODE>(W1, mW, vW, W1o, mWo, vWo, dpre, ldp, Xt, nblk, ngrp, st, adam, geno, row_bytes, rows,
                                          next_off, next_n, fpart, ts, adam_ops, adam_units, smem, gW, Xn);
        PV_WT(1, 3);
        return;
    }
    // Gradient dump (MODE_GRAD) and Adam on this pass + the dump (MODE_TRAIN_ACC): tiles of a batch above 32 rows.  One
    // wave per subtile, round-robin; a workgroup stays on one unit group (gridDim.x is a multiple of ngrp).
    // (T == 1 here: passes over several row tiles take the block-column sweep above in every mode)
    static_assert(MODE == PV_MODE_TRAIN || FUSE == 0, "fused forward: plain training pass only");
    const int grp = blockIdx.x % ngrp;
    const int slot = __builtin_amdgcn_readfirstlane((int)(blockIdx.x / ngrp) * 4 + wave), nslots = (gridDim.x / ngrp) * 4;
    const float alpha = st->alpha;
    const bool acc_grads = MODE == PV_MODE_GRAD && st->acc != 0;
    for (int i = tid; i < 32 * T * ldp; i += 256) {
        const int b = i / ldp, u = i % ldp;
        smem[((u / WG_UNITS) * (32 * T) + b) * WG_UNITS + (u % WG_UNITS)] = dpre[i];
    }
    __syncthreads();
    for (int blk = slot; blk < nblk; blk += nslots) {
        const int64_t t = (int64_t)blk * ngrp + grp;
        const float* dp = smem + grp * (32 * T) * WG_UNITS;
        uint32_t lo[T], hi[T];
#pragma unroll
        for (int tt = 0; tt < T; ++tt) {
            const uint64_t xw = Xt[((int64_t)tt * nblk + blk) * 32 + c] >> (8 * h);
            lo[tt] = (uint32_t)xw, hi[tt] = (uint32_t)(xw >> 32);
        }
        const f32x4* tW = reinterpret_cast<const f32x4*>(W1) + t * ts;  // interleaved W|m|v tiles, see pvk_dec
        const f32x4* tM = reinterpret_cast<const f32x4*>(mW) + t * ts;
        const f32x4* tV = reinterpret_cast<const f32x4*>(vW) + t * ts;
        f32x4* oW = reinterpret_cast<f32x4*>(W1o) + t * ts;
        f32x4* oM = reinterpret_cast<f32x4*>(mWo) + t * (MODE == PV_MODE_GRAD ? (int64_t)(NWB * 256) : (int64_t)ts);
        f32x4* oV = reinterpret_cast<f32x4*>(vWo) + t * ts;
#pragma unroll
        for (int w = 0; w < NWB; ++w) {
            f32x16 dw;
#pragma unroll
            for (int r = 0; r < 16; ++r) dw[r] = 0.f;
#pragma unroll
            for (int tt = 0; tt < T; ++tt)
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const float bv = dp[(32 * tt + PV_ROW(r, h)) * WG_UNITS + w * 32 + c];  // B[k=b][j=unit]
                    dw = __builtin_amdgcn_mfma_f32_32x32x2f32(pv_code(lo[tt], hi[tt], r), bv, dw, 0, 0, 0);
                }
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const int off = (w * 4 + q) * 64 + lane;
                if (MODE == PV_MODE_GRAD) {
                    f32x4 gq = {0.5f * dw[q * 4 + 0], 0.5f * dw[q * 4 + 1], 0.5f * dw[q * 4 + 2], 0.5f * dw[q * 4 + 3]};
                    if (acc_grads) gq += __builtin_nontemporal_load(oM + off);
                    __builtin_nontemporal_store(gq, oM + off);
                } else {
                    f32x4 wq = __builtin_nontemporal_load(tW + off), mq = __builtin_nontemporal_load(tM + off),
                          vq = __builtin_nontemporal_load(tV + off);
                    const f32x4 ga = __builtin_nontemporal_load(reinterpret_cast<const f32x4*>(gW) + t * (NWB * 256) + off);
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        float w_ = wq[e], m_ = mq[e], v_ = vq[e];
                        pv_adam(w_, m_, v_, 0.5f * dw[q * 4 + e] + ga[e], alpha, adam);
                        wq[e] = w_;
                        mq[e] = m_;
                        vq[e] = v_;
                    }
                    __builtin_nontemporal_store(wq, oW + off);
                    __builtin_nontemporal_store(mq, oM + off);
                    __builtin_nontemporal_store(vq, oV + off);
                }
            }
        }
    }
    PV_WT(1, 3);
}

// ---------------------------------------------------------------------------------------------
// One-shot all-reduce of an exchange buffer among the ranks of an SNP-sharded model (SURVEY 8e "tuned path"; opt-in,
// pvae_p2p_*): every rank owns an inbox in fine-grained device memory that its peers have mapped (hipIpc; over xGMI on a node),
// [parity 2][sender world][words].  A rank writes its n floats into EVERY rank's inbox (its own included) as 8-byte words
// {sequence number, value}: a word arrives whole, so no fence and no separate flag is needed (the idea of RCCL's LL protocol);
// then it waits per word for the sequence number of this exchange and sums the senders in RANK ORDER -- every rank adds the
// same numbers in the same order, so the replicated narrow stack stays bit-identical across ranks.  Two parities: a peer can be
// at most one exchange ahead (it needs this rank's words of exchange k + 1, sent only after exchange k was read), so the slot
// of exchange k + 2 is free when it writes it.  Waits are bounded; a wait that gives up poisons the buffer with NaN and raises
// *status (loudly wrong, not hung).  Vector stores / loads with system scope only.
// ---------------------------------------------------------------------------------------------
#define PV_P2P_MAX_WORLD 16
struct PvPeers { unsigned long long* inbox[PV_P2P_MAX_WORLD]; };
__global__ __launch_bounds__(256) void pvk_p2p_allreduce(float* __restrict__ buf, int n, PvPeers peers, int rank, int world,
                                                         unsigned seq, int parity, int64_t words, int* __restrict__ status,
                                                         int max_spin) {
    // parity: the inbox half of this exchange; it flips with EVERY exchange (the host keeps it apart from the sequence number,
    // which skips 0 when it wraps: derived from seq & 1, the exchanges 0xFFFFFFFF and 1 would share a half)
    const int64_t slot = ((int64_t)parity * world) * words;
    const int tid = blockIdx.x * blockDim.x + threadIdx.x, nth = gridDim.x * blockDim.x;
    for (int i = tid; i < n; i += nth) {
        const unsigned long long word = ((unsigned long long)seq << 32) | __float_as_uint(buf[i]);
        for (int p = 0; p < world; ++p)
            __hip_atomic_store(peers.inbox[p] + slot + (int64_t)rank * words + i, word, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    }
    for (int i = tid; i < n; i += nth) {
        float sum = 0.f;
        bool ok = true;
        for (int r = 0; r < world && ok; ++r) {
            const unsigned long long* src = peers.inbox[rank] + slot + (int64_t)r * words + i;
            unsigned long long w = 0;
            ok = false;
            for (int spin = 0; spin < max_spin; ++spin) {
                w = __hip_atomic_load(src, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
                if ((unsigned)(w >> 32) == seq) { ok = true; break; }
                __builtin_amdgcn_s_sleep(2);
            }
            sum += __uint_as_float((unsigned)w);
        }
        if (!ok) {
            sum = __builtin_nanf("");
            *status = 1;
        }
        buf[i] = sum;
    }
}

// ---------------------------------------------------------------------------------------------
// N(0,1) draws for the reparameterisation (K3: TF's Philox stream is not reproducible outside TF, any
// standard-normal generator is admissible).  Philox4x32-10 + Box-Muller, counter = (call, index).
// ---------------------------------------------------------------------------------------------
__device__ __forceinline__ void pv_philox(uint32_t c0, uint32_t c1, uint32_t c2, uint32_t c3, uint32_t k0, uint32_t k1,
                                          uint32_t out[4]) {
#pragma unroll
    for (int i = 0; i < 10; ++i) {
        const uint64_t p0 = (uint64_t)0xD2511F53u * c0, p1 = (uint64_t)0xCD9E8D57u * c2;
        const uint32_t n0 = (uint32_t)(p1 >> 32) ^ c1 ^ k0, n1 = (uint32_t)p1;
        const uint32_t n2 = (uint32_t)(p0 >> 32) ^ c3 ^ k1, n3 = (uint32_t)p0;
        c0 = n0; c1 = n1; c2 = n2; c3 = n3;
        k0 += 0x9E3779B9u; k1 += 0xBB67AE85u;
    }
    out[0] = c0; out[1] = c1; out[2] = c2; out[3] = c3;
}

__global__ void pvk_gen_eps(float* __restrict__ eps, int n, uint64_t seed, uint64_t* __restrict__ counter) {
    const uint64_t call = *counter;  // one workgroup (it advances the counter behind its barrier)
    for (int i = threadIdx.x; i < n; i += blockDim.x) {
        uint32_t o[4];
        pv_philox((uint32_t)i, 0u, (uint32_t)call, (uint32_t)(call >> 32), (uint32_t)seed, (uint32_t)(seed >> 32), o);
        const float u1 = ((float)(o[0] >> 8) + 0.5f) * (1.0f / 16777216.0f);
        const float u2 = ((float)(o[1] >> 8) + 0.5f) * (1.0f / 16777216.0f);
        eps[i] = sqrtf(-2.f * logf(u1)) * cosf(6.28318530717958647692f * u2);
    }
    __syncthreads();
    if (threadIdx.x == 0) *counter = call + 1;
}

// pvk_step_begin + pvk_gen_eps in one launch of 1024 threads: the two used to be back-to-back 4 us kernels at
// the head of every step
__global__ __launch_bounds__(1024) void pvk_step_begin_eps(PvStep* st, const int* __restrict__ n_total_ptr, int n_total_val,
                                                           int batch, int step, int train, float b1, float b2,
                                                           float* __restrict__ eps, int n_eps, uint64_t seed,
                                                           uint64_t* __restrict__ counter) {
    const int i = threadIdx.x;
    const uint64_t call = *counter;
    if (i == 0) {
        int remaining = (n_total_ptr ? *n_total_ptr : n_total_val) - step * batch;
        int nrows = remaining < batch ? remaining : batch;
        if (nrows < 0) nrows = 0;
        st->nrows = nrows;
        st->inv_b = nrows > 0 ? 1.f / (float)nrows : 0.f;
        st->step_in_epoch = step;
        st->acc = 0;
        if (train) {
            int t = st->t + 1;
            st->t = t;
            float b1p = powf(b1, (float)t), b2p = powf(b2, (float)t);
            st->alpha = st->lr * sqrtf(1.f - b2p) / (1.f - b1p);
        }
    }
    for (int j = i; j < n_eps; j += 1024) {
        uint32_t o[4];
        pv_philox((uint32_t)j, 0u, (uint32_t)call, (uint32_t)(call >> 32), (uint32_t)seed, (uint32_t)(seed >> 32), o);
        const float u1 = ((float)(o[0] >> 8) + 0.5f) * (1.0f / 16777216.0f);
        const float u2 = ((float)(o[1] >> 8) + 0.5f) * (1.0f / 16777216.0f);
        eps[j] = sqrtf(-2.f * logf(u1)) * cosf(6.28318530717958647692f * u2);
    }
    __syncthreads();
    if (i == 0) *counter = call + 1;
}

// The head of a step of the C-side epoch loop in ONE launch: the last workgroup writes the per-step scalars and the
// noise (as pvk_step_begin_eps), all others gather/transpose the batch (as pvk_gather_transpose; they derive the
// batch's row count from the same arguments instead of reading it back from PvStep).  Saves one ~4.7 us launch slot.
// The first nred workgroups (0 = none) instead sum the first-layer partial tiles the previous step's pvk_enc_bwd left
// behind (pvk_reduce_parts: P = 0.5 * sum of fpart), which is independent of the gather: one launch slot less per step.
__global__ __launch_bounds__(256) void pvk_step_begin_gather(PvStep* st, const int* __restrict__ n_total_ptr, int batch, int step,
                                                             int train, float b1, float b2, float* __restrict__ eps,
                                                             int n_eps, uint64_t seed, uint64_t* __restrict__ counter,
                                                             const uint8_t* __restrict__ geno, int64_t row_bytes,
                                                             const int* __restrict__ rows, int row_offset, int nblk,
                                                             uint64_t* __restrict__ Xt, int nred, const float* __restrict__ fpart,
                                                             int nparts, int nP, float* __restrict__ P, int ngather, int ngather_next,
                                                             int next_off, int next_n, uint64_t* __restrict__ Xt_next,
                                                             uint64_t* __restrict__ Xrn) {
    // blocks [0, nred): partial-tile sums; then ngather blocks (0 when the previous step's head already did it) that gather /
    // transpose THIS batch into Xt; then ngather_next blocks that gather / transpose the NEXT batch (rows[next_off ..], next_n of
    // them) into Xt_next and leave its words in Xrn, the compact operand of the forward fused into this step's wide encoder
    // backward: from the resident matrix that kernel touched 32 cache lines per block and wave (8 bytes used of each; +3 % HBM
    // traffic, profiles/r02_v6_pmc_traffic_C4.json), from Xrn it reads 256 contiguous bytes; last block: scalars + noise
    if ((int)blockIdx.x < nred) {
        pv_reduce_block(fpart, nparts, nP, 0.5f, P, nullptr, nullptr, blockIdx.x);
        return;
    }
    const int remaining = *n_total_ptr - step * batch;
    const int nrows = max(0, min(remaining, batch));
    if ((int)blockIdx.x < nred + ngather) {
        pv_gather_block(geno, row_bytes, rows, nrows, row_offset, nblk, Xt, nullptr, (int)blockIdx.x - nred);
        return;
    }
    if ((int)blockIdx.x < nred + ngather + ngather_next) {
        pv_gather_block(geno, row_bytes, rows, next_n, next_off, nblk, Xt_next, Xrn, (int)blockIdx.x - nred - ngather);
        return;
    }
    const int i = threadIdx.x;
    const uint64_t call = *counter;
    if (i == 0) {
        st->nrows = nrows;
        st->inv_b = nrows > 0 ? 1.f / (float)nrows : 0.f;
        st->step_in_epoch = step;
        st->acc = 0;
        if (train) {
            int t = st->t + 1;
            st->t = t;
            float b1p = powf(b1, (float)t), b2p = powf(b2, (float)t);
            st->alpha = st->lr * sqrtf(1.f - b2p) / (1.f - b1p);
        }
    }
    for (int j = i; j < n_eps; j += 256) {
        uint32_t o[4];
        pv_philox((uint32_t)j, 0u, (uint32_t)call, (uint32_t)(call >> 32), (uint32_t)seed, (uint32_t)(seed >> 32), o);
        const float u1 = ((float)(o[0] >> 8) + 0.5f) * (1.0f / 16777216.0f);
        const float u2 = ((float)(o[1] >> 8) + 0.5f) * (1.0f / 16777216.0f);
        eps[j] = sqrtf(-2.f * logf(u1)) * cosf(6.28318530717958647692f * u2);
    }
    __syncthreads();
    if (i == 0) *counter = call + 1;
}


// The NEXT batch of the epoch loop, gathered on a second stream while the narrow forward chain runs (HBM idles there): the
// transposed copy into Xt_next and the compact words into Xrn (operand of the forward fused into pvk_enc_bwd).  grid = ngb blocks.
__global__ __launch_bounds__(256) void pvk_gather_next(const uint8_t* __restrict__ geno, int64_t row_bytes,
                                                       const int* __restrict__ rows, int next_off, int next_n, int nblk,
                                                       uint64_t* __restrict__ Xt_next, uint64_t* __restrict__ Xrn) {
    pv_gather_block(geno, row_bytes, rows, next_n, next_off, nblk, Xt_next, Xrn, (int)blockIdx.x);
}

// ---------------------------------------------------------------------------------------------
// Layout conversion between Keras row-major tensors and the fragment-native tiled layout.
// kind 0: encoder W1 (Keras [L][W]);  kind 1: decoder V (Keras [W][L]).  dir 0: keras->tiled, 1: tiled->keras.
// One thread per tiled element (padding elements are written as zero on import).
// ---------------------------------------------------------------------------------------------
__global__ void pvk_convert_wide(float* __restrict__ tiled, float* __restrict__ keras, int kind, int dir, int64_t L,
                                 int W, int nblk, int ngrp, int NWB, int tile_stride /* floats between subtiles */) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const int64_t total = (int64_t)nblk * ngrp * NWB * 1024;
    if (i >= total) return;
    const int e = i & 3, lane = (i >> 2) & 63, q = (i >> 8) & 3;
    const int64_t sub = i >> 10;  // (blk*ngrp + grp)*NWB + wbl
    const int wbl = sub % NWB;
    const int grp = (sub / NWB) % ngrp;
    const int64_t blk = sub / ((int64_t)NWB * ngrp);
    const int r = q * 4 + e, h = lane >> 5, c = lane & 31;
    int64_t l;
    int u;
    if (kind == 0) {
        l = blk * 32 + PV_ROW(r, h);
        u = (grp * NWB + wbl) * 32 + c;
    } else {
        u = (grp * NWB + wbl) * 32 + PV_ROW(r, h);
        l = blk * 32 + c;
    }
    const bool inside = l < L && u < W;
    const int64_t kidx = kind == 0 ? l * W + u : (int64_t)u * L + l;
    const int64_t ti = (i / (NWB * 1024)) * tile_stride + (i % (NWB * 1024));
    if (dir == 0)
        tiled[ti] = inside ? keras[kidx] : 0.f;
    else if (inside)
        keras[kidx] = tiled[ti];
}
