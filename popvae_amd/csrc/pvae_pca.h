// PCA front end on the resident 2-bit matrix (reference popvae.py:528-533 calls
// allel.pca(dc.T, scaler, n_components) = per-site scaler + SVD of the scaled N x L matrix).
//
// The SVD of the scaled matrix Xs (N x L) is taken through its Gram matrix G = Xs Xs^T (N x N, N = samples):
// G = U S^2 U^T, coordinates = U S.  Everything below is float64 (the reference's SVD is float64):
//   pvk_site_scale  per-site genotype counts over all samples (integer, exact) -> mean, std, and the two
//                   constants (h, t0) with scaled value(code) = code * h + t0
//   pvk_gram        G partials with v_mfma_f64_16x16x4_f64; operands are decoded from the 2-bit words in
//                   registers; a workgroup owns a 128 x 128 tile of G and a slice of the SNP axis
//   pvk_gram_reduce fixed-order sum of the SNP-axis slices + mirror of the upper triangle
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

typedef double f64x4 __attribute__((ext_vector_type(4)));

enum { PV_SCALE_CENTER = 0, PV_SCALE_PATTERSON = 1, PV_SCALE_STANDARD = 2 };

// ---------------------------------------------------------------------------------------------
// One workgroup per 16 SNP blocks (512 SNPs = one 128-byte line of every sample row).  Thread t reads word
// (row = t/16 + 16*it, block = t%16): 16 rows x 128 B per iteration, coalesced.  Low and high bits of the codes
// are summed in byte-wide fields (255 rows before a flush), then in LDS with integer atomics: exact, order-free.
// sc[l] = {h, t0}; sites at or beyond n_snps get {0, 0} so that padding contributes nothing to G.
// ---------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void pvk_site_scale(const uint64_t* __restrict__ geno, int64_t row_words, int n_rows,
                                                      int nblk, int64_t n_snps, int scaler, double2* __restrict__ sc,
                                                      double* __restrict__ mean_out, double* __restrict__ std_out,
                                                      int* __restrict__ bad_sites) {
    __shared__ unsigned cnt[16][32][2];  // [block][snp][n1 | n2]
    const int tid = threadIdx.x, bl = tid & 15, rl = tid >> 4;
    const int kb = blockIdx.x * 16 + bl;
    for (int i = tid; i < 16 * 32 * 2; i += 256) (&cnt[0][0][0])[i] = 0u;
    __syncthreads();
    const uint64_t M = 0x0101010101010101ull;
    uint64_t lo[4] = {0, 0, 0, 0}, hi[4] = {0, 0, 0, 0};
    int pending = 0;
    auto flush = [&]() {
#pragma unroll
        for (int s = 0; s < 4; ++s)
#pragma unroll
            for (int b = 0; b < 8; ++b) {
                // field b of sub-word s is SNP 4*b + s of the block; code 1 <-> (lo=1, hi=0), code 2 <-> hi=1
                const unsigned l_ = (unsigned)(lo[s] >> (8 * b)) & 255u, h_ = (unsigned)(hi[s] >> (8 * b)) & 255u;
                if (l_) atomicAdd(&cnt[bl][4 * b + s][0], l_);
                if (h_) atomicAdd(&cnt[bl][4 * b + s][1], h_);
            }
#pragma unroll
        for (int s = 0; s < 4; ++s) lo[s] = hi[s] = 0;
        pending = 0;
    };
    if (kb < nblk) {
        for (int i = rl; i < n_rows; i += 16) {
            const uint64_t w = geno[(int64_t)i * row_words + kb];
#pragma unroll
            for (int s = 0; s < 4; ++s) {
                lo[s] += (w >> (2 * s)) & M;
                hi[s] += (w >> (2 * s + 1)) & M;
            }
            if (++pending == 255) flush();
        }
        flush();
    }
    __syncthreads();
    for (int j = tid; j < 512; j += 256) {
        const int b = j >> 5, s = j & 31;
        const int64_t l = ((int64_t)blockIdx.x * 16 + b) * 32 + s;
        if (blockIdx.x * 16 + b >= nblk) continue;
        double2 o = {0.0, 0.0};
        if (l < n_snps) {
            // cnt[..][0] counts set low bits = code 1 or 3; code 3 never occurs (rejected at load time)
            const double n1 = (double)cnt[b][s][0], n2 = (double)cnt[b][s][1], n = (double)n_rows, n0 = n - n1 - n2;
            const double mean = (0.5 * (n1 + 2.0 * n2)) / n;  // mean of the VALUES code/2 (popvae.py:237 feeds dc*0.5)
            double sd = 1.0;
            if (scaler == PV_SCALE_PATTERSON) {
                const double p = mean / 2.0;  // allel's ploidy = 2 applied to the already halved values: kept as is
                sd = sqrt(p * (1.0 - p));
            } else if (scaler == PV_SCALE_STANDARD) {
                const double d0 = 0.0 - mean, d1 = 0.5 - mean, d2 = 1.0 - mean;
                sd = sqrt((n0 * d0 * d0 + n1 * d1 * d1 + n2 * d2 * d2) / n);
            }
            if (!(sd > 0.0)) atomicAdd(bad_sites, 1);
            o.x = 0.5 / sd;
            o.y = -mean / sd;
            if (mean_out) mean_out[l] = mean;
            if (std_out) std_out[l] = sd;
        }
        sc[l] = o;
    }
}

// ---------------------------------------------------------------------------------------------
// G tile: workgroup (tile pair, SNP slice).  4 waves = 2 x 2 sub-tiles of 64 x 64; a wave keeps 4 x 4 MFMA
// accumulators (16 x 16 f64 each, 64 f64 per lane).  Per 32-SNP block a lane holds the 2-bit words of 4 A rows and
// 4 B rows (row = lane & 15 of each 16-row group); k = lane >> 4 picks the SNP of each 4-SNP MFMA step.
// A[i][k] = B^T: both operands are the scaled value of (sample, SNP).
// part layout: [slice][tile pair][128][128].
// ---------------------------------------------------------------------------------------------
__device__ __forceinline__ double pv_scaled(uint64_t w, int sh, double h, double t0) {
    return fma((double)(unsigned)((w >> sh) & 3u), h, t0);
}

__global__ __launch_bounds__(256) void pvk_gram(const uint64_t* __restrict__ geno, int64_t row_words, int n_rows, int nblk,
                                                const double2* __restrict__ sc, const int2* __restrict__ pairs,
                                                int n_pairs, int blocks_per_slice, double* __restrict__ part) {
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63, r16 = lane & 15, kq = lane >> 4;
    const int pair = blockIdx.x % n_pairs, slice = blockIdx.x / n_pairs;
    const int2 tt = pairs[pair];
    const int i0 = tt.x * 128 + (wave >> 1) * 64, j0 = tt.y * 128 + (wave & 1) * 64;
    const int kb0 = slice * blocks_per_slice, kb1 = min(nblk, kb0 + blocks_per_slice);

    const uint64_t* pa[4];
    const uint64_t* pb[4];
#pragma unroll
    for (int a = 0; a < 4; ++a) {
        pa[a] = geno + (int64_t)min(i0 + 16 * a + r16, n_rows - 1) * row_words;  // rows past the end: clamped, discarded later
        pb[a] = geno + (int64_t)min(j0 + 16 * a + r16, n_rows - 1) * row_words;
    }
    f64x4 acc[4][4];
#pragma unroll
    for (int a = 0; a < 4; ++a)
#pragma unroll
        for (int b = 0; b < 4; ++b) acc[a][b] = (f64x4){0.0, 0.0, 0.0, 0.0};

    uint64_t wa[4], wb[4], na[4], nb[4];
    if (kb0 < kb1) {
#pragma unroll
        for (int a = 0; a < 4; ++a) { na[a] = pa[a][kb0]; nb[a] = pb[a][kb0]; }
    }
    for (int kb = kb0; kb < kb1; ++kb) {
#pragma unroll
        for (int a = 0; a < 4; ++a) { wa[a] = na[a] >> (2 * kq); wb[a] = nb[a] >> (2 * kq); }
        const int kn = min(kb + 1, kb1 - 1);  // prefetch the next block's words under this block's MFMAs
#pragma unroll
        for (int a = 0; a < 4; ++a) { na[a] = pa[a][kn]; nb[a] = pb[a][kn]; }
        const double2* s = sc + (int64_t)kb * 32 + kq;
#pragma unroll
        for (int st = 0; st < 8; ++st) {
            const double2 ht = s[4 * st];  // SNP 4*st + kq of the block
            double va[4], vb[4];
#pragma unroll
            for (int a = 0; a < 4; ++a) {
                va[a] = pv_scaled(wa[a], 8 * st, ht.x, ht.y);
                vb[a] = pv_scaled(wb[a], 8 * st, ht.x, ht.y);
            }
#pragma unroll
            for (int a = 0; a < 4; ++a)
#pragma unroll
                for (int b = 0; b < 4; ++b)
                    acc[a][b] = __builtin_amdgcn_mfma_f64_16x16x4f64(va[a], vb[b], acc[a][b], 0, 0, 0);
        }
    }
    // C/D of v_mfma_f64_16x16x4_f64: col = lane & 15, row = (lane >> 4) + 4 * reg
    double* out = part + ((int64_t)slice * n_pairs + pair) * (128 * 128) + ((wave >> 1) * 64) * 128 + (wave & 1) * 64;
#pragma unroll
    for (int a = 0; a < 4; ++a)
#pragma unroll
        for (int b = 0; b < 4; ++b)
#pragma unroll
            for (int r = 0; r < 4; ++r) out[(16 * a + kq + 4 * r) * 128 + 16 * b + r16] = acc[a][b][r];
}

// one thread per element of an upper-triangle tile; slices are summed in slice order
__global__ __launch_bounds__(256) void pvk_gram_reduce(const double* __restrict__ part, const int2* __restrict__ pairs,
                                                       int n_pairs, int n_slices, int n, double* __restrict__ G) {
    const int pair = blockIdx.y;
    const int e = blockIdx.x * 256 + threadIdx.x;  // 0 .. 128*128-1
    const int2 tt = pairs[pair];
    const int i = tt.x * 128 + (e >> 7), j = tt.y * 128 + (e & 127);
    if (i >= n || j >= n) return;
    double s = 0.0;
    for (int k = 0; k < n_slices; ++k) s += part[((int64_t)k * n_pairs + pair) * (128 * 128) + e];
    if (tt.x == tt.y) {
        G[(int64_t)i * n + j] = s;  // diagonal tiles are computed in full
    } else {
        G[(int64_t)i * n + j] = s;
        G[(int64_t)j * n + i] = s;
    }
}
