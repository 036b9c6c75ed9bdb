/* pvae_prep.h -- C ABI of libpvae_prep.so: the host-side genotype front end of popvae_amd.
 *
 * Every entry point replaces a piece of the reference's integer path in /root/reference/popvae.py
 * (the reference has no FFI of its own: these are the calls a maintainer would bind from popvae.py
 * through ctypes).  All functions return 0 on success and a negative code on
 * failure; pvae_prep_last_error() returns the message of the calling thread's last failure.
 * Buffers are caller-owned.  Plain C types only.
 */
#ifndef PVAE_PREP_H
#define PVAE_PREP_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct pvae_rng pvae_rng; /* numpy legacy RandomState (MT19937) */
typedef struct pvae_vcf pvae_vcf;

const char* pvae_prep_last_error(void);

/* np.random.seed(seed) -- popvae.py:192 */
pvae_rng* pvae_rng_create(uint32_t seed);
void pvae_rng_destroy(pvae_rng* r);
void pvae_rng_seed(pvae_rng* r, uint32_t seed);
/* legacy random_sample() double / raw 32-bit draw (self-tests) */
double pvae_rng_double(pvae_rng* r);
uint32_t pvae_rng_uint32(pvae_rng* r);
/* np.random.binomial(n, p[i]) for i < count -- popvae.py:235 (inversion branch, n*p <= 30) */
int pvae_rng_binomial(pvae_rng* r, int64_t n, const double* p, int64_t count, int64_t* out);
/* np.random.permutation(n) */
int pvae_rng_permutation(pvae_rng* r, int64_t n, int64_t* out);
/* np.random.choice(range(n), k, replace=False) -- popvae.py:252,257,264 */
int pvae_rng_choice_noreplace(pvae_rng* r, int64_t n, int64_t k, int64_t* out);

/* allel.read_vcf(infile)["calldata/GT"], ["samples"] -- popvae.py:202-204 (.vcf and .vcf.gz) */
int pvae_vcf_read(const char* path, pvae_vcf** out);
void pvae_vcf_free(pvae_vcf* v);
int64_t pvae_vcf_n_variants(const pvae_vcf* v);
int64_t pvae_vcf_n_samples(const pvae_vcf* v);
const char* pvae_vcf_sample(const pvae_vcf* v, int64_t i);
const int8_t* pvae_vcf_gt(const pvae_vcf* v); /* int8 [n_variants][n_samples][2], -1 = missing */

/* count_alleles / is_biallelic / derived count / "singleton" filter / af -- popvae.py:214-232.
 * keep[L0] (0/1), af[L0] (valid where keep), *n_keep = number of kept sites. */
int pvae_prep_filter(const int8_t* gen, int64_t L0, int64_t N, uint8_t* keep, double* af, int64_t* n_keep);
/* to_allele_counts[:, :, 1] + imputation loop + transpose + (x0.5 as a 2-bit code) -- popvae.py:220-238.
 * packed: [N][row_bytes], SNP j at bits 2*(j%4) of byte j/4, code = derived-allele count (value = code/2);
 * row_bytes >= ceil(n_keep/32)*8. */
int pvae_prep_impute_pack(const int8_t* gen, int64_t L0, int64_t N, const uint8_t* keep, const double* af,
                          pvae_rng* rng, uint8_t* packed, int64_t row_bytes, int64_t* n_imputed);
/* dc[:, cols] on the packed matrix -- popvae.py:252 */
int pvae_prep_gather_columns(const uint8_t* src, int64_t N, int64_t src_row_bytes, const int64_t* cols,
                             int64_t n_cols, uint8_t* dst, int64_t dst_row_bytes);

#ifdef __cplusplus
}
#endif
#endif
