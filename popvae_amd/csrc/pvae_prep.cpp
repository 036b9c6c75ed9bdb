// Host-side genotype front end for popvae_amd: VCF GT reader, allele counting / site filters,
// seeded binomial imputation and 2-bit packing, train/validation split.
//
// This is the product's implementation of the reference's integer path
// (/root/reference/popvae.py:201-269).  It must be BIT-EXACT under the same seed, so it carries its
// own MT19937 + the numpy *legacy* RandomState algorithms the reference reaches through
// np.random.seed / np.random.binomial / np.random.choice (popvae.py:192,235,252,257,264).
// Exported entry points are declared in include/pvae_prep.h.
#include <zlib.h>

#include <algorithm>
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>

#include "../../include/pvae_prep.h"

namespace {

thread_local std::string g_err;
int fail(int code, const std::string& msg) {
    g_err = msg;
    return code;
}

}  // namespace

// ---------------------------------------------------------------------------------------------
// numpy legacy RandomState (MT19937)
// ---------------------------------------------------------------------------------------------
struct pvae_rng {
    uint32_t key[624];
    int pos;
    // legacy binomial cache (numpy binomial_t); values are a pure function of (n, p)
    bool has_binomial;
    int64_t nsave;
    double psave, q, r, c;
    int64_t m;

    void seed(uint32_t s) {  // numpy mt19937_seed (init_genrand)
        for (int i = 0; i < 624; ++i) {
            key[i] = s;
            s = 1812433253u * (s ^ (s >> 30)) + (uint32_t)i + 1u;
        }
        pos = 624;
        has_binomial = false;
    }
    void gen() {
        const uint32_t UPPER = 0x80000000u, LOWER = 0x7fffffffu, MAT = 0x9908b0dfu;
        int kk;
        uint32_t y;
        for (kk = 0; kk < 624 - 397; ++kk) {
            y = (key[kk] & UPPER) | (key[kk + 1] & LOWER);
            key[kk] = key[kk + 397] ^ (y >> 1) ^ ((y & 1u) ? MAT : 0u);
        }
        for (; kk < 623; ++kk) {
            y = (key[kk] & UPPER) | (key[kk + 1] & LOWER);
            key[kk] = key[kk + (397 - 624)] ^ (y >> 1) ^ ((y & 1u) ? MAT : 0u);
        }
        y = (key[623] & UPPER) | (key[0] & LOWER);
        key[623] = key[396] ^ (y >> 1) ^ ((y & 1u) ? MAT : 0u);
        pos = 0;
    }
    inline uint32_t next32() {
        if (pos == 624) gen();
        uint32_t y = key[pos++];
        y ^= (y >> 11);
        y ^= (y << 7) & 0x9d2c5680u;
        y ^= (y << 15) & 0xefc60000u;
        y ^= (y >> 18);
        return y;
    }
    inline uint64_t next64() {
        uint64_t hi = next32();
        return (hi << 32) | next32();
    }
    inline double next_double() {  // 53-bit: (a >> 5, b >> 6)
        int32_t a = next32() >> 5, b = next32() >> 6;
        return (a * 67108864.0 + b) / 9007199254740992.0;
    }
    // numpy random_interval: masked rejection on [0, max]
    uint64_t interval(uint64_t max) {
        if (max == 0) return 0;
        uint64_t mask = max, value;
        mask |= mask >> 1;
        mask |= mask >> 2;
        mask |= mask >> 4;
        mask |= mask >> 8;
        mask |= mask >> 16;
        mask |= mask >> 32;
        if (max <= 0xffffffffull) {
            while ((value = (next32() & mask)) > max) {
            }
        } else {
            while ((value = (next64() & mask)) > max) {
            }
        }
        return value;
    }
    // numpy legacy_random_binomial_inversion
    int64_t binomial_inversion(int64_t n, double p) {
        double qq, qn, np_, px, U;
        int64_t X, bound;
        if (!has_binomial || nsave != n || psave != p) {
            nsave = n;
            psave = p;
            has_binomial = true;
            q = qq = 1.0 - p;
            r = qn = std::exp(n * std::log(qq));
            c = np_ = n * p;
            m = bound = (int64_t)std::min((double)n, np_ + 10.0 * std::sqrt(np_ * qq + 1));
        } else {
            qq = q;
            qn = r;
            np_ = c;
            bound = m;
        }
        (void)np_;
        X = 0;
        px = qn;
        U = next_double();
        while (U > px) {
            X++;
            if (X > bound) {
                X = 0;
                px = qn;
                U = next_double();
            } else {
                U -= px;
                px = ((n - X + 1) * p * px) / (X * qq);
            }
        }
        return X;
    }
    // numpy legacy_random_binomial for n*min(p,1-p) <= 30 (always true for n = 2)
    int64_t binomial_small(int64_t n, double p) {
        if (p <= 0.5) return binomial_inversion(n, p);
        return n - binomial_inversion(n, 1.0 - p);
    }
};

extern "C" {

const char* pvae_prep_last_error(void) { return g_err.c_str(); }

pvae_rng* pvae_rng_create(uint32_t seed) {
    pvae_rng* r = new pvae_rng();
    r->seed(seed);
    return r;
}
void pvae_rng_destroy(pvae_rng* r) { delete r; }
void pvae_rng_seed(pvae_rng* r, uint32_t seed) { r->seed(seed); }
double pvae_rng_double(pvae_rng* r) { return r->next_double(); }
uint32_t pvae_rng_uint32(pvae_rng* r) { return r->next32(); }

int pvae_rng_binomial(pvae_rng* r, int64_t n, const double* p, int64_t count, int64_t* out) {
    for (int64_t i = 0; i < count; ++i) {
        double pi = p[i];
        if (!(pi >= 0.0 && pi <= 1.0)) return fail(-2, "binomial: p outside [0,1] or NaN");
        if (n * std::min(pi, 1.0 - pi) > 30.0) return fail(-3, "binomial: n*p > 30 (BTPE) not implemented");
        out[i] = r->binomial_small(n, pi);
    }
    return 0;
}

// np.random.permutation(n): arange + legacy shuffle (i from n-1 down to 1, j = interval(i))
int pvae_rng_permutation(pvae_rng* r, int64_t n, int64_t* out) {
    for (int64_t i = 0; i < n; ++i) out[i] = i;
    for (int64_t i = n - 1; i >= 1; --i) {
        int64_t j = (int64_t)r->interval((uint64_t)i);
        std::swap(out[i], out[j]);
    }
    return 0;
}

// np.random.choice(range(n), k, replace=False) == permutation(n)[:k]   (popvae.py:252,257,264)
int pvae_rng_choice_noreplace(pvae_rng* r, int64_t n, int64_t k, int64_t* out) {
    if (k > n || k < 0) return fail(-2, "choice: cannot take a larger sample than population when replace=False");
    std::vector<int64_t> perm((size_t)n);
    pvae_rng_permutation(r, n, perm.data());
    std::copy(perm.begin(), perm.begin() + k, out);
    return 0;
}

// ---------------------------------------------------------------------------------------------
// VCF GT reader (stands in for allel.read_vcf, popvae.py:202-204)
// ---------------------------------------------------------------------------------------------
struct pvae_vcf {
    std::vector<std::string> samples;
    std::vector<int8_t> gt;  // [L0][N][2], -1 missing
    int64_t n_variants = 0;
};

static bool read_line(gzFile f, std::string& line, std::vector<char>& buf) {
    line.clear();
    for (;;) {
        if (!gzgets(f, buf.data(), (int)buf.size())) return !line.empty();
        size_t n = strlen(buf.data());
        line.append(buf.data(), n);
        if (n && buf[n - 1] == '\n') return true;
        if (gzeof(f)) return true;
    }
}

int pvae_vcf_read(const char* path, pvae_vcf** out) {
    gzFile f = gzopen(path, "rb");
    if (!f) return fail(-1, std::string("cannot open ") + path);
    gzbuffer(f, 1 << 20);
    pvae_vcf* v = new pvae_vcf();
    std::string line;
    std::vector<char> buf(1 << 20);
    size_t N = 0;
    bool have_header = false;
    while (read_line(f, line, buf)) {
        while (!line.empty() && (line.back() == '\n' || line.back() == '\r')) line.pop_back();
        if (line.empty()) continue;
        if (line[0] == '#') {
            if (line.compare(0, 6, "#CHROM") == 0) {
                size_t p = 0;
                int col = 0;
                while (p <= line.size()) {
                    size_t e = line.find('\t', p);
                    if (e == std::string::npos) e = line.size();
                    if (col >= 9 && e > p) v->samples.emplace_back(line, p, e - p);
                    p = e + 1;
                    ++col;
                }
                N = v->samples.size();
                have_header = true;
            }
            continue;
        }
        if (!have_header) {
            delete v;
            gzclose(f);
            return fail(-2, "VCF record before #CHROM header");
        }
        // locate FORMAT (col 8) and the GT index inside it
        const char* s = line.c_str();
        const char* end = s + line.size();
        const char* p = s;
        int col = 0;
        while (col < 8 && p < end) {
            if (*p == '\t') ++col;
            ++p;
        }
        if (col < 8) continue;
        int gt_idx = -1, fi = 0;
        const char* fs = p;
        while (p <= end) {
            if (p == end || *p == ':' || *p == '\t') {
                size_t len = p - fs;
                while (len && (fs[len - 1] == ' ')) --len;  // pyRAD pads INFO/FORMAT with blanks
                if (len == 2 && fs[0] == 'G' && fs[1] == 'T') gt_idx = fi;
                ++fi;
                fs = p + 1;
                if (p == end || *p == '\t') break;
            }
            ++p;
        }
        if (gt_idx < 0) {
            delete v;
            gzclose(f);
            return fail(-3, "VCF record without GT in FORMAT");
        }
        if (p < end) ++p;  // past the tab after FORMAT
        size_t base = v->gt.size();
        v->gt.resize(base + N * 2, (int8_t)-1);
        for (size_t i = 0; i < N && p < end; ++i) {
            // field i spans [p, tab)
            const char* fe = (const char*)memchr(p, '\t', end - p);
            if (!fe) fe = end;
            const char* q = p;
            for (int k = 0; k < gt_idx && q < fe; ++q)
                if (*q == ':') ++k;
            // parse up to two alleles
            int which = 0;
            while (q < fe && *q != ':' && which < 2) {
                if (*q == '.') {
                    v->gt[base + i * 2 + which] = -1;
                    ++q;
                } else if (*q >= '0' && *q <= '9') {
                    int a = 0;
                    while (q < fe && *q >= '0' && *q <= '9') a = a * 10 + (*q++ - '0');
                    v->gt[base + i * 2 + which] = (int8_t)std::min(a, 127);
                } else {
                    ++q;
                    continue;
                }
                ++which;
                if (q < fe && (*q == '/' || *q == '|')) ++q;
            }
            p = fe < end ? fe + 1 : end;
        }
        v->n_variants++;
    }
    gzclose(f);
    *out = v;
    return 0;
}
void pvae_vcf_free(pvae_vcf* v) { delete v; }
int64_t pvae_vcf_n_variants(const pvae_vcf* v) { return v->n_variants; }
int64_t pvae_vcf_n_samples(const pvae_vcf* v) { return (int64_t)v->samples.size(); }
const char* pvae_vcf_sample(const pvae_vcf* v, int64_t i) { return v->samples[(size_t)i].c_str(); }
const int8_t* pvae_vcf_gt(const pvae_vcf* v) { return v->gt.data(); }

// ---------------------------------------------------------------------------------------------
// Site filters (popvae.py:214-229): biallelic (exactly two alleles observed), derived = allele 1,
// drop derived count <= 2.  keep[l] in {0,1}; af[l] = dc_all / (2 * n_called_samples) (popvae.py:232)
// ---------------------------------------------------------------------------------------------
int pvae_prep_filter(const int8_t* gen, int64_t L0, int64_t N, uint8_t* keep, double* af, int64_t* n_keep) {
    int64_t kept = 0;
    for (int64_t l = 0; l < L0; ++l) {
        const int8_t* g = gen + l * N * 2;
        int32_t cnt[128];
        int maxa = -1;
        int64_t nind = 0;
        for (int64_t i = 0; i < N; ++i) {
            int a = g[2 * i], b = g[2 * i + 1];
            if (a >= 0) {
                while (maxa < a) cnt[++maxa] = 0;
                cnt[a]++;
            }
            if (b >= 0) {
                while (maxa < b) cnt[++maxa] = 0;
                cnt[b]++;
            }
            if (a >= 0 && b >= 0) ++nind;  // is_missing = any allele < 0
        }
        int allelism = 0;
        for (int a = 0; a <= maxa; ++a) allelism += cnt[a] > 0;
        int64_t dc_all = maxa >= 1 ? cnt[1] : 0;
        bool k = allelism == 2 && dc_all > 2;
        keep[l] = k;
        af[l] = k ? (double)dc_all / (double)(nind * 2) : 0.0;
        kept += k;
    }
    *n_keep = kept;
    return 0;
}

// Derived-allele codes for kept sites, transposed to sample-major and packed 2 bits per call
// (SNP j of a row at bits 2*(j%4) of byte j/4; rows padded with code 0 to a multiple of 32 SNPs).
// Missing calls are imputed with binomial(2, af) in the reference's draw order: samples outer,
// kept sites ascending inner (popvae.py:233-235).  Codes: 0,1,2 = derived-allele count.
int pvae_prep_impute_pack(const int8_t* gen, int64_t L0, int64_t N, const uint8_t* keep, const double* af,
                          pvae_rng* rng, uint8_t* packed, int64_t row_bytes, int64_t* n_imputed) {
    std::vector<int64_t> sites;
    for (int64_t l = 0; l < L0; ++l)
        if (keep[l]) sites.push_back(l);
    const int64_t L = (int64_t)sites.size();
    if (row_bytes * 4 < (L + 31) / 32 * 32) return fail(-2, "row_bytes too small");
    memset(packed, 0, (size_t)(N * row_bytes));
    // pass 1 (site-major, blocked by 4 sites = one output byte): code 3 marks a missing call
    for (int64_t j0 = 0; j0 < L; j0 += 4) {
        int nb = (int)std::min<int64_t>(4, L - j0);
        for (int64_t i = 0; i < N; ++i) {
            uint8_t byte = 0;
            for (int t = 0; t < nb; ++t) {
                const int8_t* g = gen + (sites[j0 + t] * N + i) * 2;
                uint8_t code = (g[0] < 0 || g[1] < 0) ? 3 : (uint8_t)((g[0] == 1) + (g[1] == 1));
                byte |= code << (2 * t);
            }
            packed[i * row_bytes + j0 / 4] = byte;
        }
    }
    // pass 2 (sample-major): draw in the reference's order
    int64_t drawn = 0;
    std::vector<double> afk((size_t)L);
    for (int64_t j = 0; j < L; ++j) afk[j] = af[sites[j]];
    // np.random.binomial validates the whole p array before drawing (ValueError in the reference)
    for (int64_t j = 0; j < L; ++j)
        if (!(afk[j] >= 0.0 && afk[j] <= 1.0)) return fail(-4, "binomial: p < 0, p > 1 or p contains NaNs");
    for (int64_t i = 0; i < N; ++i) {
        uint8_t* row = packed + i * row_bytes;
        for (int64_t jb = 0; jb < (L + 3) / 4; ++jb) {
            uint8_t b = row[jb];
            if (!(b & (b >> 1) & 0x55)) continue;  // no 3-code in this byte
            for (int t = 0; t < 4; ++t) {
                if (((b >> (2 * t)) & 3) == 3) {
                    int64_t x = rng->binomial_small(2, afk[jb * 4 + t]);
                    b = (uint8_t)((b & ~(3 << (2 * t))) | (x << (2 * t)));
                    ++drawn;
                }
            }
            row[jb] = b;
        }
    }
    if (n_imputed) *n_imputed = drawn;
    return 0;
}

// Column gather on a packed matrix (used by --max_SNPs, popvae.py:250-252)
int pvae_prep_gather_columns(const uint8_t* src, int64_t N, int64_t src_row_bytes, const int64_t* cols,
                             int64_t n_cols, uint8_t* dst, int64_t dst_row_bytes) {
    if (dst_row_bytes * 4 < (n_cols + 31) / 32 * 32) return fail(-2, "dst_row_bytes too small");
    memset(dst, 0, (size_t)(N * dst_row_bytes));
    for (int64_t i = 0; i < N; ++i) {
        const uint8_t* s = src + i * src_row_bytes;
        uint8_t* d = dst + i * dst_row_bytes;
        for (int64_t j = 0; j < n_cols; ++j) {
            int64_t c = cols[j];
            uint8_t code = (s[c >> 2] >> (2 * (c & 3))) & 3;
            d[j >> 2] |= code << (2 * (j & 3));
        }
    }
    return 0;
}

}  // extern "C"
