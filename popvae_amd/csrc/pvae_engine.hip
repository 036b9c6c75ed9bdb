// pvae_engine.hip -- engine object + C ABI (include/pvae.h) around the gfx950 kernels in pvae_kernels.h.
//
// What lives where in HBM (per rank; L = local SNP count padded to 32, Wp = padded width):
//   geno     N x L/4 bytes          resident 2-bit genotype matrix (all samples, this rank's SNP columns)
//   W1,m,v   3 x L*Wp floats        wide encoder kernel + Adam slots, fragment-native tiles
//   V,m,v    3 x L*Wp floats        wide decoder kernel + Adam slots, fragment-native tiles
//   cb,m,v   3 x L floats           decoder output bias
//   narrow   (2D-1) dense layers    row-major [in_p][out_p], padded to multiples of 32, + Adam slots
//   best     1 x params             ModelCheckpoint(save_best_only) snapshot (allocated on first use)
//   scratch  split-K partials of the two wide kernels, activations, Xt (transposed batch codes)
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cmath>
#include <cstdarg>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <map>
#include <mutex>
#include <optional>
#include <string>
#include <vector>

#include <dlfcn.h>

#include "../../include/pvae.h"
#include "pvae_kernels.h"
#include "pvae_narrow.h"
#include "pvae_pca.h"

#define PV_VERSION "popvae_amd 0.3.0 (gfx950)"
// passes over several row tiles: first-layer products on the bf16 matrix pipe at f32 accuracy (pvk_enc_fwd<.., BF = 1>);
// -DPV_MT_BF16=0 builds the f32-pipe kernels instead (A/B)
#ifndef PV_MT_BF16
#define PV_MT_BF16 1
#endif

// pvae_train_step_noalloc: the split-K / split-N partial sums of the wide kernels (encpart, dgpart, losspart) live in this
// file-scope device array instead of hipMalloc'ed buffers.  32 MiB per GPU the library runs on, held until the process
// ends; every call on a GPU shares that GPU's copy (calls that overlap on one GPU race).  In host code the array's name
// is only a handle for hipGetSymbolAddress: kernels get the per-device address kept in g_scratch_addr.
#define PV_SCRATCH_BYTES (32u << 20)
static __device__ __align__(256) float pv_scratch[PV_SCRATCH_BYTES / sizeof(float)];

namespace {

struct Buf3 {  // weight + Adam slots (+ lazily allocated gradient dump / best snapshot)
    float *w = nullptr, *m = nullptr, *v = nullptr, *g = nullptr, *best = nullptr;
    // wide tensors only: the "other" copy.  A train step streams w/m/v -> w2/m2/v2 and swaps: in-place
    // read-modify-write tops out ~15% below what separate read and write streams reach on HBM3E.
    float *w2 = nullptr, *m2 = nullptr, *v2 = nullptr;
    // wide tensors: w | m | v live tile-interleaved in ONE allocation ([W 16K | m 16K | v 16K] per subtile), so that
    // a step reads one contiguous window and writes one; m and v are then views (w + ts, w + 2 ts), ts = tile floats
    int64_t il_tile = 0;  // 0 = plain separate arrays
    int64_t n = 0;
    void swap() { std::swap(w, w2); std::swap(m, m2); std::swap(v, v2); }
};

struct Dense {  // narrow layer, padded row-major [kin][nout]
    int kin = 0, nout = 0, act = PV_ACT_LINEAR;
    Buf3 W, B;
    float *wf = nullptr, *wb = nullptr;  // fused narrow stack: operand-order copies of W.w (pvae_narrow.h)
};

enum TKind { T_W1, T_B1, T_NARROW_W, T_NARROW_B, T_HEAD_W, T_HEAD_B, T_VOUT, T_COUT };
struct TensorRef {
    TKind kind;
    int layer;      // index into narrow[] (or 0/1 = mu/s for heads)
    int64_t rows, cols;  // Keras shape (cols = 0 for bias)
};

enum Cat { C_GATHER, C_ENC_FWD, C_NARROW_FWD, C_DEC, C_NARROW_BWD, C_ENC_BWD, C_XCH0, C_XCH1, C_DEC_UPD, C_REDUCE, C_NCAT };
const char* kCatNames[C_NCAT] = {"gather", "enc_fwd", "narrow_fwd", "dec", "narrow_bwd", "enc_bwd", "xch0", "xch1", "dec_upd", "reduce"};

}  // namespace

struct pvae_engine {
    pvae_config cfg{};
    std::string err;
    int dev = 0;
    hipStream_t own_stream = nullptr, stream = nullptr;
    // The decoder pass in two launches (SNP shards and other short SNP axes; pvk_dec2<.., PART>): the loss + dg launch stays on
    // `stream` and feeds the backward chain, the dV + Adam launch goes to `side`, a stream whose CU mask leaves a few CUs of
    // every XCD free for the chain (its workgroups need 135 KB of LDS: they fit on no CU that holds a decoder workgroup).
    hipStream_t side = nullptr;
    hipEvent_t ev_da = nullptr, ev_upd = nullptr;
    bool split_dec = false;    // configured at pvae_create
    bool split_now = false;    // the epoch loop is in its plain one-tile training steps
    bool upd_pending = false;  // a dV + Adam launch is in flight on `side`: `stream` has not waited for it yet
    // dims
    int64_t L = 0, Lp = 0;
    int nblk = 0, W = 0, Wp = 0, NWB = 0, ngrp = 0, D = 0, K = 0, Kh = 0, Kz = 0;
    // genotypes
    uint8_t* geno = nullptr;
    int64_t n_samples = 0, row_bytes = 0;
    // parameters
    Buf3 W1, Vout, cout, b1;
    std::vector<Dense> enc;  // enc[0..D-2] = hidden layers 2..D
    Dense heads;             // [Wp][Kh]: cols [0,K) mu, [K,2K) log-var
    std::vector<Dense> dec;  // dec[0] = K->W linear, dec[1..D-1] elu
    std::vector<TensorRef> tensors;
    // activations ([32][ld] row-major)
    float* P = nullptr;                 // exchange buffer 0: first-layer pre-activation sums [32][Wp]
    std::vector<float*> h;              // h[0..D-1]
    float *headout = nullptr, *z = nullptr, *kl = nullptr, *eps = nullptr;
    std::vector<float*> g;              // g[0..D-1]
    float* xch1 = nullptr;              // exchange buffer 1: [32*Wp] dg_D + [32] recon
    float *dA = nullptr, *dB = nullptr, *dpre1 = nullptr, *dheads = nullptr, *dz = nullptr;
    uint64_t* Xt = nullptr;
    uint64_t* Xr = nullptr;    // the gathered (untransposed) words of the batch: operand order of pvk_enc_fwd
    uint64_t* Xrn = nullptr;   // the same for the NEXT batch of the epoch loop (one tile): operand of the fused forward in pvk_enc_bwd
    uint64_t* Xt2 = nullptr;   // second transposed-batch buffer (one tile): the epoch loop's step head gathers the NEXT batch
    uint64_t* xt_act = nullptr;  // the transposed batch the decoder / encoder-backward launches read (Xt, or Xt2 in the epoch loop)
    float* probs = nullptr;  // decoder.predict staging [32][L], allocated on first pvae_decode
    float* enc_out = nullptr;  // encoder.predict staging: per 32-row batch [32][Kh] heads + [32][Kz] z, read back once
    int64_t enc_out_cap = 0;   // in floats
    // scratch
    float *encpart = nullptr, *dgpart = nullptr, *losspart = nullptr;
    int64_t encpart_cap = 0, dgpart_cap = 0, losspart_cap = 0;  // floats allocated: checked against the grids of the multi-tile passes
    float* scratch = nullptr;      // this device's copy of pv_scratch (hipGetSymbolAddress at pvae_create)
    float* h_out3 = nullptr;       // pinned staging of {loss, recon, kl}
    int enc_wgs = 0, dec_wgs = 0, encb_wgs = 0, dec_eval_wgs = 0;
    int dec2_wgs = 0;          // plain training pass at widths up to 128: pvk_dec2 (0 = use pvk_dec)
    // Batches of 33 .. 128 rows in ONE weight pass (width 128, one rank): T = 2 or 4 row tiles of 32 go through pvk_enc_fwd /
    // pvk_enc_bwd <.., T>, pvk_dec3<T> and 2 T chain workers; activation buffers hold 32 T rows.  T == 1: off.
    int T = 1;
    int dec3_wgs = 0;          // grid of pvk_dec3<T> (one workgroup per CU, 4 / T subtiles per iteration)
    bool dense16 = true;       // widths 256 / 512: pvk_dense_fwd16 for the hidden layers (PVAE_NO_DENSE16=1: pvk_dense_fwd)
    bool dec3b = true;         // T = 4: pvk_dec3b (logits and dV on the bf16 matrix pipe) instead of pvk_dec3<4>
    int dece_mt_wgs = 0;       // grid of pvk_dec_eval_mt<T> (two workgroups per CU: the T g tiles are 16.5 T KB of LDS)
    int encf_mt_wgs = 0;       // grid of pvk_enc_fwd<4, T> (T > 1: no fused forward to match, two workgroups on every CU) = its partial tiles
    int encb_parts = 0;        // partial tiles [32][Wp] the fused forward of pvk_enc_bwd leaves in encpart (one per subtile slot)
    int64_t encpart_floats = 0;
    int64_t tile_floats = 0;
    int ts4 = 0;
    // fused narrow stack (width <= 128): op tables in device memory
    bool fused_narrow = false, narrow_warm = true, fuse_fwd = false;
    // epoch loop: the step head gathers the NEXT batch (transposed copy + compact words) and the forward fused into the wide
    // encoder backward reads the compact words.  Takes the kernel's HBM traffic from 1.03x to 1.011x of the algorithmic bytes
    // (profiles/r03_exp_xrn_pmc_traffic_C4.json) but not its time (546-547 us against 548), and the head writes 8 MB more per
    // step at 1M SNPs (8.2 -> 11.7 us): a net loss of ~2 us per step, so it is opt-in (PVAE_XRN=1).
    bool use_xrn = false;
    // Opt-in (PVAE_GATHER_AHEAD=1): the next batch is gathered by a launch of its own (pvk_gather_next) on `gstream`, beside the
    // narrow forward chain of the current step, where HBM idles; the head keeps its partial sums, scalars and noise and the fused
    // forward reads the compact words.  Bit-identical, and measured SLOWER (C2 0.184 against 0.175 ms/step, C4 1.171 against
    // 1.158): each of the two cross-stream event waits of a step opens a ~6 us bubble on the waiting stream (rocprofv3 timeline,
    // tools/gpu_ga.sh) -- more than the ~5 us the head saves.  The same bubbles are part of what the two-launch decoder pass loses.
    bool gather_ahead = false;
    hipStream_t gstream = nullptr;
    hipEvent_t ev_pre = nullptr, ev_gather = nullptr;
    bool gather_pending = false;  // `stream` has not waited for the gather in flight on `gstream` yet
    bool narrow_pack_dirty = true;  // the operand-order copies lag behind the Keras-layout kernels
    PvFwdOp* d_fwd_ops = nullptr;
    PvBwdOp *d_bwd_ops = nullptr, *d_bwd_ops_grad = nullptr;
    std::vector<float*> dpre_bufs;  // per narrow dense layer, [32][nout]
    int n_fwd_ops = 0, n_enc_ops = 0, n_bwd_ops = 0;
    // step state
    PvStep* st = nullptr;
    int* d_rows = nullptr;
    int64_t rows_cap = 0;
    int* d_ntotal = nullptr;
    uint64_t* d_counter = nullptr;
    float* d_loss = nullptr;  // [loss_cap][3]
    int64_t loss_cap = 0;
    float lr_host = 1e-3f;
    int bce_probs = 0;  // pvae_set_bce_mode
    bool has_best = false;
    // SNP-axis sharding with the engine's own RCCL communicator (pvae_comm_init): the two all-reduces of a step are
    // enqueued on the engine's stream between the stages, no host code and no cross-stream events in between
    void* comm = nullptr;
    // one-shot exchange among processes (pvae_p2p_*): this rank's inbox, the peers' mappings of theirs, the exchange counter
    unsigned long long* p2p_inbox = nullptr;
    void* p2p_peer[PV_P2P_MAX_WORLD] = {};
    int64_t p2p_words = 0;
    unsigned p2p_seq = 0;
    int p2p_parity = 0;        // inbox half of the next exchange: flips with every exchange, whatever the sequence number does
    int p2p_max_spin = 1 << 26;  // polls of a word before an exchange gives up (PVAE_P2P_SPIN_LOG2; ~a minute at the default)
    int* p2p_status = nullptr;
    bool p2p = false;
    // staged API: rows whose first-layer partial sums the last pvae_stage_c_next left in encpart (empty = none)
    std::vector<int32_t> promised_rows;
    // pinned host staging ring for the per-step API (rows + eps), so that async copies never read caller memory
    struct Slot { char* host = nullptr; hipEvent_t ev = nullptr; bool pending = false; };
    Slot slots[8];
    int slot_next = 0;
    PvAdam adam{};
    // timing
    bool timing = false;
    unsigned timing_mask = ~0u;  // categories that get events (an event pair costs ~5-10 us of stream time per launch)
    struct Ev { hipEvent_t a, b; int cat; };
    std::vector<Ev> evs;
    size_t ev_used = 0;
    double cat_ms[C_NCAT] = {0};
    int64_t cat_n[C_NCAT] = {0};
};

namespace {

int set_err(pvae_engine* e, int code, const char* fmt, ...) {
    char buf[512];
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(buf, sizeof buf, fmt, ap);
    va_end(ap);
    if (e) e->err = buf;
    return code;
}

#define HIPCHK(expr)                                                                                  \
    do {                                                                                              \
        hipError_t _s = (expr);                                                                       \
        if (_s != hipSuccess)                                                                         \
            return set_err(e, -10, "%s failed: %s (%s:%d)", #expr, hipGetErrorString(_s), __FILE__, __LINE__); \
    } while (0)

int pad32(int x) { return (x + 31) / 32 * 32; }

// address of pv_scratch on device `dev` (each GPU holds its own copy at its own address): looked up once per device,
// with that device current
std::mutex g_scratch_mu;
float* g_scratch_addr[64] = {nullptr};
int scratch_address(pvae_engine* e, int dev, float** out) {
    std::lock_guard<std::mutex> lk(g_scratch_mu);
    if (dev < 0 || dev >= 64) return set_err(e, -2, "device ordinal %d beyond the scratch table", dev);
    if (!g_scratch_addr[dev]) {
        HIPCHK(hipSetDevice(dev));
        void* p = nullptr;
        HIPCHK(hipGetSymbolAddress(&p, HIP_SYMBOL(pv_scratch)));
        if (!p) return set_err(e, -10, "hipGetSymbolAddress(pv_scratch) returned a null address on device %d", dev);
        g_scratch_addr[dev] = static_cast<float*>(p);
    }
    *out = g_scratch_addr[dev];
    return 0;
}

int alloc3(pvae_engine* e, Buf3& b, int64_t n) {
    b.n = n;
    HIPCHK(hipMalloc(&b.w, n * sizeof(float)));
    HIPCHK(hipMalloc(&b.m, n * sizeof(float)));
    HIPCHK(hipMalloc(&b.v, n * sizeof(float)));
    // zero-fills are issued on the engine's stream: a null-stream hipMemset is not ordered against a
    // non-blocking stream and could land after a kernel has already written the buffer
    HIPCHK(hipMemsetAsync(b.w, 0, n * sizeof(float), e->stream));
    HIPCHK(hipMemsetAsync(b.m, 0, n * sizeof(float), e->stream));
    HIPCHK(hipMemsetAsync(b.v, 0, n * sizeof(float), e->stream));
    return 0;
}
// wide tensor: two tile-interleaved copies (ping-pong pair), tile = floats per subtile
int alloc_wide(pvae_engine* e, Buf3& b, int64_t n, int64_t tile) {
    b.n = n;
    b.il_tile = tile;
    HIPCHK(hipMalloc(&b.w, 3 * n * sizeof(float)));
    HIPCHK(hipMalloc(&b.w2, 3 * n * sizeof(float)));
    HIPCHK(hipMemsetAsync(b.w, 0, 3 * n * sizeof(float), e->stream));
    HIPCHK(hipMemsetAsync(b.w2, 0, 3 * n * sizeof(float), e->stream));
    b.m = b.w + tile; b.v = b.w + 2 * tile;
    b.m2 = b.w2 + tile; b.v2 = b.w2 + 2 * tile;
    return 0;
}
void free3(Buf3& b) {
    hipFree(b.w); hipFree(b.g); hipFree(b.best); hipFree(b.w2);
    if (!b.il_tile) { hipFree(b.m); hipFree(b.v); hipFree(b.m2); hipFree(b.v2); }
    b = Buf3();
}
// copies between a (possibly tile-interleaved) parameter array and a plain one
hipError_t copy_params(pvae_engine* e, Buf3& b, float* plain, bool to_plain) {
    if (!b.il_tile) {
        return to_plain ? hipMemcpyAsync(plain, b.w, b.n * 4, hipMemcpyDeviceToDevice, e->stream)
                        : hipMemcpyAsync(b.w, plain, b.n * 4, hipMemcpyDeviceToDevice, e->stream);
    }
    const size_t tb = (size_t)b.il_tile * 4, rows = (size_t)(b.n / b.il_tile);
    return to_plain ? hipMemcpy2DAsync(plain, tb, b.w, 3 * tb, tb, rows, hipMemcpyDeviceToDevice, e->stream)
                    : hipMemcpy2DAsync(b.w, 3 * tb, plain, tb, tb, rows, hipMemcpyDeviceToDevice, e->stream);
}
hipError_t zero_moments(pvae_engine* e, Buf3& b) {
    if (!b.il_tile) {
        hipError_t s1 = hipMemsetAsync(b.m, 0, b.n * 4, e->stream);
        return s1 != hipSuccess ? s1 : hipMemsetAsync(b.v, 0, b.n * 4, e->stream);
    }
    const size_t tb = (size_t)b.il_tile * 4;
    return hipMemset2DAsync(b.m, 3 * tb, 0, 2 * tb, (size_t)(b.n / b.il_tile), e->stream);
}
int allocf(pvae_engine* e, float** p, int64_t n) {
    HIPCHK(hipMalloc(p, n * sizeof(float)));
    HIPCHK(hipMemsetAsync(*p, 0, n * sizeof(float), e->stream));
    return 0;
}
int alloc_dense(pvae_engine* e, Dense& d, int kin, int nout, int act) {
    d.kin = kin; d.nout = nout; d.act = act;
    int rc = alloc3(e, d.W, (int64_t)kin * nout);
    if (rc) return rc;
    return alloc3(e, d.B, nout);
}

template <typename F>
void for_each_buf(pvae_engine* e, F f) {
    f(e->W1); f(e->b1);
    for (auto& d : e->enc) { f(d.W); f(d.B); }
    f(e->heads.W); f(e->heads.B);
    for (auto& d : e->dec) { f(d.W); f(d.B); }
    f(e->Vout); f(e->cout);
}

struct Timer {
    pvae_engine* e;
    bool on;
    size_t idx = 0;
    hipStream_t s;
    Timer(pvae_engine* e_, int cat, hipStream_t s_ = nullptr) : e(e_), on(e_->timing && ((e_->timing_mask >> cat) & 1u)), s(s_ ? s_ : e_->stream) {
        if (!on) return;
        if (e->ev_used == e->evs.size()) {
            pvae_engine::Ev ev;
            hipEventCreate(&ev.a);
            hipEventCreate(&ev.b);
            e->evs.push_back(ev);
        }
        idx = e->ev_used++;
        e->evs[idx].cat = cat;
        hipEventRecord(e->evs[idx].a, s);
    }
    ~Timer() {
        if (on) hipEventRecord(e->evs[idx].b, s);
    }
};

void flush_timing(pvae_engine* e) {
    if (!e->ev_used) return;
    hipStreamSynchronize(e->stream);
    if (e->side) hipStreamSynchronize(e->side);
    for (size_t i = 0; i < e->ev_used; ++i) {
        float ms = 0.f;
        if (hipEventElapsedTime(&ms, e->evs[i].a, e->evs[i].b) == hipSuccess) {
            e->cat_ms[e->evs[i].cat] += ms;
            e->cat_n[e->evs[i].cat] += 1;
        }
    }
    e->ev_used = 0;
}

// ---- dynamic-LDS sizes ----
size_t lds_dense_fwd(int K) { return (size_t)(32 * (K + 1) + 4096) * 4; }
size_t lds_dense_bwd(int N) { return (size_t)(2 * 32 * (N + 1) + 4096) * 4; }
static size_t lds_dec_eval_mt(int T) { return (size_t)T * 32 * 129 * sizeof(float); }  // the T g tiles
// evaluation launches of pvk_dec use no per-wave staging tiles and no dg reduction: the g tile alone (more blocks per CU)
size_t lds_dec_eval(int NWB) { return (size_t)(32 * (NWB * 32 + 1) + 64) * 4; }
size_t lds_dec(int NWB) { return (size_t)(32 * (NWB * 32 + 1) + 4 * 3 * 32 * 33) * 4; }
size_t lds_enc_bwd(int Wp) { return (size_t)32 * Wp * 4; }
constexpr int dec_grp_waves(int ngrp) { return ngrp == 2 ? 4 : 8; }  // waves per workgroup of pvk_dec_grp
size_t lds_dec_grp(int ngrp) {
    const size_t nw = dec_grp_waves(ngrp);
    return std::max<size_t>((size_t)32 * (ngrp * 128 + 1) + nw * 2 * 32 * 33, nw * 4 * 1024 + nw) * 4;
}

template <typename Kern>
hipError_t allow_lds(Kern k, size_t bytes) {
    return hipFuncSetAttribute(reinterpret_cast<const void*>(k), hipFuncAttributeMaxDynamicSharedMemorySize, (int)bytes);
}

int setup_kernel_attrs(pvae_engine* e) {
    int maxK = std::max(e->Wp, std::max(e->Kh, e->Kz));
    HIPCHK(allow_lds(pvk_dense_fwd, lds_dense_fwd(maxK)));
    HIPCHK(allow_lds(pvk_dense_fwd16<512>, (size_t)(32 * (512 + 4) + 4096) * 4));
    HIPCHK(allow_lds(pvk_dense_fwd16<256>, (size_t)(32 * (256 + 4) + 4096) * 4));
    HIPCHK(allow_lds(pvk_dense_bwd16<512>, (size_t)(32 * (512 + 4) + 4096) * 4));
    HIPCHK(allow_lds(pvk_dense_bwd16<256>, (size_t)(32 * (256 + 4) + 4096) * 4));
    HIPCHK(allow_lds(pvk_dense_bwd, lds_dense_bwd(maxK)));
    HIPCHK(allow_lds(pvk_narrow_fwd, pvn_lds_fwd()));
    HIPCHK(allow_lds(pvk_narrow_bwd<PV_MODE_TRAIN>, pvn_lds_bwd()));
    HIPCHK(allow_lds(pvk_narrow_bwd<PV_MODE_GRAD>, pvn_lds_bwd()));
    HIPCHK(allow_lds(pvk_narrow_bwd<PV_MODE_TRAIN_ACC>, pvn_lds_bwd()));
#define PV_ATTR(N)                                                                    \
    HIPCHK(allow_lds(pvk_dec2<N>, PvDec2<N>::lds()));                                 \
    HIPCHK(allow_lds(pvk_dec2<N, 1>, PvDec2<N>::lds()));                              \
    HIPCHK(allow_lds(pvk_dec2<N, 2>, PvDec2<N>::lds()));                              \
    HIPCHK(allow_lds(pvk_dec<N, 1, PV_MODE_TRAIN>, lds_dec(N)));                      \
    HIPCHK(allow_lds(pvk_dec<N, 1, PV_MODE_GRAD>, lds_dec(N)));                       \
    HIPCHK(allow_lds(pvk_dec<N, 1, PV_MODE_TRAIN_ACC>, lds_dec(N)));                  \
    HIPCHK(allow_lds(pvk_dec<N, 0, PV_MODE_TRAIN>, lds_dec(N)));                      \
    HIPCHK(allow_lds((pvk_enc_bwd<N, PV_MODE_TRAIN, 0>), lds_enc_bwd(e->Wp)));        \
    HIPCHK(allow_lds((pvk_enc_bwd<N, PV_MODE_TRAIN, 1>), lds_enc_bwd(e->Wp)));        \
    HIPCHK(allow_lds((pvk_enc_bwd<N, PV_MODE_GRAD, 0>), lds_enc_bwd(e->Wp)));         \
    HIPCHK(allow_lds((pvk_enc_bwd<N, PV_MODE_TRAIN_ACC, 0>), lds_enc_bwd(e->Wp)));
#define PV_ATTR_MT(TT)                                                                                     \
    HIPCHK(allow_lds((pvk_dec3b<PV_MODE_TRAIN>), PvDec3b::lds()));                                          \
    HIPCHK(allow_lds((pvk_dec3b<PV_MODE_GRAD>), PvDec3b::lds()));                                           \
    HIPCHK(allow_lds((pvk_dec3b<PV_MODE_TRAIN_ACC>), PvDec3b::lds()));                                      \
    HIPCHK(allow_lds((pvk_dec3<TT, PV_MODE_TRAIN>), PvDec3<TT>::lds()));                                    \
    HIPCHK(allow_lds((pvk_dec3<TT, PV_MODE_GRAD>), PvDec3<TT>::lds()));                                     \
    HIPCHK(allow_lds((pvk_dec3<TT, PV_MODE_TRAIN_ACC>), PvDec3<TT>::lds()));                                \
    HIPCHK(allow_lds((pvk_enc_bwd<4, PV_MODE_TRAIN, 0, TT>), TT * lds_enc_bwd(e->Wp)));                     \
    HIPCHK(allow_lds((pvk_enc_bwd<4, PV_MODE_GRAD, 0, TT>), TT * lds_enc_bwd(e->Wp)));                      \
    HIPCHK(allow_lds((pvk_enc_bwd<4, PV_MODE_TRAIN_ACC, 0, TT>), TT * lds_enc_bwd(e->Wp)));                 \
    HIPCHK(allow_lds((pvk_dec_eval_mt<TT, 0>), lds_dec_eval_mt(TT)));                                      \
    HIPCHK(allow_lds((pvk_dec_eval_mt<TT, 1>), lds_dec_eval_mt(TT)));                                      \
    HIPCHK(allow_lds((pvk_dec_eval_bf<TT, 0>), PvDecEvalBf<TT>::lds()));                                   \
    HIPCHK(allow_lds((pvk_dec_eval_bf<TT, 1>), PvDecEvalBf<TT>::lds()));
    if (e->T == 2) {
        PV_ATTR_MT(2)
    } else if (e->T == 4) {
        PV_ATTR_MT(4)
    }
#undef PV_ATTR_MT
    switch (e->NWB) {
        case 1: PV_ATTR(1) break;
        case 2: PV_ATTR(2) break;
        case 3: PV_ATTR(3) break;
        default: PV_ATTR(4) break;
    }
#undef PV_ATTR
#define PV_ATTR_G(G)                                                                  \
    HIPCHK(allow_lds((pvk_dec_grp<G, 1, PV_MODE_TRAIN, dec_grp_waves(G)>), lds_dec_grp(G)));     \
    HIPCHK(allow_lds((pvk_dec_grp<G, 1, PV_MODE_GRAD, dec_grp_waves(G)>), lds_dec_grp(G)));      \
    HIPCHK(allow_lds((pvk_dec_grp<G, 1, PV_MODE_TRAIN_ACC, dec_grp_waves(G)>), lds_dec_grp(G))); \
    HIPCHK(allow_lds((pvk_dec_grp<G, 0, PV_MODE_TRAIN, dec_grp_waves(G)>), lds_dec_grp(G)));
    if (e->ngrp == 2) { PV_ATTR_G(2) }
    if (e->ngrp == 4) { PV_ATTR_G(4) }
#undef PV_ATTR_G
    return 0;
}

#define PV_DISPATCH_NWB(e, CALL)        \
    switch ((e)->NWB) {                 \
        case 1: { constexpr int N_ = 1; CALL; } break; \
        case 2: { constexpr int N_ = 2; CALL; } break; \
        case 3: { constexpr int N_ = 3; CALL; } break; \
        default: { constexpr int N_ = 4; CALL; } break; \
    }

int launch_check(pvae_engine* e, const char* what) {
    hipError_t s = hipGetLastError();
    if (s != hipSuccess) return set_err(e, -11, "launch of %s failed: %s", what, hipGetErrorString(s));
    return 0;
}

// ------------------------------------------------------------------------------------------------
// RCCL, bound at run time (dlopen: the process may already hold a librccl -- PyTorch ships one -- and must not get a
// second one; nothing is linked).  Signatures as in rccl.h (ROCm 7.2): ncclUniqueId is 128 opaque bytes passed BY VALUE,
// ncclFloat32 = 7, ncclSum = 0.
// ------------------------------------------------------------------------------------------------
struct PvNcclId { char internal[128]; };
struct PvRccl {
    void* lib = nullptr;
    int (*GetUniqueId)(PvNcclId*) = nullptr;
    int (*CommInitRank)(void**, int, PvNcclId, int) = nullptr;
    int (*CommDestroy)(void*) = nullptr;
    int (*CommCount)(void*, int*) = nullptr;
    int (*CommUserRank)(void*, int*) = nullptr;
    int (*AllReduce)(const void*, void*, size_t, int, int, void*, hipStream_t) = nullptr;
    const char* (*GetErrorString)(int) = nullptr;
    std::string err;
};
PvRccl* rccl() {
    static PvRccl r;
    static bool tried = false;
    if (tried) return &r;
    tried = true;
    const char* names[] = {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"};
    for (const char* n : names)  // one that the process already holds (PyTorch's) first
        if ((r.lib = dlopen(n, RTLD_NOW | RTLD_NOLOAD))) break;
    if (!r.lib)
        for (const char* n : names)
            if ((r.lib = dlopen(n, RTLD_NOW | RTLD_GLOBAL))) break;
    if (!r.lib) { r.err = std::string("librccl not loadable: ") + dlerror(); return &r; }
    r.GetUniqueId = (decltype(r.GetUniqueId))dlsym(r.lib, "ncclGetUniqueId");
    r.CommInitRank = (decltype(r.CommInitRank))dlsym(r.lib, "ncclCommInitRank");
    r.CommDestroy = (decltype(r.CommDestroy))dlsym(r.lib, "ncclCommDestroy");
    r.CommCount = (decltype(r.CommCount))dlsym(r.lib, "ncclCommCount");
    r.CommUserRank = (decltype(r.CommUserRank))dlsym(r.lib, "ncclCommUserRank");
    r.AllReduce = (decltype(r.AllReduce))dlsym(r.lib, "ncclAllReduce");
    r.GetErrorString = (decltype(r.GetErrorString))dlsym(r.lib, "ncclGetErrorString");
    if (!r.GetUniqueId || !r.CommInitRank || !r.CommDestroy || !r.AllReduce || !r.GetErrorString || !r.CommCount ||
        !r.CommUserRank) {
        r.err = "librccl lacks one of ncclGetUniqueId / ncclCommInitRank / ncclCommDestroy / ncclAllReduce / ncclGetErrorString / "
                "ncclCommCount / ncclCommUserRank";
        r.lib = nullptr;
    }
    return &r;
}
// SUM all-reduce of exchange buffer `which` over the shards, in place, on the engine's stream; nothing to do on one rank
// rows: rows of the device pass whose partial sums are exchanged (32; 32 T in a multi-tile pass)
static inline bool has_comm(const pvae_engine* e) { return e->comm || e->p2p; }
int shard_allreduce(pvae_engine* e, int which, int rows = 32) {
    if (e->cfg.world == 1 && !has_comm(e)) return 0;  // a one-rank communicator (rehearsal on one GPU) is still driven
    if (!has_comm(e)) return set_err(e, -5, "world > 1 without a communicator: call pvae_comm_init, or drive the step through "
                                            "pvae_stage_a/b/c with all-reduces in between");
    float* buf = which == 0 ? e->P : e->xch1;
    const size_t n = which == 0 ? (size_t)rows * e->Wp : (size_t)rows * e->Wp + 32;  // dg rows, then the loss sums
    Timer t(e, which == 0 ? C_XCH0 : C_XCH1);
    if (e->p2p) {
        if ((int64_t)n > e->p2p_words) return set_err(e, -12, "exchange of %zu floats, inbox slots hold %lld", n, (long long)e->p2p_words);
        PvPeers peers;
        for (int r = 0; r < e->cfg.world; ++r)
            peers.inbox[r] = r == e->cfg.rank ? e->p2p_inbox : reinterpret_cast<unsigned long long*>(e->p2p_peer[r]);
        ++e->p2p_seq;
        if (e->p2p_seq == 0) ++e->p2p_seq;  // 0 is what an untouched inbox holds
        const int parity = e->p2p_parity;
        e->p2p_parity ^= 1;
        hipLaunchKernelGGL(pvk_p2p_allreduce, dim3(std::min<size_t>(16, (n + 1023) / 1024)), dim3(256), 0, e->stream, buf, (int)n, peers,
                           e->cfg.rank, e->cfg.world, e->p2p_seq, parity, e->p2p_words, e->p2p_status, e->p2p_max_spin);
        return launch_check(e, "p2p_allreduce");
    }
    const int rc = rccl()->AllReduce(buf, buf, n, 7 /* ncclFloat32 */, 0 /* ncclSum */, e->comm, e->stream);
    if (rc) return set_err(e, -12, "ncclAllReduce: %s", rccl()->GetErrorString(rc));
    return 0;
}

// ------------------------------------------------------------------------------------------------
// forward / backward building blocks (all asynchronous on e->stream)
// ------------------------------------------------------------------------------------------------
int run_dense_fwd(pvae_engine* e, const Dense& d, const float* in, float* out, const float* in_bias = nullptr,
                  float* in_act = nullptr, PvRepFwd rp = PvRepFwd{}) {
    if (rp.z && d.nout != 32) return set_err(e, -6, "sampling tail on a layer of %d outputs", d.nout);
    if (!rp.z && e->dense16 && (d.kin == 256 || d.kin == 512) && d.nout % 16 == 0) {  // hidden layers at widths 256 / 512
        const size_t lds = (size_t)(32 * (d.kin + 4) + 4096) * 4;
        if (d.kin == 512)
            hipLaunchKernelGGL(pvk_dense_fwd16<512>, dim3(d.nout / 16), dim3(512), lds, e->stream, in, (const float*)d.W.w,
                               (const float*)d.B.w, d.nout, d.act, out, in_bias, in_act);
        else
            hipLaunchKernelGGL(pvk_dense_fwd16<256>, dim3(d.nout / 16), dim3(512), lds, e->stream, in, (const float*)d.W.w,
                               (const float*)d.B.w, d.nout, d.act, out, in_bias, in_act);
        return launch_check(e, "dense_fwd16");
    }
    hipLaunchKernelGGL(pvk_dense_fwd, dim3(d.nout / 32), dim3(256), lds_dense_fwd(d.kin), e->stream, in, d.kin, d.W.w,
                       d.B.w, d.nout, d.act, out, in_bias, in_act, rp);
    return launch_check(e, "dense_fwd");
}

// hidden layers at widths 256 / 512 take pvk_dense_bwd16 (16 input units per workgroup)
bool dense16_bwd_ok(const pvae_engine* e, const Dense& d) {
    return e->dense16 && (d.nout == 256 || d.nout == 512) && d.kin >= 16 && d.kin % 16 == 0;
}
// in_is_dpre / Ynext (pvk_dense_bwd16 only): dY already is this layer's dpre / the epilogue forms the next layer's dpre in dIn
int run_dense_bwd(pvae_engine* e, Dense& d, const float* Y, const float* dY, float* dIn, float* dpre_out,
                  PvRepBwd rp = PvRepBwd{}, bool in_is_dpre = false, const float* Ynext = nullptr, int act_next = PV_ACT_LINEAR) {
    if (rp.dheads && d.kin != 32) return set_err(e, -6, "sampling-gradient tail on a layer of %d inputs", d.kin);
    if (!rp.dheads && dense16_bwd_ok(e, d)) {
        const size_t lds = (size_t)(32 * (d.nout + 4) + 4096) * 4;
        if (d.nout == 512)
            hipLaunchKernelGGL(pvk_dense_bwd16<512>, dim3(d.kin / 16), dim3(512), lds, e->stream, d.kin, (const float*)d.W.w, d.act, Y, dY,
                               in_is_dpre ? 1 : 0, dIn, dpre_out, (const PvStep*)e->st, Ynext, act_next);
        else
            hipLaunchKernelGGL(pvk_dense_bwd16<256>, dim3(d.kin / 16), dim3(512), lds, e->stream, d.kin, (const float*)d.W.w, d.act, Y, dY,
                               in_is_dpre ? 1 : 0, dIn, dpre_out, (const PvStep*)e->st, Ynext, act_next);
        return launch_check(e, "dense_bwd16");
    }
    if (in_is_dpre || Ynext) return set_err(e, -6, "chained dpre on a layer the 32-unit kernel handles");
    hipLaunchKernelGGL(pvk_dense_bwd, dim3(d.kin / 32), dim3(256), lds_dense_bwd(d.nout), e->stream, d.kin, d.W.w, d.nout,
                       d.act, Y, dY, dIn, dpre_out, e->st, rp);
    return launch_check(e, "dense_bwd");
}

// pvk_enc_fwd is launched with this much (unused) dynamic LDS so that a CU takes two of its workgroups at most.  Its grid is
// at most two workgroups per CU, but on its registers alone a CU would take four, and the dispatcher then packs some CUs with
// three or four while others get none: the kernel is bound by the matrix pipe of the fullest SIMD (150 cycles per matrix
// instruction per wave with two waves on a SIMD), and took 120 us at 1M SNPs with no loads at all against 98 us with no
// matrix instructions (tools/trace_dec3.py, -DPV_EXP_NOLOAD / -DPV_EXP_NOMMA builds).
static constexpr size_t kEncFwdLds = 60 * 1024;

// stage A: batch gather/transposes + partial wide-encoder sums -> P (pre-bias, this rank's SNPs only)
int stage_a(pvae_engine* e, int row_offset, bool need_xt, bool fwd_done = false, bool reduced = false) {
    int rc;
    if (!e->geno) return set_err(e, -3, "no genotypes loaded (pvae_load_genotypes)");
    if (need_xt) {
        Timer t(e, C_GATHER);
        e->xt_act = e->Xt;
        int pairs = (e->nblk + 1) / 2;
        hipLaunchKernelGGL(pvk_gather_transpose, dim3((pairs + 3) / 4), dim3(256), 0, e->stream, e->geno, e->row_bytes,
                           e->d_rows, e->st, row_offset, e->nblk, e->Xt, fwd_done ? (uint64_t*)nullptr : e->Xr);
        if ((rc = launch_check(e, "gather_transpose"))) return rc;
    }
    if (fwd_done && reduced) {
        // ... and the step-head launch (pvk_step_begin_gather) has summed them into P
    } else if (fwd_done) {
        // the previous step's pvk_enc_bwd<.., FUSE = 1> already left this batch's partial sums in encpart
        hipLaunchKernelGGL(pvk_reduce_parts, dim3(PV_REDUCE_GRID(32 * e->Wp)), dim3(256), 0, e->stream, e->encpart, e->encb_parts,
                           32 * e->Wp, 0.5f, e->P, (const float*)nullptr, (float*)nullptr);
        if ((rc = launch_check(e, "enc_reduce"))) return rc;
    } else {
        Timer t(e, C_ENC_FWD);
        PV_DISPATCH_NWB(e, hipLaunchKernelGGL(pvk_enc_fwd<N_>, dim3(e->enc_wgs), dim3(256), kEncFwdLds, e->stream, e->W1.w, e->geno,
                                              e->row_bytes, e->d_rows, e->st, row_offset, e->nblk, e->ngrp, e->encpart, e->ts4,
                                              need_xt ? (const uint64_t*)e->Xr : (const uint64_t*)nullptr));
        if ((rc = launch_check(e, "enc_fwd"))) return rc;
        hipLaunchKernelGGL(pvk_reduce_parts, dim3(PV_REDUCE_GRID(32 * e->Wp)), dim3(256), 0, e->stream, e->encpart, e->encb_parts, 32 * e->Wp,
                           0.5f, e->P, (const float*)nullptr, (float*)nullptr);
        if ((rc = launch_check(e, "enc_reduce"))) return rc;
    }
    return 0;
}

// op tables of the fused narrow kernels (pvae_narrow.h); grad = true builds the gradient-dump variant
int build_bwd_ops(pvae_engine* e, bool grad, PvBwdOp** out) {
    std::vector<PvBwdOp> ops;
    size_t nd = 0;
    bool alloc_failed = false;
    auto dense = [&](Dense& d, const float* in, const float* Y) {
        PvBwdOp o{};
        if (nd == e->dpre_bufs.size()) {
            float* p = nullptr;
            if (hipMalloc(&p, (size_t)32 * e->T * d.nout * sizeof(float)) != hipSuccess) p = nullptr;
            e->dpre_bufs.push_back(p);
        }
        o.dpre = e->dpre_bufs[nd++];
        if (e->fused_narrow && !d.wf) {
            if (hipMalloc(&d.wf, pvn_wf_floats(d.kin, d.nout) * sizeof(float)) != hipSuccess) d.wf = nullptr;
            if (hipMalloc(&d.wb, pvn_wb_floats(d.kin, d.nout) * sizeof(float)) != hipSuccess) d.wb = nullptr;
            if (!d.wf || !d.wb) alloc_failed = true;
        }
        o.Wf = d.wf; o.Wb = d.wb;
        o.W = d.W.w; o.mW = grad ? d.W.g : d.W.m; o.vW = grad ? nullptr : d.W.v;
        o.B = d.B.w; o.mB = grad ? d.B.g : d.B.m; o.vB = grad ? nullptr : d.B.v;
        o.gW = d.W.g; o.gB = d.B.g;
        o.in = in; o.Y = Y; o.kin = d.kin; o.nout = d.nout; o.act = d.act; o.kind = PVN_DENSE;
        ops.push_back(o);
    };
    const int D = e->D;
    for (int i = D - 1; i >= 1; --i) dense(e->dec[i], e->g[i - 1], e->g[i]);
    dense(e->dec[0], e->z, e->g[0]);
    PvBwdOp r{};
    r.kind = PVN_REPARAM; r.nout = e->Kh; r.kin = e->Kz;
    ops.push_back(r);
    dense(e->heads, e->h[D - 1], e->headout);
    for (int i = D - 2; i >= 0; --i) dense(e->enc[i], e->h[i], e->h[i + 1]);
    PvBwdOp f{};
    f.kind = PVN_FIRST_BIAS; f.dpre = e->dpre1; f.B = e->b1.w; f.mB = grad ? e->b1.g : e->b1.m; f.vB = grad ? nullptr : e->b1.v;
    f.gB = e->b1.g;
    f.Y = e->h[0]; f.nout = e->Wp; f.kin = e->fused_narrow ? 1 : 0 /* 1: pvk_narrow_adam updates b1 */; f.act = PV_ACT_ELU;
    ops.push_back(f);
    e->n_bwd_ops = (int)ops.size();
    for (auto p : e->dpre_bufs)
        if (!p) return set_err(e, -10, "hipMalloc failed for a dpre buffer");
    if (alloc_failed) return set_err(e, -10, "hipMalloc failed for an operand-order kernel copy");
    if (!*out) HIPCHK(hipMalloc(out, ops.size() * sizeof(PvBwdOp)));
    HIPCHK(hipMemcpy(*out, ops.data(), ops.size() * sizeof(PvBwdOp), hipMemcpyHostToDevice));
    return 0;
}

int build_fwd_ops(pvae_engine* e) {
    std::vector<PvFwdOp> ops;
    auto dense = [&](const Dense& d, float* out) {
        ops.push_back(PvFwdOp{d.wf, d.B.w, out, d.kin, d.nout, d.act, PVN_DENSE});
    };
    const int D = e->D;
    ops.push_back(PvFwdOp{nullptr, e->b1.w, e->h[0], 0, e->Wp, PV_ACT_ELU, PVN_BIAS_ACT});
    for (int i = 0; i < D - 1; ++i) dense(e->enc[i], e->h[i + 1]);
    dense(e->heads, e->headout);
    ops.push_back(PvFwdOp{nullptr, e->heads.B.w /* dummy, never used */, nullptr, e->Kh, e->Kz, PV_ACT_LINEAR, PVN_REPARAM});
    e->n_enc_ops = (int)ops.size();
    dense(e->dec[0], e->g[0]);
    for (int i = 1; i < D; ++i) dense(e->dec[i], e->g[i]);
    e->n_fwd_ops = (int)ops.size();
    HIPCHK(hipMalloc(&e->d_fwd_ops, ops.size() * sizeof(PvFwdOp)));
    HIPCHK(hipMemcpy(e->d_fwd_ops, ops.data(), ops.size() * sizeof(PvFwdOp), hipMemcpyHostToDevice));
    return 0;
}

// operand-order copies of the narrow kernels: rebuilt when something other than pvk_narrow_adam wrote the kernels
int ensure_narrow_packed(pvae_engine* e) {
    if (!e->fused_narrow || !e->narrow_pack_dirty) return 0;
    hipLaunchKernelGGL(pvk_narrow_pack, dim3(16, e->n_bwd_ops), dim3(256), 0, e->stream, (const PvBwdOp*)e->d_bwd_ops);
    e->narrow_pack_dirty = false;
    return launch_check(e, "narrow_pack");
}

int narrow_fused_fwd(pvae_engine* e, bool encoder_only, bool decoder_only = false, int tiles = 1) {
    int rc0;
    if ((rc0 = ensure_narrow_packed(e))) return rc0;
    const int nops = encoder_only ? e->n_enc_ops : e->n_fwd_ops, nworkers = PVQ_WORKERS_PER_TILE * tiles;
    hipLaunchKernelGGL(pvk_narrow_fwd, dim3(e->narrow_warm ? nworkers + 8 * nops : nworkers), dim3(PVQ_THREADS), pvn_lds_fwd(), e->stream,
                       e->d_fwd_ops, decoder_only ? e->n_enc_ops : 0, nops, e->P, e->eps, e->K, e->Kz, e->z, e->kl, nworkers);
    return launch_check(e, "narrow_fwd");
}

// narrow encoder: h1 = elu(P + b1), hidden layers, heads, reparameterisation + KL
int narrow_encoder(pvae_engine* e) {
    int rc;
    if (e->fused_narrow) return narrow_fused_fwd(e, true);
    // h1 = elu(P + b1) is formed inside the first consumer's staging loop (workgroup 0 writes it out for the backward)
    for (int i = 0; i < e->D - 1; ++i)
        if ((rc = i == 0 ? run_dense_fwd(e, e->enc[0], e->P, e->h[1], e->b1.w, e->h[0])
                         : run_dense_fwd(e, e->enc[i], e->h[i], e->h[i + 1]))) return rc;
    // the heads layer is one workgroup (Kh = 32 outputs): the sampling layer + KL run in its tail
    if (e->Kh == 32) {
        const PvRepFwd rp{e->eps, e->z, e->kl, e->K, e->Kz};
        return e->D == 1 ? run_dense_fwd(e, e->heads, e->P, e->headout, e->b1.w, e->h[0], rp)
                         : run_dense_fwd(e, e->heads, e->h[e->D - 1], e->headout, nullptr, nullptr, rp);
    }
    if ((rc = e->D == 1 ? run_dense_fwd(e, e->heads, e->P, e->headout, e->b1.w, e->h[0])
                        : run_dense_fwd(e, e->heads, e->h[e->D - 1], e->headout))) return rc;
    hipLaunchKernelGGL(pvk_reparam, dim3(1), dim3(64), 0, e->stream, e->headout, e->Kh, e->K, e->eps, e->z, e->Kz, e->kl);
    return launch_check(e, "reparam");
}

int narrow_decoder(pvae_engine* e) {
    int rc;
    if ((rc = run_dense_fwd(e, e->dec[0], e->z, e->g[0]))) return rc;
    for (int i = 1; i < e->D; ++i)
        if ((rc = run_dense_fwd(e, e->dec[i], e->g[i - 1], e->g[i]))) return rc;
    return 0;
}

// `stream` waits for the decoder update in flight on `side` (everything that reads or writes V | m | v, the output bias, the
// step scalars, the batch words or the last decoder activation comes behind it)
int join_side(pvae_engine* e) {
    if (!e->upd_pending) return 0;
    e->upd_pending = false;
    HIPCHK(hipStreamWaitEvent(e->stream, e->ev_upd, 0));
    return 0;
}
int join_gather(pvae_engine* e) {  // `stream` waits for the next-batch gather in flight on `gstream`
    if (!e->gather_pending) return 0;
    e->gather_pending = false;
    HIPCHK(hipStreamWaitEvent(e->stream, e->ev_gather, 0));
    return 0;
}
struct GatherScope {  // leaves the epoch loop with the gather stream joined, whatever the exit
    pvae_engine* e;
    explicit GatherScope(pvae_engine* e_) : e(e_) {}
    ~GatherScope() { join_gather(e); }
};
struct SplitScope {  // the epoch loop's plain training steps; leaves with the side stream joined, whatever the exit
    pvae_engine* e;
    explicit SplitScope(pvae_engine* e_, bool on) : e(e_) { e->split_now = on; }
    ~SplitScope() { e->split_now = false; join_side(e); }
};

// stage B: narrow forward + fused wide decoder -> xch1 (dg partial sums over this rank's SNPs, recon sums)
int stage_b(pvae_engine* e, int mode /*0 train,1 eval,2 grad*/) {
    int rc;
    {
        Timer t(e, C_NARROW_FWD);
        if (e->fused_narrow) {
            if ((rc = narrow_fused_fwd(e, false))) return rc;
        } else {
            if ((rc = narrow_encoder(e))) return rc;
            if ((rc = narrow_decoder(e))) return rc;
        }
    }
    {
        // the events of "dec" bracket the decoder kernel alone (bench.py's roofline object divides its algorithmic bytes by this
        // time); the fixed-order sum of its partial tiles is a category of its own
        std::optional<Timer> t;
        t.emplace(e, C_DEC);
        const float* gD = e->g[e->D - 1];
        if (mode == 2 && !e->Vout.g) {
            if ((rc = allocf(e, &e->Vout.g, e->Vout.n))) return rc;
            if ((rc = allocf(e, &e->cout.g, e->cout.n))) return rc;
        }
        // TRAIN / TRAIN_ACC stream copy -> copy and swap; GRAD dumps into the g buffers; eval only reads V and c
        float* nul = nullptr;
        const bool upd = mode == 0 || mode == 4;
        const float *aV = e->Vout.w, *amV = upd ? e->Vout.m : nul, *avV = upd ? e->Vout.v : nul;
        float *aV2 = upd ? e->Vout.w2 : nul, *amV2 = upd ? e->Vout.m2 : (mode == 2 ? e->Vout.g : nul), *avV2 = upd ? e->Vout.v2 : nul;
        float *acb = e->cout.w, *amC = upd ? e->cout.m : (mode == 2 ? e->cout.g : nul), *avC = upd ? e->cout.v : nul;
        const float *agV = mode == 4 ? e->Vout.g : nul, *agC = mode == 4 ? e->cout.g : nul;
        if (mode == 4 && (!agV || !agC)) return set_err(e, -6, "train-with-accumulated-gradients step without a gradient dump");
#define PV_DEC_LAUNCH(KERN, THREADS, LDS)                                                                                      \
    hipLaunchKernelGGL(KERN, dim3(e->dec_wgs), dim3(THREADS), LDS, e->stream, aV, amV, avV, aV2, amV2, avV2, acb, amC, avC, gD, \
                       e->xt_act, e->nblk, (int)e->L, e->st, e->adam, e->dgpart, e->losspart, e->ts4, agV, agC)
#define PV_DEC_MODES(K3, THREADS, LDS)                                                     \
    do {                                                                                   \
        if (mode == 0) PV_DEC_LAUNCH((K3(1, PV_MODE_TRAIN)), THREADS, LDS);                \
        else if (mode == 2) PV_DEC_LAUNCH((K3(1, PV_MODE_GRAD)), THREADS, LDS);            \
        else if (mode == 4) PV_DEC_LAUNCH((K3(1, PV_MODE_TRAIN_ACC)), THREADS, LDS);       \
        else PV_DEC_LAUNCH((K3(0, PV_MODE_TRAIN)), THREADS, LDS);                          \
    } while (0)
        if (e->ngrp == 2) {
#define PV_K3(TR, MD) pvk_dec_grp<2, TR, MD, dec_grp_waves(2)>
            PV_DEC_MODES(PV_K3, 64 * dec_grp_waves(2), lds_dec_grp(2));
#undef PV_K3
        } else if (e->ngrp == 4) {
#define PV_K3(TR, MD) pvk_dec_grp<4, TR, MD, dec_grp_waves(4)>
            PV_DEC_MODES(PV_K3, 64 * dec_grp_waves(4), lds_dec_grp(4));
#undef PV_K3
        } else {
#define PV_K3(TR, MD) pvk_dec<N_, TR, MD>
            if (mode == 0 && e->dec2_wgs && e->split_now) {  // the pass in two launches, the second beside the backward chain
                PV_DISPATCH_NWB(e, hipLaunchKernelGGL((pvk_dec2<N_, 1>), dim3(e->dec2_wgs), dim3(PvDec2<N_>::NW * 64), PvDec2<N_>::lds(),
                                                      e->stream, aV, amV, avV, aV2, amV2, avV2, acb, amC, avC, gD, e->xt_act, e->nblk,
                                                      (int)e->L, e->st, e->adam, e->dgpart, e->losspart, e->ts4));
                HIPCHK(hipEventRecord(e->ev_da, e->stream));
                HIPCHK(hipStreamWaitEvent(e->side, e->ev_da, 0));
                {
                    Timer t2(e, C_DEC_UPD, e->side);
                    PV_DISPATCH_NWB(e, hipLaunchKernelGGL((pvk_dec2<N_, 2>), dim3(e->dec2_wgs), dim3(PvDec2<N_>::NW * 64), PvDec2<N_>::lds(),
                                                          e->side, aV, amV, avV, aV2, amV2, avV2, acb, amC, avC, gD, e->xt_act, e->nblk,
                                                          (int)e->L, e->st, e->adam, e->dgpart, e->losspart, e->ts4));
                }
                HIPCHK(hipEventRecord(e->ev_upd, e->side));
                e->upd_pending = true;
            } else if (mode == 0 && e->dec2_wgs) {  // plain training pass: block-column split
                PV_DISPATCH_NWB(e, hipLaunchKernelGGL(pvk_dec2<N_>, dim3(e->dec2_wgs), dim3(PvDec2<N_>::NW * 64), PvDec2<N_>::lds(),
                                                      e->stream, aV, amV, avV, aV2, amV2, avV2, acb, amC, avC, gD, e->xt_act, e->nblk,
                                                      (int)e->L, e->st, e->adam, e->dgpart, e->losspart, e->ts4));
            } else if (mode == 1) {  // read-only pass: small LDS footprint, its own (larger) grid
                PV_DISPATCH_NWB(e, hipLaunchKernelGGL((pvk_dec<N_, 0, PV_MODE_TRAIN>), dim3(e->dec_eval_wgs), dim3(256), lds_dec_eval(N_),
                                                      e->stream, aV, amV, avV, aV2, amV2, avV2, acb, amC, avC, gD, e->xt_act, e->nblk,
                                                      (int)e->L, e->st, e->adam, e->dgpart, e->losspart, e->ts4, agV, agC));
            } else {
                PV_DISPATCH_NWB(e, PV_DEC_MODES(PV_K3, 256, lds_dec(N_)));
            }
#undef PV_K3
        }
#undef PV_DEC_MODES
#undef PV_DEC_LAUNCH
        if (upd) e->Vout.swap();
        if ((rc = launch_check(e, "dec"))) return rc;
        t.reset();
        Timer tr(e, C_REDUCE);
        const int n_dg = mode == 1 ? 0 : 32 * e->Wp;
        // eval: only the recon sums are produced (they live behind the dg block of the exchange buffer)
        hipLaunchKernelGGL(pvk_reduce_parts, dim3(PV_REDUCE_GRID(n_dg)), dim3(256), 0, e->stream, e->dgpart,
                           mode == 1 ? e->dec_eval_wgs : (mode == 0 && e->dec2_wgs) ? e->dec2_wgs : e->dec_wgs, n_dg, 1.0f, e->xch1,
                           e->losspart, e->xch1 + 32 * e->Wp);
        if ((rc = launch_check(e, "dec_reduce"))) return rc;
    }
    return 0;
}

int enc_bwd_launch(pvae_engine* e, int km, bool fuse = false, int next_off = 0, int next_n = 0, const PvBwdOp* adam_ops = nullptr,
                   const uint64_t* Xn = nullptr) {
    int rc;
    Timer t(e, C_ENC_BWD);
    const uint8_t* nog = nullptr;
    const int* nor = nullptr;
    float* nof = nullptr;
    const float* nogw = nullptr;
    const PvBwdOp* noops = nullptr;
    // plain training: the narrow layers' dW + Adam units take the first workgroups of the launch (one unit per wave)
    const int adam_units = adam_ops ? 16 * e->n_bwd_ops : 0, adam_wgs = (adam_units + 3) / 4;
    if (km == PV_MODE_GRAD) {
        if (!e->W1.g && (rc = allocf(e, &e->W1.g, e->W1.n))) return rc;
        PV_DISPATCH_NWB(e, hipLaunchKernelGGL((pvk_enc_bwd<N_, PV_MODE_GRAD, 0>), dim3(e->encb_wgs), dim3(256),
                                              lds_enc_bwd(e->Wp), e->stream, e->W1.w, (float*)nullptr, (float*)nullptr,
                                              (float*)nullptr, e->W1.g, (float*)nullptr, e->dpre1, e->Wp, e->xt_act, e->nblk,
                                              e->ngrp, e->st, e->adam, nog, (int64_t)0, nor, 0, 0, nof, e->ts4, nogw, noops, 0));
    } else if (km == PV_MODE_TRAIN_ACC) {
        if (!e->W1.g) return set_err(e, -6, "train-with-accumulated-gradients step without a gradient dump");
        PV_DISPATCH_NWB(e, hipLaunchKernelGGL((pvk_enc_bwd<N_, PV_MODE_TRAIN_ACC, 0>), dim3(e->encb_wgs), dim3(256),
                                              lds_enc_bwd(e->Wp), e->stream, e->W1.w, e->W1.m, e->W1.v, e->W1.w2, e->W1.m2,
                                              e->W1.v2, e->dpre1, e->Wp, e->xt_act, e->nblk, e->ngrp, e->st, e->adam, nog,
                                              (int64_t)0, nor, 0, 0, nof, e->ts4, (const float*)e->W1.g, noops, 0));
        e->W1.swap();
    } else if (fuse) {
        PV_DISPATCH_NWB(e, hipLaunchKernelGGL((pvk_enc_bwd<N_, PV_MODE_TRAIN, 1>), dim3(e->encb_wgs + adam_wgs), dim3(256),
                                              lds_enc_bwd(e->Wp), e->stream, e->W1.w, e->W1.m, e->W1.v, e->W1.w2, e->W1.m2,
                                              e->W1.v2, e->dpre1, e->Wp, e->xt_act, e->nblk, e->ngrp, e->st, e->adam,
                                              (const uint8_t*)e->geno, e->row_bytes, (const int*)e->d_rows, next_off, next_n,
                                              e->encpart, e->ts4, nogw, adam_ops, adam_units, Xn));
        e->W1.swap();
    } else {
        PV_DISPATCH_NWB(e, hipLaunchKernelGGL((pvk_enc_bwd<N_, PV_MODE_TRAIN, 0>), dim3(e->encb_wgs + adam_wgs), dim3(256),
                                              lds_enc_bwd(e->Wp), e->stream, e->W1.w, e->W1.m, e->W1.v, e->W1.w2, e->W1.m2,
                                              e->W1.v2, e->dpre1, e->Wp, e->xt_act, e->nblk, e->ngrp, e->st, e->adam, nog,
                                              (int64_t)0, nor, 0, 0, nof, e->ts4, nogw, adam_ops, adam_units));
        e->W1.swap();
    }
    return launch_check(e, "enc_bwd");
}

int ensure_grad_bufs(pvae_engine* e) {
    int rc = 0;
    HIPCHK(hipStreamSynchronize(e->stream));  // the op tables are rewritten below
    for_each_buf(e, [&](Buf3& b) { if (!rc && !b.g) rc = allocf(e, &b.g, b.n); });
    if (rc) return rc;
    // both op tables carry the g pointers (TRAIN_ACC reads them through the train table)
    if ((rc = build_bwd_ops(e, false, &e->d_bwd_ops))) return rc;
    return build_bwd_ops(e, true, &e->d_bwd_ops_grad);
}

// stage C: loss, narrow backward (+Adam), wide encoder backward (+Adam).  mode: 0 train, 1 eval, 2 gradient dump,
// 4 train on (this pass + dumped gradients)
int stage_c(pvae_engine* e, int mode, float* d_out3, bool fuse = false, int next_off = 0, int next_n = 0,
            const uint64_t* Xn = nullptr) {
    int rc;
    const int km = mode == 2 ? PV_MODE_GRAD : mode == 4 ? PV_MODE_TRAIN_ACC : PV_MODE_TRAIN;
    if ((mode == 2 || mode == 4) && !e->d_bwd_ops_grad && (rc = ensure_grad_bufs(e))) return rc;
    const PvBwdOp* table = km == PV_MODE_GRAD ? e->d_bwd_ops_grad : e->d_bwd_ops;
    if (e->fused_narrow) {
        if ((rc = ensure_narrow_packed(e))) return rc;
        // plain training: the narrow layers' dW + Adam units ride on the launch of the wide encoder backward (its first
        // workgroups, one unit per wave) instead of an 8 us launch of their own on an otherwise idle chip
        const bool ride = mode == 0;
        {
        Timer t(e, C_NARROW_BWD);
        const dim3 chain_grid(mode == 1 ? 1 : e->narrow_warm ? PVQ_WORKERS_PER_TILE + 8 * e->n_bwd_ops : PVQ_WORKERS_PER_TILE);
#define PV_NARROW_BWD(MD)                                                                                                      \
    hipLaunchKernelGGL(pvk_narrow_bwd<MD>, chain_grid, dim3(PVQ_THREADS), pvn_lds_bwd(), e->stream, table, e->n_bwd_ops, e->xch1, \
                       e->Wp, e->kl, e->eps, e->headout, e->Kh, e->K, e->st, e->adam, d_out3, mode == 1 ? 0 : 1, (int)PVQ_WORKERS_PER_TILE)
        if (km == PV_MODE_GRAD) PV_NARROW_BWD(PV_MODE_GRAD);
        else if (km == PV_MODE_TRAIN_ACC) PV_NARROW_BWD(PV_MODE_TRAIN_ACC);
        else PV_NARROW_BWD(PV_MODE_TRAIN);
#undef PV_NARROW_BWD
        if (km == PV_MODE_GRAD)
            hipLaunchKernelGGL(pvk_narrow_adam<PV_MODE_GRAD>, dim3(16, e->n_bwd_ops), dim3(64), 0, e->stream, table, e->st, e->adam, 1);
        else if (km == PV_MODE_TRAIN_ACC)
            hipLaunchKernelGGL(pvk_narrow_adam<PV_MODE_TRAIN_ACC>, dim3(16, e->n_bwd_ops), dim3(64), 0, e->stream, table, e->st, e->adam, 1);
        if (mode == 0 && !ride)
            hipLaunchKernelGGL(pvk_narrow_adam<PV_MODE_TRAIN>, dim3(16, e->n_bwd_ops), dim3(64), 0, e->stream, table, e->st, e->adam, 1);
        if ((rc = launch_check(e, "narrow_bwd"))) return rc;
        }
        if (mode == 1) return 0;
        return enc_bwd_launch(e, km, fuse, next_off, next_n, ride ? table : nullptr, Xn);
    }
    if (mode == 1) {  // evaluation: the loss scalars are all there is left to do
        hipLaunchKernelGGL(pvk_loss_final, dim3(1), dim3(64), 0, e->stream, e->xch1 + 32 * e->Wp, e->kl, e->st, d_out3);
        return launch_check(e, "loss_final");
    }
    {   // (training: the loss scalars ride on the last launch of the chain, pvk_narrow_adam_bias)
        Timer t(e, C_NARROW_BWD);
        const int D = e->D;
        // the chain: per layer dpre (kept in dpre_bufs, in op-table order) and dIn; dY ping-pongs between xch1/dA/dB
        const float* dY = e->xch1;
        float* bufs[2] = {e->dA, e->dB};
        int pp = 0;
        size_t li = 0;
        // Consecutive hidden layers chain through dpre: the producer's epilogue forms the consumer's dpre in the consumer's dpre
        // buffer (what the Adam launch reads) and the consumer stages it as it comes -- no pass over Y and no act' in between
        bool chained = false;  // dY is already the dpre of the layer about to run
        for (int i = D - 1; i >= 1; --i) {
            const bool to_next = i - 1 >= 1 && dense16_bwd_ok(e, e->dec[i]) && dense16_bwd_ok(e, e->dec[i - 1]);
            float* dst = to_next ? e->dpre_bufs[li + 1] : bufs[pp];
            if ((rc = run_dense_bwd(e, e->dec[i], e->g[i], dY, dst, e->dpre_bufs[li], PvRepBwd{}, chained,
                                    to_next ? e->g[i - 1] : nullptr, e->dec[i - 1].act))) return rc;
            ++li;
            dY = dst;
            chained = to_next;
            if (!to_next) pp ^= 1;
        }
        if (e->Kz == 32) {  // one workgroup: dz -> d heads in the tail
            const PvRepBwd rp{e->headout, e->eps, e->dheads, e->Kh, e->K};
            if ((rc = run_dense_bwd(e, e->dec[0], e->g[0], dY, e->dz, e->dpre_bufs[li++], rp))) return rc;
        } else {
            if ((rc = run_dense_bwd(e, e->dec[0], e->g[0], dY, e->dz, e->dpre_bufs[li++]))) return rc;
            hipLaunchKernelGGL(pvk_reparam_bwd, dim3(1), dim3(64), 0, e->stream, e->headout, e->Kh, e->K, e->eps, e->dz, e->Kz,
                               e->dheads, e->st);
            if ((rc = launch_check(e, "reparam_bwd"))) return rc;
        }
        if ((rc = run_dense_bwd(e, e->heads, e->headout, e->dheads, bufs[pp], e->dpre_bufs[li++]))) return rc;
        dY = bufs[pp];
        pp ^= 1;
        chained = false;
        for (int i = D - 2; i >= 0; --i) {
            const bool to_next = i - 1 >= 0 && dense16_bwd_ok(e, e->enc[i]) && dense16_bwd_ok(e, e->enc[i - 1]);
            float* dst = to_next ? e->dpre_bufs[li + 1] : bufs[pp];
            if ((rc = run_dense_bwd(e, e->enc[i], e->h[i + 1], dY, dst, e->dpre_bufs[li], PvRepBwd{}, chained,
                                    to_next ? e->h[i] : nullptr, to_next ? e->enc[i - 1].act : (int)PV_ACT_LINEAR))) return rc;
            ++li;
            dY = dst;
            chained = to_next;
            if (!to_next) pp ^= 1;
        }
        // weight / bias gradients + Adam of all narrow layers: one launch, one wave per 32x32 block
        // one launch: rows 0 .. n_bwd_ops-1 of the grid are the Adam units, the last row the first (wide) layer's dpre / bias
        const int adam_x = (e->Wp / 32) * (e->Wp / 32);
        const dim3 agrid(std::max(adam_x, (e->Wp + 63) / 64 + 1), e->n_bwd_ops + 1);
        if (km == PV_MODE_GRAD) {
            hipLaunchKernelGGL(pvk_narrow_adam_bias<PV_MODE_GRAD>, agrid, dim3(64), 0, e->stream, table, e->st, e->adam, e->h[0], dY,
                               e->Wp, (int)PV_ACT_ELU, e->dpre1, e->b1.w, e->b1.g, (float*)nullptr, (const float*)nullptr, (const float*)(e->xch1 + 32 * e->Wp),
                               (const float*)e->kl, d_out3, adam_x);
        } else if (km == PV_MODE_TRAIN_ACC) {
            hipLaunchKernelGGL(pvk_narrow_adam_bias<PV_MODE_TRAIN_ACC>, agrid, dim3(64), 0, e->stream, table, e->st, e->adam, e->h[0], dY,
                               e->Wp, (int)PV_ACT_ELU, e->dpre1, e->b1.w, e->b1.m, e->b1.v, (const float*)e->b1.g, (const float*)(e->xch1 + 32 * e->Wp),
                               (const float*)e->kl, d_out3, adam_x);
        } else {
            hipLaunchKernelGGL(pvk_narrow_adam_bias<PV_MODE_TRAIN>, agrid, dim3(64), 0, e->stream, table, e->st, e->adam, e->h[0], dY,
                               e->Wp, (int)PV_ACT_ELU, e->dpre1, e->b1.w, e->b1.m, e->b1.v, (const float*)nullptr, (const float*)(e->xch1 + 32 * e->Wp),
                               (const float*)e->kl, d_out3, adam_x);
        }
        if ((rc = launch_check(e, "narrow_adam/bias_bwd"))) return rc;
    }
    return enc_bwd_launch(e, km, fuse, next_off, next_n, nullptr, Xn);
}

int ensure_rows(pvae_engine* e, int64_t n) {
    if (n <= e->rows_cap) return 0;
    if (e->d_rows) hipFree(e->d_rows);
    e->rows_cap = std::max<int64_t>(n, 4096);
    HIPCHK(hipMalloc(&e->d_rows, e->rows_cap * sizeof(int)));
    return 0;
}
int ensure_loss(pvae_engine* e, int64_t steps) {
    if (steps <= e->loss_cap) return 0;
    if (e->d_loss) hipFree(e->d_loss);
    e->loss_cap = std::max<int64_t>(steps, 1024);
    HIPCHK(hipMalloc(&e->d_loss, e->loss_cap * 3 * sizeof(float)));
    return 0;
}

int check_rows(pvae_engine* e, const int32_t* rows, int64_t n) {
    for (int64_t i = 0; i < n; ++i)
        if (rows[i] < 0 || rows[i] >= e->n_samples)
            return set_err(e, -4, "row index %d out of range [0, %lld)", rows[i], (long long)e->n_samples);
    return 0;
}

// upload rows/eps for a single batch and run pvk_step_begin
constexpr size_t kSlotBytes = 128 * sizeof(int) + 128 * 64 * sizeof(float);  // 128 rows + their noise at latent_dim <= 64
// noise_per_32: draw the engine's noise with one generator call per 32 rows (the calls a batch-by-batch pass over the same rows
// would make: the stream position after the call does not depend on how many rows went through together)
int begin_batch(pvae_engine* e, const int32_t* rows, int32_t n, const float* eps, bool train, int max_rows = 32,
                bool noise_per_32 = false) {
    int rc;
    if (n <= 0 || n > e->cfg.batch_size) return set_err(e, -2, "batch of %d rows (batch_size %d)", n, e->cfg.batch_size);
    if (n > max_rows) return set_err(e, -2, "a device pass takes at most %d rows (got %d): larger batches go through pvae_train_step / "
                                            "pvae_train_epoch, which tile them", max_rows, n);
    if (!e->geno) return set_err(e, -3, "no genotypes loaded (pvae_load_genotypes)");
    if ((rc = check_rows(e, rows, n))) return rc;
    if ((rc = ensure_rows(e, 128))) return rc;
    pvae_engine::Slot& sl = e->slots[e->slot_next];  // pinned ring, allocated by pvae_create
    e->slot_next = (e->slot_next + 1) % 8;
    if (sl.pending) HIPCHK(hipEventSynchronize(sl.ev));
    memcpy(sl.host, rows, n * sizeof(int));
    if (eps) memcpy(sl.host + 128 * sizeof(int), eps, (size_t)n * e->K * sizeof(float));
    HIPCHK(hipMemcpyAsync(e->d_rows, sl.host, n * sizeof(int), hipMemcpyHostToDevice, e->stream));
    if (eps || noise_per_32)
        hipLaunchKernelGGL(pvk_step_begin, dim3(1), dim3(64), 0, e->stream, e->st, (const int*)nullptr, n, e->cfg.batch_size, 0,
                           train ? 1 : 0, e->adam.b1, e->adam.b2);
    else
        hipLaunchKernelGGL(pvk_step_begin_eps, dim3(1), dim3(1024), 0, e->stream, e->st, (const int*)nullptr, n,
                           e->cfg.batch_size, 0, train ? 1 : 0, e->adam.b1, e->adam.b2, e->eps, max_rows * e->K, e->cfg.seed,
                           e->d_counter);
    if (!eps && noise_per_32)
        for (int t = 0; 32 * t < n; ++t)
            hipLaunchKernelGGL(pvk_gen_eps, dim3(1), dim3(1024), 0, e->stream, e->eps + (size_t)32 * t * e->K, 32 * e->K, e->cfg.seed,
                               e->d_counter);
    if ((rc = launch_check(e, "step_begin"))) return rc;
    if (eps) {
        HIPCHK(hipMemcpyAsync(e->eps, sl.host + 128 * sizeof(int), (size_t)n * e->K * sizeof(float), hipMemcpyHostToDevice,
                              e->stream));
        HIPCHK(hipEventRecord(sl.ev, e->stream));
        sl.pending = true;
    } else {
        HIPCHK(hipEventRecord(sl.ev, e->stream));
        sl.pending = true;
    }
    return 0;
}

int fetch_out3(pvae_engine* e, float* out3) {
    if (!out3) return 0;
    HIPCHK(hipMemcpyAsync(e->h_out3, e->d_loss, 3 * sizeof(float), hipMemcpyDeviceToHost, e->stream));
    HIPCHK(hipStreamSynchronize(e->stream));
    memcpy(out3, e->h_out3, 3 * sizeof(float));
    return 0;
}

// A train step on 33 .. 32 T rows in ONE pass over the wide weights (rows on the device at d_rows[row_offset ..], PvStep and
// the noise set up by the caller): every kernel of the step sees T row tiles -- the wide kernels multiply the block they
// hold with each tile's codes and sum dW / dV over the tiles before the one Adam update, the chain runs 2 T workers.
template <int T>
int multi_tile_launches(pvae_engine* e, int row_offset, float* d_out3, int mode) {
    // mode: 0 = the whole batch (Adam in the pass); chunks of a batch above 32 T rows: 2 = gradient dump (PvStep.acc: store / add),
    // 4 = last chunk (Adam on this pass + the dump) -- the conventions of the 32-row tile path (run_tile)
    int rc;
    const int R = 32 * T, nR = R * e->Wp;
    const int km = mode == 2 ? PV_MODE_GRAD : mode == 4 ? PV_MODE_TRAIN_ACC : PV_MODE_TRAIN;
    if (mode != 0 && (!e->d_bwd_ops_grad || !e->Vout.g || !e->W1.g) && (rc = ensure_grad_bufs(e))) return rc;
    {
        Timer t(e, C_GATHER);
        hipLaunchKernelGGL(pvk_gather_transpose, dim3(((e->nblk + 1) / 2 + 3) / 4, T), dim3(256), 0, e->stream, e->geno, e->row_bytes,
                           e->d_rows, e->st, row_offset, e->nblk, e->Xt, e->Xr);
        if ((rc = launch_check(e, "gather_transpose"))) return rc;
    }
    if ((int64_t)e->encf_mt_wgs * nR > e->encpart_cap || (int64_t)e->dec3_wgs * nR > e->dgpart_cap ||
        (int64_t)T * e->dece_mt_wgs > e->losspart_cap || e->dec3_wgs > e->losspart_cap)
        return set_err(e, -6, "multi-tile pass: partial-sum buffers smaller than the grids (%d / %d / %d workgroups)", e->encf_mt_wgs,
                       e->dec3_wgs, e->dece_mt_wgs);
    {
        Timer t(e, C_ENC_FWD);
        hipLaunchKernelGGL((pvk_enc_fwd<4, T, PV_MT_BF16>), dim3(e->encf_mt_wgs), dim3(256), kEncFwdLds, e->stream, e->W1.w, e->geno, e->row_bytes,
                           e->d_rows, e->st, row_offset, e->nblk, e->ngrp, e->encpart, e->ts4, (const uint64_t*)e->Xr);
        hipLaunchKernelGGL(pvk_reduce_parts, dim3(PV_REDUCE_GRID(nR)), dim3(256), 0, e->stream, e->encpart, e->encf_mt_wgs, nR, 0.5f,
                           e->P, (const float*)nullptr, (float*)nullptr);
        if ((rc = launch_check(e, "enc_fwd"))) return rc;
    }
    if ((rc = shard_allreduce(e, 0, R))) return rc;
    {
        Timer t(e, C_NARROW_FWD);
        if ((rc = narrow_fused_fwd(e, false, false, T))) return rc;
    }
    {
        Timer t(e, C_DEC);
        const int wgs = e->dec3_wgs;
        float* nul = nullptr;
        const bool upd = mode != 2;
#define PV_DEC3_LAUNCH(MD)                                                                                                          \
    hipLaunchKernelGGL((pvk_dec3<T, MD>), dim3(wgs), dim3(256), PvDec3<T>::lds(), e->stream, (const float*)e->Vout.w,                \
                       (const float*)(upd ? e->Vout.m : nul), (const float*)(upd ? e->Vout.v : nul), upd ? e->Vout.w2 : nul,         \
                       upd ? e->Vout.m2 : e->Vout.g, upd ? e->Vout.v2 : nul, e->cout.w, upd ? e->cout.m : e->cout.g,                 \
                       upd ? e->cout.v : nul, (const float*)e->g[e->D - 1], (const uint64_t*)e->Xt, e->nblk, (int)e->L,              \
                       (const PvStep*)e->st, e->adam, e->dgpart, e->losspart, e->ts4, (const float*)(mode == 4 ? e->Vout.g : nul),   \
                       (const float*)(mode == 4 ? e->cout.g : nul))
        // four tiles: logits and dV on the bf16 matrix pipe at f32 accuracy (pvk_dec3b); PVAE_NO_DEC3B=1: the fp32 kernel
#define PV_DEC3B_LAUNCH(MD)                                                                                                         \
    hipLaunchKernelGGL((pvk_dec3b<MD>), dim3(wgs), dim3(256), PvDec3b::lds(), e->stream, (const float*)e->Vout.w,                    \
                       (const float*)(upd ? e->Vout.m : nul), (const float*)(upd ? e->Vout.v : nul), upd ? e->Vout.w2 : nul,         \
                       upd ? e->Vout.m2 : e->Vout.g, upd ? e->Vout.v2 : nul, e->cout.w, upd ? e->cout.m : e->cout.g,                 \
                       upd ? e->cout.v : nul, (const float*)e->g[e->D - 1], (const uint64_t*)e->Xt, e->nblk, (int)e->L,              \
                       (const PvStep*)e->st, e->adam, e->dgpart, e->losspart, e->ts4, (const float*)(mode == 4 ? e->Vout.g : nul),   \
                       (const float*)(mode == 4 ? e->cout.g : nul))
        if (T == 4 && PV_MT_BF16 && e->dec3b) {
            if (km == PV_MODE_GRAD) PV_DEC3B_LAUNCH(PV_MODE_GRAD);
            else if (km == PV_MODE_TRAIN_ACC) PV_DEC3B_LAUNCH(PV_MODE_TRAIN_ACC);
            else PV_DEC3B_LAUNCH(PV_MODE_TRAIN);
        } else if (km == PV_MODE_GRAD) PV_DEC3_LAUNCH(PV_MODE_GRAD);
        else if (km == PV_MODE_TRAIN_ACC) PV_DEC3_LAUNCH(PV_MODE_TRAIN_ACC);
        else PV_DEC3_LAUNCH(PV_MODE_TRAIN);
#undef PV_DEC3B_LAUNCH
#undef PV_DEC3_LAUNCH
        if (upd) e->Vout.swap();
        hipLaunchKernelGGL(pvk_reduce_parts, dim3(PV_REDUCE_GRID(nR)), dim3(256), 0, e->stream, e->dgpart, wgs, nR, 1.0f,
                           e->xch1, e->losspart, e->xch1 + nR);
        if ((rc = launch_check(e, "dec"))) return rc;
    }
    if ((rc = shard_allreduce(e, 1, R))) return rc;
    const PvBwdOp* table = km == PV_MODE_GRAD ? e->d_bwd_ops_grad : e->d_bwd_ops;
    {
        Timer t(e, C_NARROW_BWD);
        if ((rc = ensure_narrow_packed(e))) return rc;
        const dim3 chain_grid(e->narrow_warm ? PVQ_WORKERS_PER_TILE * T + 8 * e->n_bwd_ops : PVQ_WORKERS_PER_TILE * T);
#define PV_NARROW_BWD_T(MD)                                                                                                        \
    hipLaunchKernelGGL(pvk_narrow_bwd<MD>, chain_grid, dim3(PVQ_THREADS), pvn_lds_bwd(), e->stream, table, e->n_bwd_ops, e->xch1,   \
                       e->Wp, e->kl, e->eps, e->headout, e->Kh, e->K, e->st, e->adam, d_out3, 1, (int)PVQ_WORKERS_PER_TILE * T)
        if (km == PV_MODE_GRAD) {
            PV_NARROW_BWD_T(PV_MODE_GRAD);
            hipLaunchKernelGGL(pvk_narrow_adam<PV_MODE_GRAD>, dim3(16, e->n_bwd_ops), dim3(64), 0, e->stream, table, e->st, e->adam, T);
        } else if (km == PV_MODE_TRAIN_ACC) {
            PV_NARROW_BWD_T(PV_MODE_TRAIN_ACC);
            hipLaunchKernelGGL(pvk_narrow_adam<PV_MODE_TRAIN_ACC>, dim3(16, e->n_bwd_ops), dim3(64), 0, e->stream, table, e->st, e->adam, T);
        } else {
            PV_NARROW_BWD_T(PV_MODE_TRAIN);
        }
#undef PV_NARROW_BWD_T
        if ((rc = launch_check(e, "narrow_bwd"))) return rc;
    }
    {
        Timer t(e, C_ENC_BWD);
        const uint8_t* nog = nullptr;
        const int* nor = nullptr;
        float* nof = nullptr;
        const PvBwdOp* noops = nullptr;
        if (km == PV_MODE_GRAD) {
            hipLaunchKernelGGL((pvk_enc_bwd<4, PV_MODE_GRAD, 0, T>), dim3(e->encb_wgs), dim3(256), T * lds_enc_bwd(e->Wp), e->stream,
                               e->W1.w, (float*)nullptr, (float*)nullptr, (float*)nullptr, e->W1.g, (float*)nullptr, e->dpre1, e->Wp, e->Xt,
                               e->nblk, e->ngrp, e->st, e->adam, nog, (int64_t)0, nor, 0, 0, nof, e->ts4, (const float*)nullptr, noops, 0);
        } else if (km == PV_MODE_TRAIN_ACC) {
            hipLaunchKernelGGL((pvk_enc_bwd<4, PV_MODE_TRAIN_ACC, 0, T>), dim3(e->encb_wgs), dim3(256), T * lds_enc_bwd(e->Wp), e->stream,
                               e->W1.w, e->W1.m, e->W1.v, e->W1.w2, e->W1.m2, e->W1.v2, e->dpre1, e->Wp, e->Xt, e->nblk, e->ngrp, e->st,
                               e->adam, nog, (int64_t)0, nor, 0, 0, nof, e->ts4, (const float*)e->W1.g, noops, 0);
            e->W1.swap();
        } else {
            const int adam_units = 16 * e->n_bwd_ops, adam_wgs = (adam_units + 3) / 4;
            hipLaunchKernelGGL((pvk_enc_bwd<4, PV_MODE_TRAIN, 0, T>), dim3(e->encb_wgs + adam_wgs), dim3(256), T * lds_enc_bwd(e->Wp),
                               e->stream, e->W1.w, e->W1.m, e->W1.v, e->W1.w2, e->W1.m2, e->W1.v2, e->dpre1, e->Wp, e->Xt, e->nblk, e->ngrp,
                               e->st, e->adam, nog, (int64_t)0, nor, 0, 0, nof, e->ts4, (const float*)nullptr,
                               (const PvBwdOp*)e->d_bwd_ops, adam_units);
            e->W1.swap();
        }
        if ((rc = launch_check(e, "enc_bwd"))) return rc;
    }
    return 0;
}
int multi_tile_train(pvae_engine* e, int row_offset, float* d_out3, int mode = 0) {
    return e->T == 2 ? multi_tile_launches<2>(e, row_offset, d_out3, mode) : multi_tile_launches<4>(e, row_offset, d_out3, mode);
}
// chunk c of a batch of n_batch rows above 32 T (rows on the device at d_rows[row_offset ..], noise in place): nrows rows,
// gradients dumped / accumulated, Adam in the last chunk -- one pass over the wide weights per 32 T rows
int run_chunk(pvae_engine* e, int row_offset, int nrows, int n_batch, int c, bool last, float* d_out3) {
    int rc;
    hipLaunchKernelGGL(pvk_tile_begin, dim3(1), dim3(64), 0, e->stream, e->st, nrows, 1.f / (float)n_batch, c > 0 ? 1 : 0);
    if ((rc = launch_check(e, "tile_begin"))) return rc;
    return multi_tile_train(e, row_offset, d_out3, last ? 4 : 2);
}

// evaluation of up to 32 T rows of a batch of n_batch (validation batches above 32 rows): one read of the wide weights for all
// of them -- first-layer forward of T tiles, the chain's forward on 2 T workers, the read-only decoder pass, the loss scalars
// nbatches > 0: the rows are nbatches validation batches of 32 rows (the last one may be short), each with a loss of its own ->
// d_out3[3 t ..]; nbatches == 0: one batch of nrows rows out of n_batch (1 / n_batch weights its loss part)
template <int T>
int multi_tile_eval_launches(pvae_engine* e, int row_offset, int nrows, int n_batch, float* d_out3, int nbatches = 0) {
    int rc;
    const int R = 32 * T, nR = R * e->Wp;
    hipLaunchKernelGGL(pvk_tile_begin, dim3(1), dim3(64), 0, e->stream, e->st, nrows, 1.f / (float)n_batch, 0);
    if ((rc = launch_check(e, "tile_begin"))) return rc;
    {
        Timer t(e, C_GATHER);
        hipLaunchKernelGGL(pvk_gather_transpose, dim3(((e->nblk + 1) / 2 + 3) / 4, T), dim3(256), 0, e->stream, e->geno, e->row_bytes,
                           e->d_rows, e->st, row_offset, e->nblk, e->Xt, e->Xr);
        if ((rc = launch_check(e, "gather_transpose"))) return rc;
    }
    if ((int64_t)e->encf_mt_wgs * nR > e->encpart_cap || (int64_t)e->dec3_wgs * nR > e->dgpart_cap ||
        (int64_t)T * e->dece_mt_wgs > e->losspart_cap || e->dec3_wgs > e->losspart_cap)
        return set_err(e, -6, "multi-tile pass: partial-sum buffers smaller than the grids (%d / %d / %d workgroups)", e->encf_mt_wgs,
                       e->dec3_wgs, e->dece_mt_wgs);
    {
        Timer t(e, C_ENC_FWD);
        hipLaunchKernelGGL((pvk_enc_fwd<4, T, PV_MT_BF16>), dim3(e->encf_mt_wgs), dim3(256), kEncFwdLds, e->stream, e->W1.w, e->geno, e->row_bytes,
                           e->d_rows, e->st, row_offset, e->nblk, e->ngrp, e->encpart, e->ts4, (const uint64_t*)e->Xr);
        hipLaunchKernelGGL(pvk_reduce_parts, dim3(PV_REDUCE_GRID(nR)), dim3(256), 0, e->stream, e->encpart, e->encf_mt_wgs, nR, 0.5f,
                           e->P, (const float*)nullptr, (float*)nullptr);
        if ((rc = launch_check(e, "enc_fwd"))) return rc;
    }
    if ((rc = shard_allreduce(e, 0, R))) return rc;
    {
        Timer t(e, C_NARROW_FWD);
        if ((rc = narrow_fused_fwd(e, false, false, T))) return rc;
    }
    if (nbatches > 0) {  // one loss per tile
        Timer t(e, C_DEC);
        if (PV_MT_BF16)
            hipLaunchKernelGGL((pvk_dec_eval_bf<T, 1>), dim3(e->dece_mt_wgs), dim3(512), PvDecEvalBf<T>::lds(), e->stream,
                               (const float*)e->Vout.w, (const float*)e->cout.w, (const float*)e->g[e->D - 1], (const uint64_t*)e->Xt,
                               e->nblk, (int)e->L, (const PvStep*)e->st, e->losspart, e->ts4);
        else
        hipLaunchKernelGGL((pvk_dec_eval_mt<T, 1>), dim3(e->dece_mt_wgs), dim3(256), lds_dec_eval_mt(T), e->stream, (const float*)e->Vout.w,
                           (const float*)e->cout.w, (const float*)e->g[e->D - 1], (const uint64_t*)e->Xt, e->nblk, (int)e->L,
                           (const PvStep*)e->st, e->losspart, e->ts4);
        const bool sharded = e->cfg.world != 1 || has_comm(e);
        hipLaunchKernelGGL(pvk_loss_tiles, dim3(1), dim3(256), 0, e->stream, (const float*)e->losspart, e->dece_mt_wgs, nbatches,
                           (const float*)e->kl, (const PvStep*)e->st, e->xch1 + nR, d_out3, sharded ? 1 : 3);
        if ((rc = launch_check(e, "dec_eval"))) return rc;
        if (!sharded) return 0;
        if ((rc = shard_allreduce(e, 1, R))) return rc;
        hipLaunchKernelGGL(pvk_loss_tiles, dim3(1), dim3(256), 0, e->stream, (const float*)e->losspart, e->dece_mt_wgs, nbatches,
                           (const float*)e->kl, (const PvStep*)e->st, e->xch1 + nR, d_out3, 2);
        return launch_check(e, "loss");
    }
    {
        Timer t(e, C_DEC);
        if (PV_MT_BF16)
            hipLaunchKernelGGL((pvk_dec_eval_bf<T, 0>), dim3(e->dece_mt_wgs), dim3(512), PvDecEvalBf<T>::lds(), e->stream,
                               (const float*)e->Vout.w, (const float*)e->cout.w, (const float*)e->g[e->D - 1], (const uint64_t*)e->Xt,
                               e->nblk, (int)e->L, (const PvStep*)e->st, e->losspart, e->ts4);
        else
        hipLaunchKernelGGL((pvk_dec_eval_mt<T, 0>), dim3(e->dece_mt_wgs), dim3(256), lds_dec_eval_mt(T), e->stream, (const float*)e->Vout.w,
                           (const float*)e->cout.w, (const float*)e->g[e->D - 1], (const uint64_t*)e->Xt, e->nblk, (int)e->L,
                           (const PvStep*)e->st, e->losspart, e->ts4);
        hipLaunchKernelGGL(pvk_reduce_parts, dim3(PV_REDUCE_GRID(0)), dim3(256), 0, e->stream, e->dgpart, e->dece_mt_wgs, 0, 1.0f,
                           e->xch1, e->losspart, e->xch1 + nR);
        if ((rc = launch_check(e, "dec_eval"))) return rc;
    }
    if ((rc = shard_allreduce(e, 1, R))) return rc;
    hipLaunchKernelGGL(pvk_narrow_bwd<PV_MODE_TRAIN>, dim3(1), dim3(PVQ_THREADS), pvn_lds_bwd(), e->stream, (const PvBwdOp*)e->d_bwd_ops,
                       e->n_bwd_ops, e->xch1, e->Wp, e->kl, e->eps, e->headout, e->Kh, e->K, e->st, e->adam, d_out3, 0,
                       (int)PVQ_WORKERS_PER_TILE * T);
    return launch_check(e, "loss");
}
int multi_tile_eval(pvae_engine* e, int row_offset, int nrows, int n_batch, float* d_out3, int nbatches = 0) {
    return e->T == 2 ? multi_tile_eval_launches<2>(e, row_offset, nrows, n_batch, d_out3, nbatches)
                     : multi_tile_eval_launches<4>(e, row_offset, nrows, n_batch, d_out3, nbatches);
}

// one 32-row tile of a larger batch (rows already on the device at d_rows[row_offset..]): forward, loss part,
// gradients stored (first tile) or added (later tiles) in the g buffers; the LAST tile of a train step runs the
// training kernels, which add the dumped gradients to their own before Adam (PV_MODE_TRAIN_ACC) -- every wide
// weight / moment byte is still touched once per batch.  inv_b is 1 / (rows of the WHOLE batch), so that the
// tiles' loss parts and gradients add up to the batch mean.  mode: 1 eval, 2 gradient dump, 4 last tile of a train step
int run_tile(pvae_engine* e, int row_offset, int nrows, int n_batch, int tile, int mode, float* d_out3) {
    int rc;
    hipLaunchKernelGGL(pvk_tile_begin, dim3(1), dim3(64), 0, e->stream, e->st, nrows, 1.f / (float)n_batch, tile > 0 ? 1 : 0);
    if ((rc = launch_check(e, "tile_begin"))) return rc;
    if ((rc = stage_a(e, row_offset, true))) return rc;
    if ((rc = shard_allreduce(e, 0))) return rc;
    if ((rc = stage_b(e, mode))) return rc;
    if ((rc = shard_allreduce(e, 1))) return rc;
    return stage_c(e, mode, d_out3);
}

int full_step(pvae_engine* e, const int32_t* rows, int32_t n, const float* eps, float* out3, int mode) {
    e->promised_rows.clear();
    int rc;
    if (e->cfg.world != 1 && !has_comm(e))
        return set_err(e, -5, "world > 1: call pvae_comm_init first, or drive the step through pvae_stage_a/b/c with "
                              "all-reduces in between");
    if (n <= 0 || n > e->cfg.batch_size) return set_err(e, -2, "batch of %d rows (batch_size %d)", n, e->cfg.batch_size);
    if (n <= 32) {
        if ((rc = ensure_loss(e, 1))) return rc;
        if ((rc = begin_batch(e, rows, n, eps, mode == 0))) return rc;
        if ((rc = stage_a(e, 0, true))) return rc;
        if ((rc = shard_allreduce(e, 0))) return rc;
        if ((rc = stage_b(e, mode))) return rc;
        if ((rc = shard_allreduce(e, 1))) return rc;
        if ((rc = stage_c(e, mode, e->d_loss))) return rc;
        return fetch_out3(e, out3);
    }
    if (mode == 0 && e->T > 1 && n <= 32 * e->T) {  // one weight pass for the whole batch
        if ((rc = ensure_loss(e, 1))) return rc;
        if ((rc = begin_batch(e, rows, n, eps, true, 32 * e->T))) return rc;
        if ((rc = multi_tile_train(e, 0, e->d_loss))) return rc;
        return fetch_out3(e, out3);
    }
    const int R = 32 * e->T;
    if (mode == 1 && e->T > 1 && e->fused_narrow) {  // evaluation: one read of the wide weights per 32 T rows
        const int nc = (n + R - 1) / R;
        if ((rc = ensure_loss(e, nc))) return rc;
        for (int c = 0; c < nc; ++c) {
            const int nb = std::min(R, n - R * c);
            if ((rc = begin_batch(e, rows + R * c, nb, eps ? eps + (size_t)R * c * e->K : nullptr, false, R))) return rc;
            if ((rc = multi_tile_eval(e, 0, nb, n, e->d_loss + 3 * c))) return rc;
        }
        std::vector<float> hc((size_t)nc * 3);
        HIPCHK(hipMemcpyAsync(hc.data(), e->d_loss, hc.size() * sizeof(float), hipMemcpyDeviceToHost, e->stream));
        HIPCHK(hipStreamSynchronize(e->stream));
        if (out3)
            for (int k = 0; k < 3; ++k) {
                float a = 0.f;
                for (int c = 0; c < nc; ++c) a += hc[(size_t)c * 3 + k];
                out3[k] = a;
            }
        return 0;
    }
    const bool chunks = mode == 0 && e->T > 1;  // above 32 T rows: one weight pass per 32 T rows
    const int nt = chunks ? (n + R - 1) / R : (n + 31) / 32;
    if (chunks) {
        if ((rc = ensure_loss(e, nt))) return rc;
        for (int c = 0; c < nt; ++c) {
            const int nb = std::min(R, n - R * c);
            // begin_batch uploads this chunk's rows (+ eps) and, on the first chunk, advances Adam's t
            if ((rc = begin_batch(e, rows + R * c, nb, eps ? eps + (size_t)R * c * e->K : nullptr, c == 0, R))) return rc;
            if ((rc = run_chunk(e, 0, nb, n, c, c == nt - 1, e->d_loss + 3 * c))) return rc;
        }
    }
    if (!chunks && (rc = ensure_loss(e, nt))) return rc;
    for (int t = 0; t < nt && !chunks; ++t) {
        const int nb = std::min(32, n - 32 * t);
        // begin_batch uploads this tile's rows (+ eps) and, on the first tile of a train step, advances Adam's t
        if ((rc = begin_batch(e, rows + 32 * t, nb, eps ? eps + (size_t)32 * t * e->K : nullptr, mode == 0 && t == 0))) return rc;
        const int tm = mode == 1 ? 1 : (mode == 0 && t == nt - 1) ? 4 : 2;
        if ((rc = run_tile(e, 0, nb, n, t, tm, e->d_loss + 3 * t))) return rc;
    }
    std::vector<float> hl((size_t)nt * 3);
    HIPCHK(hipMemcpyAsync(hl.data(), e->d_loss, hl.size() * sizeof(float), hipMemcpyDeviceToHost, e->stream));
    HIPCHK(hipStreamSynchronize(e->stream));
    if (out3)
        for (int c = 0; c < 3; ++c) {
            float a = 0.f;
            for (int t = 0; t < nt; ++t) a += hl[(size_t)t * 3 + c];
            out3[c] = a;
        }
    return 0;
}

// tensor <-> storage plumbing -------------------------------------------------------------------
float* pick(Buf3& b, int which) { return which == 0 ? b.w : which == 1 ? b.m : which == 2 ? b.v : b.g; }

int tensor_io(pvae_engine* e, int index, int which, float* host, int64_t count, bool to_device) {
    if (to_device) { e->promised_rows.clear(); e->narrow_pack_dirty = true; }
    if (index < 0 || index >= (int)e->tensors.size()) return set_err(e, -2, "tensor index %d out of range", index);
    const TensorRef& t = e->tensors[index];
    const int64_t expect = t.cols ? t.rows * t.cols : t.rows;
    if (count != expect) return set_err(e, -2, "tensor %d holds %lld floats, got %lld", index, (long long)expect, (long long)count);
    HIPCHK(hipStreamSynchronize(e->stream));
    auto narrow2d = [&](float* dev, int ld, int col0) -> int {
        // Keras [rows][cols] <-> padded device [.][ld] at column col0
        if (to_device)
            HIPCHK(hipMemcpy2D(dev + col0, (size_t)ld * 4, host, (size_t)t.cols * 4, (size_t)t.cols * 4, (size_t)t.rows, hipMemcpyHostToDevice));
        else
            HIPCHK(hipMemcpy2D(host, (size_t)t.cols * 4, dev + col0, (size_t)ld * 4, (size_t)t.cols * 4, (size_t)t.rows, hipMemcpyDeviceToHost));
        return 0;
    };
    auto vec = [&](float* dev, int off) -> int {
        if (to_device) HIPCHK(hipMemcpy(dev + off, host, (size_t)t.rows * 4, hipMemcpyHostToDevice));
        else HIPCHK(hipMemcpy(host, dev + off, (size_t)t.rows * 4, hipMemcpyDeviceToHost));
        return 0;
    };
    auto need = [&](Buf3& b) -> int {
        if (!pick(b, which)) return set_err(e, -6, "tensor %d: requested buffer not available (run pvae_grad_step first)", index);
        return 0;
    };
    int rc;
    switch (t.kind) {
        case T_W1:
        case T_VOUT: {
            Buf3& b = t.kind == T_W1 ? e->W1 : e->Vout;
            if ((rc = need(b))) return rc;
            float* tmp = nullptr;
            HIPCHK(hipMalloc(&tmp, (size_t)expect * 4));
            if (to_device) HIPCHK(hipMemcpy(tmp, host, (size_t)expect * 4, hipMemcpyHostToDevice));
            const int64_t total = b.n;
            hipLaunchKernelGGL(pvk_convert_wide, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, e->stream, pick(b, which),
                               tmp, t.kind == T_W1 ? 0 : 1, to_device ? 0 : 1, e->L, e->W, e->nblk, e->ngrp, e->NWB,
                               (int)(which == 3 ? e->tile_floats : 3 * e->tile_floats));
            if ((rc = launch_check(e, "convert_wide"))) { hipFree(tmp); return rc; }
            HIPCHK(hipStreamSynchronize(e->stream));
            if (!to_device) HIPCHK(hipMemcpy(host, tmp, (size_t)expect * 4, hipMemcpyDeviceToHost));
            hipFree(tmp);
            return 0;
        }
        case T_B1: if ((rc = need(e->b1))) return rc; return vec(pick(e->b1, which), 0);
        case T_COUT: if ((rc = need(e->cout))) return rc; return vec(pick(e->cout, which), 0);
        case T_NARROW_W: {
            Dense& d = t.layer < 1000 ? e->enc[t.layer] : e->dec[t.layer - 1000];
            if ((rc = need(d.W))) return rc;
            return narrow2d(pick(d.W, which), d.nout, 0);
        }
        case T_NARROW_B: {
            Dense& d = t.layer < 1000 ? e->enc[t.layer] : e->dec[t.layer - 1000];
            if ((rc = need(d.B))) return rc;
            return vec(pick(d.B, which), 0);
        }
        case T_HEAD_W: if ((rc = need(e->heads.W))) return rc; return narrow2d(pick(e->heads.W, which), e->Kh, t.layer * e->K);
        case T_HEAD_B: if ((rc = need(e->heads.B))) return rc; return vec(pick(e->heads.B, which), t.layer * e->K);
    }
    return set_err(e, -2, "bad tensor kind");
}

// counter-based uniform in [0,1): splitmix64 of (seed, tensor, global element index)
inline float hash_uniform(uint64_t seed, uint64_t tensor, uint64_t idx) {
    uint64_t x = seed * 0x9E3779B97F4A7C15ull + tensor * 0xD1B54A32D192ED03ull + idx;
    x ^= x >> 30; x *= 0xBF58476D1CE4E5B9ull;
    x ^= x >> 27; x *= 0x94D049BB133111EBull;
    x ^= x >> 31;
    return (float)((x >> 40) + 0.5) * (1.0f / 16777216.0f);
}

}  // namespace

extern "C" {

const char* pvae_version(void) { return PV_VERSION; }
#ifdef PVN_TRACE
int pvae_debug_trace(unsigned long long* out, int n) {
    return (int)hipMemcpyFromSymbol(out, HIP_SYMBOL(pvn_trace_buf), n * sizeof(unsigned long long));
}
#endif

#ifdef PV_WTRACE
int pvae_debug_wtrace(unsigned long long* out, int n) {
    return (int)hipMemcpyFromSymbol(out, HIP_SYMBOL(pv_wtrace_buf), n * sizeof(unsigned long long));
}
#endif

int pvae_device_count(void) {
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess) return 0;
    return n;
}

static thread_local std::string g_create_err;
const char* pvae_last_error(pvae_engine* e) { return e ? e->err.c_str() : g_create_err.c_str(); }

int pvae_create(const pvae_config* cfg, pvae_engine** out) {
    if (!cfg || !out) return -1;
    *out = nullptr;
    pvae_engine* e = new pvae_engine();
    e->cfg = *cfg;
    auto bail = [&](int rc) {
        g_create_err = e->err;
        pvae_destroy(e);
        return rc;
    };
    if (cfg->n_snps_local <= 0 || cfg->width <= 0 || cfg->depth < 1 || cfg->latent_dim < 1)
        return bail(set_err(e, -2, "bad dimensions"));
    if (cfg->batch_size < 1 || cfg->batch_size > 65536)
        return bail(set_err(e, -2, "batch_size %d unsupported (1..65536)", cfg->batch_size));
    if (cfg->width > 512) return bail(set_err(e, -2, "width %d unsupported: this build handles widths up to 512", cfg->width));
    // 2 * latent_dim heads are one layer of at most 128 outputs (the narrow chain's widest tile)
    if (cfg->latent_dim > 64) return bail(set_err(e, -2, "latent_dim %d unsupported (max 64)", cfg->latent_dim));
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0)
        return bail(set_err(e, -20, "no HIP device visible: popvae_amd has no CPU fallback"));
    if (cfg->device_id < 0 || cfg->device_id >= ndev) return bail(set_err(e, -2, "device_id %d of %d", cfg->device_id, ndev));
    e->dev = cfg->device_id;
#define CK(x) do { int rc_ = (x); if (rc_) return bail(rc_); } while (0)
#define HK(x) do { hipError_t s_ = (x); if (s_ != hipSuccess) return bail(set_err(e, -10, "%s: %s", #x, hipGetErrorString(s_))); } while (0)
    HK(hipSetDevice(e->dev));
    HK(hipStreamCreateWithFlags(&e->own_stream, hipStreamNonBlocking));
    e->stream = e->own_stream;
    e->L = cfg->n_snps_local;
    e->nblk = (int)((e->L + 31) / 32);
    e->Lp = (int64_t)e->nblk * 32;
    e->W = cfg->width;
    // widths above 128 are cut into groups of 128 units (padded to 256 or 512)
    e->Wp = e->W <= 128 ? pad32(e->W) : (e->W <= 256 ? 256 : 512);
    e->NWB = e->Wp <= 128 ? e->Wp / 32 : 4;
    e->ngrp = e->Wp <= 128 ? 1 : e->Wp / 128;
    e->D = cfg->depth;
    e->K = cfg->latent_dim;
    e->Kh = pad32(2 * e->K);
    e->Kz = pad32(e->K);
    e->adam = PvAdam{cfg->beta1, cfg->beta2, cfg->eps};
    e->lr_host = cfg->lr;

    const int64_t wide_n = (int64_t)e->nblk * e->ngrp * e->NWB * 1024;
    e->tile_floats = (int64_t)e->NWB * 1024;
    e->ts4 = (int)(3 * e->NWB * 256);  // tile stride of the interleaved buffers in float4 units
    CK(alloc_wide(e, e->W1, wide_n, e->tile_floats));
    CK(alloc_wide(e, e->Vout, wide_n, e->tile_floats));
    CK(alloc3(e, e->cout, e->Lp));
    CK(alloc3(e, e->b1, e->Wp));
    e->enc.resize(e->D - 1);
    for (auto& d : e->enc) CK(alloc_dense(e, d, e->Wp, e->Wp, PV_ACT_ELU));
    CK(alloc_dense(e, e->heads, e->Wp, e->Kh, PV_ACT_LINEAR));
    e->dec.resize(e->D);
    CK(alloc_dense(e, e->dec[0], e->Kz, e->Wp, PV_ACT_LINEAR));
    for (int i = 1; i < e->D; ++i) CK(alloc_dense(e, e->dec[i], e->Wp, e->Wp, PV_ACT_ELU));

    // Keras tensor order
    auto& T = e->tensors;
    T.push_back({T_W1, 0, e->L, e->W});
    T.push_back({T_B1, 0, e->W, 0});
    for (int i = 0; i < e->D - 1; ++i) {
        T.push_back({T_NARROW_W, i, e->W, e->W});
        T.push_back({T_NARROW_B, i, e->W, 0});
    }
    T.push_back({T_HEAD_W, 0, e->W, e->K});
    T.push_back({T_HEAD_B, 0, e->K, 0});
    T.push_back({T_HEAD_W, 1, e->W, e->K});
    T.push_back({T_HEAD_B, 1, e->K, 0});
    T.push_back({T_NARROW_W, 1000, e->K, e->W});
    T.push_back({T_NARROW_B, 1000, e->W, 0});
    for (int i = 1; i < e->D; ++i) {
        T.push_back({T_NARROW_W, 1000 + i, e->W, e->W});
        T.push_back({T_NARROW_B, 1000 + i, e->W, 0});
    }
    T.push_back({T_VOUT, 0, e->W, e->L});
    T.push_back({T_COUT, 0, e->L, 0});

    // activations / scratch
    // batches of 33 .. 128 rows in one weight pass, larger ones in chunks of 128 rows (gradients dumped between the chunks):
    // width 128 (four block columns, the fused chain).  Under SNP sharding the two exchanges of such a pass carry 32 T rows
    // (engine-owned communicator; the host-driven staged calls stay at 32 rows per call and ignore T)
    // Batch sizes up to 32 get four tiles as well: their validation batches go through the read-only kernels four at a time
    // (pvae_train_epoch); the training path of such an engine never sees more than one tile.
    if (e->Wp == 128 && 2 * e->D + 2 <= PVN_MAXOPS &&
        !getenv("PVAE_NO_FUSED_NARROW") && !getenv("PVAE_NO_DEC2") && !getenv("PVAE_NO_MULTITILE"))
        e->T = (cfg->batch_size > 32 && cfg->batch_size <= 64) ? 2 : 4;
    const int64_t R = 32 * (int64_t)e->T;  // rows the activation buffers hold
    const int64_t act_n = R * (int64_t)std::max(e->Wp, std::max(e->Kh, e->Kz));
    CK(allocf(e, &e->P, R * e->Wp));
    e->h.resize(e->D);
    e->g.resize(e->D);
    for (auto& p : e->h) CK(allocf(e, &p, R * e->Wp));
    for (auto& p : e->g) CK(allocf(e, &p, R * e->Wp));
    CK(allocf(e, &e->headout, R * e->Kh));
    CK(allocf(e, &e->dheads, R * e->Kh));
    CK(allocf(e, &e->z, R * e->Kz));
    CK(allocf(e, &e->dz, R * e->Kz));
    CK(allocf(e, &e->kl, R));
    CK(allocf(e, &e->eps, R * e->K));
    CK(allocf(e, &e->xch1, R * e->Wp + 32));
    CK(allocf(e, &e->dA, act_n));
    CK(allocf(e, &e->dB, act_n));
    CK(allocf(e, &e->dpre1, R * e->Wp));
    HK(hipMalloc(&e->Xt, e->T * e->Lp * sizeof(uint64_t)));
    HK(hipMemsetAsync(e->Xt, 0, e->T * e->Lp * sizeof(uint64_t), e->stream));
    HK(hipMalloc(&e->Xr, e->T * e->Lp * sizeof(uint64_t)));
    HK(hipMemsetAsync(e->Xr, 0, e->T * e->Lp * sizeof(uint64_t), e->stream));
    HK(hipMalloc(&e->Xrn, e->Lp * sizeof(uint64_t)));
    HK(hipMemsetAsync(e->Xrn, 0, e->Lp * sizeof(uint64_t), e->stream));
    HK(hipMalloc(&e->Xt2, e->Lp * sizeof(uint64_t)));
    HK(hipMemsetAsync(e->Xt2, 0, e->Lp * sizeof(uint64_t), e->stream));
    e->xt_act = e->Xt;

    auto clampi = [](int64_t v, int64_t lo, int64_t hi) { return (int)std::min(hi, std::max(lo, v)); };
    CK(setup_kernel_attrs(e));
    // the wide kernels are persistent-style: the grid is exactly what is co-resident (CUs x blocks per CU), so that
    // every wave streams the same number of subtiles and there is no second, partly filled round
    {
        hipDeviceProp_t prop;
        HK(hipGetDeviceProperties(&prop, e->dev));
        const int cus = prop.multiProcessorCount > 0 ? prop.multiProcessorCount : 256;
        int o_dec = 1, o_encb = 1;
        PV_DISPATCH_NWB(e, {
            if (e->ngrp == 1)
                HK(hipOccupancyMaxActiveBlocksPerMultiprocessor(&o_dec, pvk_dec<N_, 1, PV_MODE_TRAIN>, 256, lds_dec(N_)));
            else if (e->ngrp == 2)
                HK(hipOccupancyMaxActiveBlocksPerMultiprocessor(&o_dec, (pvk_dec_grp<2, 1, PV_MODE_TRAIN, dec_grp_waves(2)>), 64 * dec_grp_waves(2), lds_dec_grp(2)));
            else
                HK(hipOccupancyMaxActiveBlocksPerMultiprocessor(&o_dec, (pvk_dec_grp<4, 1, PV_MODE_TRAIN, dec_grp_waves(4)>), 64 * dec_grp_waves(4), lds_dec_grp(4)));
            HK(hipOccupancyMaxActiveBlocksPerMultiprocessor(&o_encb, (pvk_enc_bwd<N_, PV_MODE_TRAIN, 1>), 256, lds_enc_bwd(e->Wp)));
        });
        auto envi = [](const char* k, int dflt) { const char* v = getenv(k); return v && *v ? atoi(v) : dflt; };
        o_dec = envi("PVAE_DEC_OCC", std::max(o_dec, 1));
        o_encb = envi("PVAE_ENCB_OCC", std::min(std::max(o_encb, 1), 2));  // 2 workgroups per CU measured best (tools/gpu_occ_sweep.sh)
        // Small shapes (and SNP shards) give a wave only a few subtiles: with the full co-resident grid some waves would
        // stream one subtile more than the others and the last round would run on a partly idle chip (3125 blocks over 2048
        // waves: 2 rounds at 76 %).  Up to 8 rounds the grid is shrunk so that every wave gets the same count (1563 waves
        // x 2: dec 87 -> 82 us, enc_bwd 73 -> 70 us at 100 k SNPs); beyond that the imbalance is under 1/8 and the full grid
        // is as fast or faster (measured at 1M SNPs).
        auto balance = [](int64_t units, int64_t slots_max) -> int64_t {
            const int64_t rounds = (units + slots_max - 1) / slots_max;
            if (rounds <= 1 || rounds > 8) return std::min(units, slots_max);
            return (units + rounds - 1) / rounds;
        };
        {   // pvk_enc_bwd: the training pass works in 32 x 32 blocks, wave gw of a unit group on block column gw % NWB of subtile
            // slots gw / NWB, gw / NWB + nslots, ...: the waves of a group, 4 * workgroups, have to be a multiple of NWB.  The
            // whole grid -- sweep workgroups plus the ones that carry the narrow layers' Adam units -- is co-resident.
            const int q = e->NWB == 3 ? 3 : 1;  // workgroups per group in steps of q
            const int adam_wgs = 4 * (2 * e->D + 2);
            int64_t per_group = ((int64_t)e->nblk * e->NWB + 3) / 4;  // one block per wave: more workgroups would idle
            per_group = std::min<int64_t>(per_group, std::max<int64_t>(1, ((int64_t)cus * o_encb - adam_wgs) / e->ngrp));
            per_group = envi("PVAE_ENCB_WGS", (int)(per_group * e->ngrp)) / e->ngrp;
            per_group = std::max<int64_t>(q, (per_group + q - 1) / q * q);
            e->encb_wgs = (int)per_group * e->ngrp;
            e->encb_parts = (int)(per_group * 4 / e->NWB);
            e->enc_wgs = e->encb_wgs;  // the stand-alone forward: same decomposition, bit-identical partial tiles
        }
        // width 128: a workgroup's four waves are the block columns of one subtile slot
        if (e->T > 1) e->encf_mt_wgs = (int)balance(e->nblk, (int64_t)cus * 2);
        if (e->T > 1)  // f32 pipe: two 4-wave workgroups per CU; bf16 pipe: one 8-wave workgroup per CU (it shares the g table)
            e->dece_mt_wgs = PV_MT_BF16 ? (int)((balance(e->nblk, (int64_t)cus * 8) + 7) / 8)
                                        : (int)((balance(e->nblk, (int64_t)cus * 2 * 4) + 3) / 4);
        if (e->ngrp == 1) e->dec_wgs = (int)((balance(e->nblk, (int64_t)cus * o_dec * 4) + 3) / 4);
        else {  // a workgroup takes dec_grp_waves / ngrp SNP blocks per iteration
            const int bpi = dec_grp_waves(e->ngrp) / e->ngrp;
            e->dec_wgs = (int)balance((e->nblk + bpi - 1) / bpi, (int64_t)cus * o_dec);
        }
        e->dec_wgs = clampi(envi("PVAE_DEC_WGS", e->dec_wgs), 1, (e->nblk + 3) / 4);
        if (e->ngrp == 1 && !getenv("PVAE_NO_DEC2")) {
            // pvk_dec2: a workgroup takes BPI = NW / NWB subtiles per iteration; the whole grid is co-resident
            int o2 = 2;
            PV_DISPATCH_NWB(e, HK(hipOccupancyMaxActiveBlocksPerMultiprocessor(&o2, pvk_dec2<N_>, PvDec2<N_>::NW * 64, PvDec2<N_>::lds())));
            o2 = envi("PVAE_DEC2_OCC", std::min(std::max(o2, 1), 2));
            const int bpi = e->NWB == 3 ? 1 : 4 / e->NWB;
            e->dec2_wgs = clampi(envi("PVAE_DEC2_WGS", cus * o2), 1, (e->nblk + bpi - 1) / bpi);
        }
        e->dec_eval_wgs = e->dec_wgs;
        if (e->ngrp == 1) {
            int o_eval = 2;
            PV_DISPATCH_NWB(e, HK(hipOccupancyMaxActiveBlocksPerMultiprocessor(&o_eval, (pvk_dec<N_, 0, PV_MODE_TRAIN>), 256, lds_dec_eval(N_))));
            // 4 blocks per CU measured best (C4: eval step 0.320 ms at 2, 0.279 at 4, 0.284 at 8; tools/gpu_eval_sweep.sh)
            o_eval = envi("PVAE_DECE_OCC", std::min(std::max(o_eval, 1), 4));
            e->dec_eval_wgs = (int)((balance(e->nblk, (int64_t)cus * o_eval * 4) + 3) / 4);
        }
        if (getenv("PVAE_VERBOSE"))
            fprintf(stderr, "[pvae] CUs %d; blocks/CU dec %d enc_bwd %d; grids enc_fwd %d dec %d enc_bwd %d (+ Adam workgroups), dec eval %d\n", cus, o_dec, o_encb,
                    e->enc_wgs, e->dec_wgs, e->encb_wgs, e->dec_eval_wgs);
    }
    e->encpart_floats = (int64_t)e->encb_parts * 32 * e->Wp;  // what a one-tile pass needs (pvae_train_step_noalloc)
    e->encpart_cap = (int64_t)std::max(e->encb_parts, e->encf_mt_wgs) * 32 * e->Wp * e->T;
    CK(allocf(e, &e->encpart, e->encpart_cap));
    if (e->T > 1) {
        hipDeviceProp_t prop2;
        HK(hipGetDeviceProperties(&prop2, e->dev));
        const int cus2 = prop2.multiProcessorCount > 0 ? prop2.multiProcessorCount : 256;
        const int spw = 4 / e->T;
        e->dec3_wgs = std::max(1, std::min(cus2, (e->nblk + spw - 1) / spw));
    }
    e->fuse_fwd = !getenv("PVAE_NO_FUSE_FWD");
    e->dec3b = !getenv("PVAE_NO_DEC3B");
    e->dense16 = !getenv("PVAE_NO_DENSE16");
    e->use_xrn = getenv("PVAE_XRN") != nullptr;
    e->gather_ahead = e->fuse_fwd && !e->use_xrn && getenv("PVAE_GATHER_AHEAD") != nullptr;
    if (e->gather_ahead) {
        HK(hipStreamCreateWithFlags(&e->gstream, hipStreamNonBlocking));
        HK(hipEventCreateWithFlags(&e->ev_pre, hipEventDisableTiming));
        HK(hipEventCreateWithFlags(&e->ev_gather, hipEventDisableTiming));
    }
    e->dgpart_cap = std::max((int64_t)std::max(e->dec_wgs, e->dec2_wgs) * 32 * e->Wp, (int64_t)e->dec3_wgs * 32 * e->T * e->Wp);
    CK(allocf(e, &e->dgpart, e->dgpart_cap));
    // one BCE sum per workgroup of whichever decoder kernel runs; pvk_dec_eval_mt<T, 1> leaves one per workgroup AND row tile
    e->losspart_cap = std::max(std::max(std::max(e->dec_wgs, e->dec2_wgs), std::max(e->dec_eval_wgs, e->dec3_wgs)), e->T * e->dece_mt_wgs);
    CK(allocf(e, &e->losspart, e->losspart_cap));

    HK(hipMalloc(&e->st, sizeof(PvStep)));
    PvStep st0{};
    st0.lr = cfg->lr;
    HK(hipMemcpy(e->st, &st0, sizeof st0, hipMemcpyHostToDevice));
    HK(hipMalloc(&e->d_ntotal, sizeof(int)));
    HK(hipMalloc(&e->d_counter, sizeof(uint64_t)));
    HK(hipMemsetAsync(e->d_counter, 0, sizeof(uint64_t), e->stream));
    CK(ensure_rows(e, 4096));
    CK(ensure_loss(e, 1024));
    for (auto& sl : e->slots) {
        HK(hipHostMalloc((void**)&sl.host, kSlotBytes, hipHostMallocDefault));
        HK(hipEventCreateWithFlags(&sl.ev, hipEventDisableTiming));
    }
    HK(hipHostMalloc((void**)&e->h_out3, 4 * sizeof(float), hipHostMallocDefault));
    CK(scratch_address(e, e->dev, &e->scratch));
    e->fused_narrow = e->Wp <= 128 && 2 * e->D + 2 <= PVN_MAXOPS && !getenv("PVAE_NO_FUSED_NARROW");
    e->narrow_warm = !getenv("PVAE_NO_NARROW_WARM");
    {
        const char* sd = getenv("PVAE_SPLIT_DEC");
        e->split_dec = e->dec2_wgs && e->fused_narrow && sd && atoi(sd) > 0;
        if (e->split_dec) {
            // CU mask of the side stream: R CUs of every XCD stay free.  Bit i of the mask is a CU of XCD i % 8 under the
            // driver's interleaved numbering and of XCD i / 32 under a contiguous one; clearing in word w the bits b with
            // b % 8 == w % 8 (b / 8 < R) reserves R CUs per XCD either way.
            hipDeviceProp_t pr;
            HK(hipGetDeviceProperties(&pr, e->dev));
            const int cus = pr.multiProcessorCount > 0 ? pr.multiProcessorCount : 256, words = (cus + 31) / 32;
            const char* rv = getenv("PVAE_SPLIT_RESERVE");
            const int R = rv ? std::max(0, std::min(4, atoi(rv))) : 4;
            std::vector<uint32_t> mask(words, 0xFFFFFFFFu);
            for (int w = 0; w < words; ++w)
                for (int b = 0; b < 32; ++b)
                    if (b % 8 == w % 8 && b / 8 < R) mask[w] &= ~(1u << b);
            if (cus % 32) mask[words - 1] &= (1u << (cus % 32)) - 1;
            HK(hipExtStreamCreateWithCUMask(&e->side, (uint32_t)words, mask.data()));
            HK(hipEventCreateWithFlags(&e->ev_da, hipEventDisableTiming));
            HK(hipEventCreateWithFlags(&e->ev_upd, hipEventDisableTiming));
        }
    }
    CK(build_bwd_ops(e, false, &e->d_bwd_ops));  // also allocates the operand-order copies the forward table points to
    if (e->fused_narrow) CK(build_fwd_ops(e));
    HK(hipStreamSynchronize(e->stream));
#undef CK
#undef HK
    *out = e;
    return 0;
}

int pvae_comm_destroy(pvae_engine* e);
int pvae_destroy(pvae_engine* e) {
    if (!e) return 0;
    hipSetDevice(e->dev);
    if (e->own_stream) hipStreamSynchronize(e->own_stream);
    if (e->side) { hipStreamSynchronize(e->side); hipStreamDestroy(e->side); }
    if (e->gstream) { hipStreamSynchronize(e->gstream); hipStreamDestroy(e->gstream); }
    if (e->ev_pre) hipEventDestroy(e->ev_pre);
    if (e->ev_gather) hipEventDestroy(e->ev_gather);
    if (e->ev_da) hipEventDestroy(e->ev_da);
    if (e->ev_upd) hipEventDestroy(e->ev_upd);
    pvae_comm_destroy(e);
    if (e->p2p_inbox) hipFree(e->p2p_inbox);
    if (e->p2p_status) hipFree(e->p2p_status);
    for_each_buf(e, [](Buf3& b) { free3(b); });
    for (auto& d : e->enc) { hipFree(d.wf); hipFree(d.wb); }
    for (auto& d : e->dec) { hipFree(d.wf); hipFree(d.wb); }
    hipFree(e->heads.wf); hipFree(e->heads.wb);
    for (auto p : e->h) hipFree(p);
    for (auto p : e->g) hipFree(p);
    float* fl[] = {e->P, e->headout, e->dheads, e->z, e->dz, e->kl, e->eps, e->xch1, e->dA, e->dB, e->dpre1,
                   e->encpart, e->dgpart, e->losspart, e->d_loss};
    for (auto p : fl) hipFree(p);
    hipFree(e->d_fwd_ops); hipFree(e->d_bwd_ops); hipFree(e->d_bwd_ops_grad);
    for (auto p : e->dpre_bufs) hipFree(p);
    hipFree(e->probs);
    hipFree(e->enc_out);
    hipFree(e->Xt); hipFree(e->Xr); hipFree(e->Xrn); hipFree(e->Xt2); hipFree(e->geno); hipFree(e->st); hipFree(e->d_rows); hipFree(e->d_ntotal); hipFree(e->d_counter);
    for (auto& ev : e->evs) { hipEventDestroy(ev.a); hipEventDestroy(ev.b); }
    for (auto& sl : e->slots) { if (sl.host) hipHostFree(sl.host); if (sl.ev) hipEventDestroy(sl.ev); }
    if (e->h_out3) hipHostFree(e->h_out3);
    if (e->own_stream) hipStreamDestroy(e->own_stream);
    delete e;
    return 0;
}

int pvae_set_stream(pvae_engine* e, void* s, int use_own) {
    HIPCHK(hipStreamSynchronize(e->stream));
    e->stream = use_own ? e->own_stream : (hipStream_t)s;  // s == NULL is the legacy default stream
    return 0;
}

int pvae_synchronize(pvae_engine* e) {
    HIPCHK(hipStreamSynchronize(e->stream));
    return 0;
}

int pvae_load_genotypes(pvae_engine* e, const uint8_t* packed, int64_t n_samples, int64_t row_bytes) {
    e->promised_rows.clear();
    if (row_bytes < e->Lp / 4 || row_bytes % 8) return set_err(e, -2, "row_bytes %lld: need a multiple of 8 >= %lld", (long long)row_bytes, (long long)(e->Lp / 4));
    if (n_samples <= 0) return set_err(e, -2, "no samples");
    // reject the reserved "missing" code 3: the reference imputes before training (popvae.py:231-235)
    for (int64_t i = 0; i < n_samples; ++i) {
        const uint8_t* r = packed + i * row_bytes;
        for (int64_t j = 0; j < e->Lp / 4; ++j)
            if (r[j] & (r[j] >> 1) & 0x55) return set_err(e, -2, "genotype code 3 (missing) at sample %lld: impute first", (long long)i);
    }
    HIPCHK(hipStreamSynchronize(e->stream));
    if (e->geno) hipFree(e->geno);
    e->geno = nullptr;
    HIPCHK(hipMalloc(&e->geno, (size_t)(n_samples * row_bytes)));
    HIPCHK(hipMemcpy(e->geno, packed, (size_t)(n_samples * row_bytes), hipMemcpyHostToDevice));
    e->n_samples = n_samples;
    e->row_bytes = row_bytes;
    return 0;
}

int pvae_num_tensors(pvae_engine* e) { return (int)e->tensors.size(); }

int pvae_tensor_shape(pvae_engine* e, int index, int64_t* rows, int64_t* cols) {
    if (index < 0 || index >= (int)e->tensors.size()) return set_err(e, -2, "tensor index %d out of range", index);
    *rows = e->tensors[index].rows;
    *cols = e->tensors[index].cols;
    return 0;
}

int pvae_set_tensor(pvae_engine* e, int index, const float* host, int64_t count) {
    return tensor_io(e, index, 0, const_cast<float*>(host), count, true);
}
int pvae_get_tensor(pvae_engine* e, int index, float* host, int64_t count) { return tensor_io(e, index, 0, host, count, false); }
int pvae_get_moment(pvae_engine* e, int which, int index, float* host, int64_t count) {
    if (which != 0 && which != 1) return set_err(e, -2, "which must be 0 (m) or 1 (v)");
    return tensor_io(e, index, 1 + which, host, count, false);
}
int pvae_get_grad(pvae_engine* e, int index, float* host, int64_t count) { return tensor_io(e, index, 3, host, count, false); }

int pvae_reset_optimizer(pvae_engine* e) {
    HIPCHK(hipStreamSynchronize(e->stream));
    int rc = 0;
    for_each_buf(e, [&](Buf3& b) {
        if (zero_moments(e, b) != hipSuccess) rc = -10;
    });
    if (rc) return set_err(e, rc, "hipMemset failed");
    PvStep st0{};
    st0.lr = e->lr_host;
    st0.bce_probs = e->bce_probs;
    HIPCHK(hipStreamSynchronize(e->stream));
    HIPCHK(hipMemcpy(e->st, &st0, sizeof st0, hipMemcpyHostToDevice));
    return 0;
}

int pvae_init_weights(pvae_engine* e, uint64_t seed) {
    e->narrow_pack_dirty = true;
    e->promised_rows.clear();
    int rc;
    std::vector<float> buf;
    for (int ti = 0; ti < (int)e->tensors.size(); ++ti) {
        const TensorRef& t = e->tensors[ti];
        const int64_t n = t.cols ? t.rows * t.cols : t.rows;
        buf.assign((size_t)n, 0.f);
        if (t.cols) {
            // glorot_uniform with the GLOBAL fans (an SNP shard is a slice of the full kernel)
            const int64_t fr = t.kind == T_W1 ? e->cfg.n_snps_global : t.rows;
            const int64_t fc = t.kind == T_VOUT ? e->cfg.n_snps_global : t.cols;
            const float lim = std::sqrt(6.0f / (float)(fr + fc));
            for (int64_t r = 0; r < t.rows; ++r)
                for (int64_t c = 0; c < t.cols; ++c) {
                    uint64_t gidx;
                    if (t.kind == T_W1) gidx = (uint64_t)(e->cfg.snp_offset + r) * t.cols + c;
                    else if (t.kind == T_VOUT) gidx = (uint64_t)r * e->cfg.n_snps_global + e->cfg.snp_offset + c;
                    else gidx = (uint64_t)r * t.cols + c;
                    buf[(size_t)(r * t.cols + c)] = (2.f * hash_uniform(seed, (uint64_t)ti, gidx) - 1.f) * lim;
                }
        }
        if ((rc = pvae_set_tensor(e, ti, buf.data(), n))) return rc;
    }
    return pvae_reset_optimizer(e);
}

int pvae_set_lr(pvae_engine* e, float lr) {
    e->lr_host = lr;
    HIPCHK(hipMemcpyAsync(&e->st->lr, &e->lr_host, sizeof(float), hipMemcpyHostToDevice, e->stream));
    HIPCHK(hipStreamSynchronize(e->stream));
    return 0;
}
int pvae_set_bce_mode(pvae_engine* e, int mode) {
    if (mode != 0 && mode != 1) return set_err(e, -2, "bce mode %d (0 = from logits, 1 = from clipped probabilities)", mode);
    HIPCHK(hipStreamSynchronize(e->stream));  // the argument is a stack variable, and no step may be half way
    HIPCHK(hipMemcpy(reinterpret_cast<char*>(e->st) + offsetof(PvStep, bce_probs), &mode, sizeof(int), hipMemcpyHostToDevice));
    e->bce_probs = mode;
    return 0;
}

int pvae_get_lr(pvae_engine* e, float* lr) {
    *lr = e->lr_host;
    return 0;
}

int pvae_train_step(pvae_engine* e, const int32_t* rows, int32_t n, const float* eps, float* out3) {
    return full_step(e, rows, n, eps, out3, 0);
}
int pvae_eval_step(pvae_engine* e, const int32_t* rows, int32_t n, const float* eps, float* out3) {
    return full_step(e, rows, n, eps, out3, 1);
}
int pvae_grad_step(pvae_engine* e, const int32_t* rows, int32_t n, const float* eps, float* out3) {
    return full_step(e, rows, n, eps, out3, 2);
}

int pvae_train_step_noalloc(pvae_engine* e, const int32_t* rows, int32_t n, const float* eps, float* out3) {
    // pvae_train_step with the wide kernels' partial sums in the file-scope device array pv_scratch: no hipMalloc,
    // hipHostMalloc, hipFree or event creation on this path (rows / eps / out3 travel through the pinned ring and the
    // staging word pvae_create allocated; every device buffer is the engine's own or pv_scratch).
    if (n <= 0 || n > 32)
        return set_err(e, -2, "pvae_train_step_noalloc takes one device pass of 1..32 rows (got %d); larger batches run as "
                              "tiles with gradient dumps that pvae_train_step allocates on first use", n);
    if (e->cfg.world != 1 || has_comm(e))
        return set_err(e, -5, "pvae_train_step_noalloc: SNP-sharded engines exchange through RCCL, which allocates");
    if (e->timing) return set_err(e, -6, "pvae_train_step_noalloc: switch event timing off first (events are created on demand)");
    auto al = [](size_t floats) { return (floats + 63) / 64 * 64; };  // 256-byte steps
    const size_t n_enc = al((size_t)e->encpart_floats);
    const size_t n_dg = al((size_t)std::max(e->dec_wgs, e->dec2_wgs) * 32 * e->Wp);
    const size_t n_loss = al((size_t)std::max(std::max(e->dec_wgs, e->dec2_wgs), e->dec_eval_wgs));
    const size_t need = (n_enc + n_dg + n_loss) * sizeof(float);
    if (need > PV_SCRATCH_BYTES)
        return set_err(e, -8, "pvae_train_step_noalloc: this model's partial-sum buffers need %zu bytes, the static scratch "
                              "array holds %u", need, (unsigned)PV_SCRATCH_BYTES);
    // nothing in the scratch has to start at zero: every workgroup writes its whole partial before the reduction reads it
    float *s_enc = e->encpart, *s_dg = e->dgpart, *s_loss = e->losspart;
    e->encpart = e->scratch;
    e->dgpart = e->scratch + n_enc;
    e->losspart = e->scratch + n_enc + n_dg;
    const int rc = full_step(e, rows, n, eps, out3, 0);
    e->encpart = s_enc; e->dgpart = s_dg; e->losspart = s_loss;
    return rc;
}

int pvae_train_epoch(pvae_engine* e, const int32_t* perm, int32_t n_train, const int32_t* val_rows, int32_t n_val,
                     float* loss, float* val_loss) {
    e->promised_rows.clear();
    int rc;
    if (e->cfg.world != 1 && !has_comm(e)) return set_err(e, -5, "world > 1: call pvae_comm_init first, or use the staged API");
    if (!e->geno) return set_err(e, -3, "no genotypes loaded (pvae_load_genotypes)");
    if (n_train <= 0) return set_err(e, -2, "empty training set");
    const int B = e->cfg.batch_size;
    const int st_tr = (n_train + B - 1) / B, st_va = n_val > 0 ? (n_val + B - 1) / B : 0;
    if ((rc = check_rows(e, perm, n_train))) return rc;
    if (n_val > 0 && (rc = check_rows(e, val_rows, n_val))) return rc;
    if ((rc = ensure_rows(e, (int64_t)n_train + std::max(n_val, 0)))) return rc;
    if ((rc = ensure_loss(e, (int64_t)(st_tr + st_va) * ((B + 31) / 32)))) return rc;
    HIPCHK(hipMemcpyAsync(e->d_rows, perm, (size_t)n_train * sizeof(int), hipMemcpyHostToDevice, e->stream));
    if (n_val > 0)
        HIPCHK(hipMemcpyAsync(e->d_rows + n_train, val_rows, (size_t)n_val * sizeof(int), hipMemcpyHostToDevice, e->stream));
    bool fwd_done = false;
    const int tp = (B + 31) / 32;  // 32-row tiles per batch
    // batches above 32 rows: every loss slot of the epoch starts at zero (tiles a short batch does not reach stay there) -- one
    // fill per epoch, and the per-step scalars share a launch with the first noise draw of the step (three launch slots per
    // step, ~15 us, were a fill, pvk_step_begin and pvk_gen_eps)
    if (tp > 1) HIPCHK(hipMemsetAsync(e->d_loss, 0, (size_t)3 * tp * (st_tr + st_va) * sizeof(float), e->stream));
    for (int phase = 0; phase < 2 && tp > 1; ++phase) {
        // batches above 32 rows: tiles with accumulated gradients, one Adam update per batch
        const int steps = phase == 0 ? st_tr : st_va, nt = phase == 0 ? n_train : n_val, base = phase == 0 ? 0 : n_train;
        for (int s = 0; s < steps; ++s) {
            const int nb = std::min(B, nt - s * B);
            float* slot = e->d_loss + 3 * (size_t)((phase == 0 ? 0 : st_tr) + s) * tp;
            bool begun = false;  // the step's scalars are written by its first noise launch
            auto begin_eps = [&](int n_eps) {
                if (!begun)
                    hipLaunchKernelGGL(pvk_step_begin_eps, dim3(1), dim3(1024), 0, e->stream, e->st, (const int*)nullptr, nb, B, 0,
                                       phase == 0 ? 1 : 0, e->adam.b1, e->adam.b2, e->eps, n_eps, e->cfg.seed, e->d_counter);
                else
                    hipLaunchKernelGGL(pvk_gen_eps, dim3(1), dim3(1024), 0, e->stream, e->eps, n_eps, e->cfg.seed, e->d_counter);
                begun = true;
            };
            if (phase == 0 && e->T > 1 && nb > 32) {  // the whole batch in one weight pass, or one pass per 32 T rows
                const int R = 32 * e->T, nc = (nb + R - 1) / R;
                for (int c = 0; c < nc; ++c) {
                    begin_eps(R * e->K);
                    if ((rc = launch_check(e, "step_begin/gen_eps"))) return rc;
                    if (nc == 1) rc = multi_tile_train(e, base + s * B, slot);
                    else rc = run_chunk(e, base + s * B + R * c, std::min(R, nb - R * c), nb, c, c == nc - 1, slot + 3 * c);
                    if (rc) return rc;
                }
                continue;
            }
            if (phase == 1 && e->T > 1 && nb > 32 && e->fused_narrow) {  // validation: one read of the wide weights per 32 T rows
                const int R = 32 * e->T, nc = (nb + R - 1) / R;
                for (int c = 0; c < nc; ++c) {
                    begin_eps(R * e->K);
                    if ((rc = launch_check(e, "step_begin/gen_eps"))) return rc;
                    if ((rc = multi_tile_eval(e, base + s * B + R * c, std::min(R, nb - R * c), nb, slot + 3 * c))) return rc;
                }
                continue;
            }
            for (int t = 0; t * 32 < nb; ++t) {
                begin_eps(32 * e->K);
                if ((rc = launch_check(e, "step_begin/gen_eps"))) return rc;
                // a short last batch of at most 32 rows is ONE tile: a plain train step (TRAIN_ACC would add the gradient
                // dump the previous batch's tiles left behind, or find none at all when n_train <= 32 < batch_size)
                const int tm = phase == 1 ? 1 : nb <= 32 ? 0 : (t + 1) * 32 >= nb ? 4 : 2;
                if ((rc = run_tile(e, base + s * B + 32 * t, std::min(32, nb - 32 * t), nb, t, tm, slot + 3 * t))) return rc;
            }
        }
    }
    // batch size 32: the validation batches go through the read-only kernels T at a time, each with its own loss (one read
    // of the wide weights per T batches instead of one per batch)
    const bool mt_val = tp == 1 && B == 32 && e->T > 1 && e->fused_narrow && st_va > 1 && !getenv("PVAE_NO_MT_VAL");
    bool xt_ready = false;  // the current batch was gathered by the previous step's head
    int xt_par = 0;         // which of Xt / Xt2 holds (or will hold) the current batch
    for (int phase = 0; phase < 2 && tp == 1; ++phase) {
        const int steps = phase == 0 ? st_tr : st_va;
        if (!steps) continue;
        if (phase == 1 && mt_val) {
            for (int s0 = 0; s0 < steps; s0 += e->T) {
                const int nbt = std::min(e->T, steps - s0), nrows = std::min(32 * nbt, (int)n_val - 32 * s0);
                for (int t = 0; t < nbt; ++t)  // one draw per batch, as the batch-by-batch path makes them
                    hipLaunchKernelGGL(pvk_gen_eps, dim3(1), dim3(1024), 0, e->stream, e->eps + (size_t)32 * t * e->K, 32 * e->K, e->cfg.seed,
                                       e->d_counter);
                if ((rc = launch_check(e, "gen_eps"))) return rc;
                if ((rc = multi_tile_eval(e, n_train + 32 * s0, nrows, 32, e->d_loss + 3 * (size_t)(st_tr + s0), nbt))) return rc;
            }
            fwd_done = false;
            continue;
        }
        int nt = phase == 0 ? n_train : n_val;
        HIPCHK(hipMemcpyAsync(e->d_ntotal, &nt, sizeof(int), hipMemcpyHostToDevice, e->stream));
        HIPCHK(hipStreamSynchronize(e->stream));  // nt is a stack variable
        const int base = phase == 0 ? 0 : n_train;
        SplitScope split(e, phase == 0 && e->split_dec);
        GatherScope gscope(e);
        for (int s = 0; s < steps; ++s) {
            if ((rc = join_side(e))) return rc;
            // the first-layer forward of the NEXT batch rides on this step's wide encoder backward
            int next_off = 0, next_n = 0;
            bool fuse = false;
            if (phase == 0 && e->fuse_fwd) {
                if (s + 1 < st_tr) { next_off = (s + 1) * B; next_n = std::min(B, n_train - next_off); fuse = true; }
                else if (st_va > 0 && !mt_val) { next_off = n_train; next_n = std::min(B, (int)n_val); fuse = true; }
            }
            // per-step scalars + noise + batch gather (+ the sum of the first-layer partial tiles the previous step's wide
            // encoder backward left behind) in one launch
            // Two transposed-batch buffers in alternation: when the next batch's forward is fused into this step's wide encoder
            // backward, this step's head gathers the NEXT batch (transposed copy + compact words) and the next step's head has
            // nothing left to gather -- one gather per step either way, and the fused forward reads a compact array.
            const int nred = fwd_done ? PV_REDUCE_GRID(32 * e->Wp) : 0, ngb = ((e->nblk + 1) / 2 + 3) / 4;
            uint64_t *xt_cur = xt_par ? e->Xt2 : e->Xt, *xt_next = xt_par ? e->Xt : e->Xt2;
            const bool ahead = fuse && e->gather_ahead;  // ... by a launch of its own on the gather stream
            const bool pre = fuse && (e->use_xrn || ahead);  // the next batch is gathered during this step
            const int ngather = xt_ready ? 0 : ngb, ngather_next = pre && !ahead ? ngb : 0;
            if (ahead) {
                // Xt_next and Xrn were last read by the previous step's kernels: the gather may start once those are done, i.e.
                // behind everything enqueued on `stream` so far -- and then runs beside this step's head and forward chain
                HIPCHK(hipEventRecord(e->ev_pre, e->stream));
                HIPCHK(hipStreamWaitEvent(e->gstream, e->ev_pre, 0));
                hipLaunchKernelGGL(pvk_gather_next, dim3(ngb), dim3(256), 0, e->gstream, (const uint8_t*)e->geno, e->row_bytes,
                                   (const int*)e->d_rows, next_off, next_n, e->nblk, xt_next, e->Xrn);
                HIPCHK(hipEventRecord(e->ev_gather, e->gstream));
                e->gather_pending = true;
            }
            hipLaunchKernelGGL(pvk_step_begin_gather, dim3(nred + ngather + ngather_next + 1), dim3(256), 0, e->stream, e->st,
                               (const int*)e->d_ntotal, B, s, phase == 0 ? 1 : 0, e->adam.b1, e->adam.b2, e->eps, 32 * e->K,
                               e->cfg.seed, e->d_counter, e->geno, e->row_bytes, (const int*)e->d_rows, base + s * B, e->nblk, xt_cur,
                               nred, (const float*)e->encpart, e->encb_parts, 32 * e->Wp, e->P, ngather, ngather_next, next_off,
                               next_n, xt_next, e->Xrn);
            e->xt_act = xt_cur;
            const uint64_t* xrn = pre ? e->Xrn : nullptr;
            xt_ready = pre;
            if (pre) xt_par ^= 1;
            if ((rc = launch_check(e, "step_begin/gen_eps/gather"))) return rc;
            if ((rc = stage_a(e, base + s * B, false, fwd_done, fwd_done))) return rc;
            if ((rc = shard_allreduce(e, 0))) return rc;
            if ((rc = stage_b(e, phase == 0 ? 0 : 1))) return rc;
            if ((rc = shard_allreduce(e, 1))) return rc;
            if ((rc = join_gather(e))) return rc;  // the fused forward of the wide encoder backward reads the next batch's words
            if ((rc = stage_c(e, phase == 0 ? 0 : 1, e->d_loss + 3 * ((phase == 0 ? 0 : st_tr) + s), fuse, next_off, next_n, xrn)))
                return rc;
            fwd_done = fuse;
        }
    }
    std::vector<float> hl((size_t)(st_tr + st_va) * tp * 3);
    HIPCHK(hipMemcpyAsync(hl.data(), e->d_loss, hl.size() * sizeof(float), hipMemcpyDeviceToHost, e->stream));
    HIPCHK(hipStreamSynchronize(e->stream));
    // K9: plain mean of the per-batch scalars (the short last batch counts equally); a batch's scalar is the sum of
    // its tiles' parts
    double a = 0, b = 0;
    for (int s = 0; s < st_tr * tp; ++s) a += hl[(size_t)s * 3];
    for (int s = 0; s < st_va * tp; ++s) b += hl[(size_t)(st_tr * tp + s) * 3];
    if (loss) *loss = (float)(a / st_tr);
    if (val_loss) *val_loss = st_va ? (float)(b / st_va) : NAN;
    return 0;
}

int pvae_stage_encode(pvae_engine* e, int32_t n, float* z_mean, float* z_log_var, float* z) {
    int rc;
    if (n <= 0 || n > 32) return set_err(e, -2, "encode batch of %d rows", n);
    if ((rc = narrow_encoder(e))) return rc;
    std::vector<float> ho((size_t)32 * e->Kh), zz((size_t)32 * e->Kz);
    HIPCHK(hipMemcpyAsync(ho.data(), e->headout, ho.size() * 4, hipMemcpyDeviceToHost, e->stream));
    HIPCHK(hipMemcpyAsync(zz.data(), e->z, zz.size() * 4, hipMemcpyDeviceToHost, e->stream));
    HIPCHK(hipStreamSynchronize(e->stream));
    for (int b = 0; b < n; ++b)
        for (int k = 0; k < e->K; ++k) {
            if (z_mean) z_mean[b * e->K + k] = ho[(size_t)b * e->Kh + k];
            if (z_log_var) z_log_var[b * e->K + k] = ho[(size_t)b * e->Kh + e->K + k];
            if (z) z[b * e->K + k] = zz[(size_t)b * e->Kz + k];
        }
    return 0;
}

int pvae_encode(pvae_engine* e, const int32_t* rows, int32_t n, const float* eps, float* z_mean, float* z_log_var,
                float* z) {
    e->promised_rows.clear();
    int rc;
    if (e->cfg.world != 1 && !has_comm(e))
        return set_err(e, -5, "world > 1: call pvae_comm_init first, or use pvae_stage_a + all-reduce + pvae_stage_encode");
    if (n <= 0) return set_err(e, -2, "encode of %d rows", n);
    // predict batches are independent of the training batch size: 32 rows per pass, or 32 T where the multi-tile forward exists
    // (one read of W1 per 32 T rows)
    const bool mt = e->T > 1 && e->fused_narrow && !getenv("PVAE_NO_MT_ENCODE");
    const int B = mt ? 32 * e->T : 32, nbat = (n + B - 1) / B;
    // the batches are queued back to back; their outputs collect in a device staging buffer and come back in ONE copy
    // (a device-to-host round trip per batch left the GPU idle for longer than a batch takes)
    const size_t per = (size_t)B * (e->Kh + e->Kz);
    if ((int64_t)((size_t)nbat * per) > e->enc_out_cap) {
        HIPCHK(hipStreamSynchronize(e->stream));
        if (e->enc_out) hipFree(e->enc_out);
        e->enc_out = nullptr;
        e->enc_out_cap = 0;
        HIPCHK(hipMalloc(&e->enc_out, (size_t)nbat * per * sizeof(float)));
        e->enc_out_cap = (int64_t)((size_t)nbat * per);
    }
    for (int i = 0; i < nbat; ++i) {
        const int i0 = i * B, nb = std::min(B, n - i0);
        const int save = e->cfg.batch_size;
        e->cfg.batch_size = B;
        rc = begin_batch(e, rows + i0, nb, eps ? eps + (size_t)i0 * e->K : nullptr, false, B, mt);
        e->cfg.batch_size = save;
        if (rc) return rc;
        if (mt) {
            const int nR = B * e->Wp;
            if ((int64_t)e->encf_mt_wgs * nR > e->encpart_cap) return set_err(e, -6, "encode: partial-sum buffer smaller than the grid");
#define PV_ENC_MT(TT)                                                                                                              \
    hipLaunchKernelGGL((pvk_enc_fwd<4, TT, PV_MT_BF16>), dim3(e->encf_mt_wgs), dim3(256), kEncFwdLds, e->stream, e->W1.w, e->geno, e->row_bytes, \
                       e->d_rows, e->st, 0, e->nblk, e->ngrp, e->encpart, e->ts4, (const uint64_t*)nullptr)
            if (e->T == 2) PV_ENC_MT(2);
            else PV_ENC_MT(4);
#undef PV_ENC_MT
            hipLaunchKernelGGL(pvk_reduce_parts, dim3(PV_REDUCE_GRID(nR)), dim3(256), 0, e->stream, e->encpart, e->encf_mt_wgs, nR, 0.5f,
                               e->P, (const float*)nullptr, (float*)nullptr);
            if ((rc = launch_check(e, "enc_fwd"))) return rc;
            if ((rc = shard_allreduce(e, 0, B))) return rc;
            if ((rc = narrow_fused_fwd(e, true, false, e->T))) return rc;
        } else {
            if ((rc = stage_a(e, 0, false))) return rc;
            if ((rc = shard_allreduce(e, 0))) return rc;
            if ((rc = narrow_encoder(e))) return rc;
        }
        float* dst = e->enc_out + (size_t)i * per;
        HIPCHK(hipMemcpyAsync(dst, e->headout, (size_t)B * e->Kh * sizeof(float), hipMemcpyDeviceToDevice, e->stream));
        HIPCHK(hipMemcpyAsync(dst + (size_t)B * e->Kh, e->z, (size_t)B * e->Kz * sizeof(float), hipMemcpyDeviceToDevice, e->stream));
    }
    std::vector<float> host((size_t)nbat * per);
    HIPCHK(hipMemcpyAsync(host.data(), e->enc_out, host.size() * sizeof(float), hipMemcpyDeviceToHost, e->stream));
    HIPCHK(hipStreamSynchronize(e->stream));
    for (int r = 0; r < n; ++r) {
        const float* ho = host.data() + (size_t)(r / B) * per + (size_t)(r % B) * e->Kh;
        const float* zz = host.data() + (size_t)(r / B) * per + (size_t)B * e->Kh + (size_t)(r % B) * e->Kz;
        for (int k = 0; k < e->K; ++k) {
            if (z_mean) z_mean[(size_t)r * e->K + k] = ho[k];
            if (z_log_var) z_log_var[(size_t)r * e->K + k] = ho[e->K + k];
            if (z) z[(size_t)r * e->K + k] = zz[k];
        }
    }
    return 0;
}

int pvae_decode(pvae_engine* e, const float* z, int32_t n, float* probs) {
    int rc;
    if (n < 0 || !z || !probs) return set_err(e, -2, "pvae_decode: bad arguments");
    if (!e->probs) HIPCHK(hipMalloc(&e->probs, (size_t)32 * e->L * sizeof(float)));
    std::vector<float> zp((size_t)32 * e->Kz);
    for (int i0 = 0; i0 < n; i0 += 32) {
        const int nb = std::min(32, n - i0);
        std::fill(zp.begin(), zp.end(), 0.f);
        for (int b = 0; b < nb; ++b)
            for (int k = 0; k < e->K; ++k) zp[(size_t)b * e->Kz + k] = z[(size_t)(i0 + b) * e->K + k];
        HIPCHK(hipStreamSynchronize(e->stream));
        HIPCHK(hipMemcpy(e->z, zp.data(), zp.size() * sizeof(float), hipMemcpyHostToDevice));
        if (e->fused_narrow) {
            if ((rc = narrow_fused_fwd(e, false, true))) return rc;
        } else if ((rc = narrow_decoder(e))) {
            return rc;
        }
        const size_t lds = (size_t)32 * (e->Wp + 1) * 4;
        PV_DISPATCH_NWB(e, {
            HIPCHK(allow_lds(pvk_dec_probs<N_>, lds));
            hipLaunchKernelGGL(pvk_dec_probs<N_>, dim3(std::min(e->nblk / 4 + 1, 1024)), dim3(256), lds, e->stream, e->Vout.w,
                               e->cout.w, e->g[e->D - 1], e->nblk, e->L, e->ngrp, e->ts4, nb, e->probs);
        });
        if ((rc = launch_check(e, "dec_probs"))) return rc;
        HIPCHK(hipMemcpyAsync(probs + (size_t)i0 * e->L, e->probs, (size_t)nb * e->L * sizeof(float), hipMemcpyDeviceToHost,
                              e->stream));
        HIPCHK(hipStreamSynchronize(e->stream));
    }
    return 0;
}

int pvae_pca_gram(pvae_engine* e, int scaler, double* gram, double* site_mean, double* site_std) {
    if (!e->geno) return set_err(e, -3, "no genotypes loaded (pvae_load_genotypes)");
    if (!gram) return set_err(e, -2, "pvae_pca_gram: gram is NULL");
    if (scaler < PV_SCALE_CENTER || scaler > PV_SCALE_STANDARD) return set_err(e, -2, "pvae_pca_gram: scaler %d", scaler);
    const int n = (int)e->n_samples, nt = (n + 127) / 128, n_pairs = nt * (nt + 1) / 2;
    const int64_t row_words = e->row_bytes / 8;
    // SNP-axis slices: enough workgroups to fill the chip a few times over, partials capped at 2 GiB
    int n_slices = std::max(1, std::min(e->nblk, (2048 + n_pairs - 1) / n_pairs));
    n_slices = (int)std::max<int64_t>(1, std::min<int64_t>(n_slices, ((int64_t)2 << 30) / ((int64_t)n_pairs * 128 * 128 * 8)));
    const int per = (e->nblk + n_slices - 1) / n_slices;
    n_slices = (e->nblk + per - 1) / per;
    std::vector<int2> pairs;
    for (int i = 0; i < nt; ++i)
        for (int j = i; j < nt; ++j) pairs.push_back(int2{i, j});
    double2* sc = nullptr;
    double *dmean = nullptr, *dstd = nullptr, *part = nullptr, *G = nullptr;
    int2* dpairs = nullptr;
    int* dbad = nullptr;
    int rc = 0, bad = 0;
    auto body = [&]() -> int {
        HIPCHK(hipMalloc(&sc, (size_t)e->nblk * 32 * sizeof(double2)));
        HIPCHK(hipMalloc(&dmean, (size_t)e->L * sizeof(double)));
        HIPCHK(hipMalloc(&dstd, (size_t)e->L * sizeof(double)));
        HIPCHK(hipMalloc(&part, (size_t)n_slices * n_pairs * 128 * 128 * sizeof(double)));
        HIPCHK(hipMalloc(&G, (size_t)n * n * sizeof(double)));
        HIPCHK(hipMalloc(&dpairs, pairs.size() * sizeof(int2)));
        HIPCHK(hipMalloc(&dbad, sizeof(int)));
        HIPCHK(hipMemsetAsync(dbad, 0, sizeof(int), e->stream));
        HIPCHK(hipMemcpyAsync(dpairs, pairs.data(), pairs.size() * sizeof(int2), hipMemcpyHostToDevice, e->stream));
        hipLaunchKernelGGL(pvk_site_scale, dim3((e->nblk + 15) / 16), dim3(256), 0, e->stream, (const uint64_t*)e->geno, row_words, n,
                           e->nblk, (int64_t)e->L, scaler, sc, dmean, dstd, dbad);
        int r2;
        if ((r2 = launch_check(e, "site_scale"))) return r2;
        HIPCHK(hipMemcpyAsync(&bad, dbad, sizeof(int), hipMemcpyDeviceToHost, e->stream));
        HIPCHK(hipStreamSynchronize(e->stream));
        if (bad)
            return set_err(e, -7, "pvae_pca_gram: %d site(s) have zero scale under this scaler (the reference divides by zero there)", bad);
        hipLaunchKernelGGL(pvk_gram, dim3(n_pairs * n_slices), dim3(256), 0, e->stream, (const uint64_t*)e->geno, row_words, n, e->nblk,
                           (const double2*)sc, (const int2*)dpairs, n_pairs, per, part);
        if ((r2 = launch_check(e, "gram"))) return r2;
        hipLaunchKernelGGL(pvk_gram_reduce, dim3(64, n_pairs), dim3(256), 0, e->stream, (const double*)part, (const int2*)dpairs, n_pairs,
                           n_slices, n, G);
        if ((r2 = launch_check(e, "gram_reduce"))) return r2;
        HIPCHK(hipMemcpyAsync(gram, G, (size_t)n * n * sizeof(double), hipMemcpyDeviceToHost, e->stream));
        if (site_mean) HIPCHK(hipMemcpyAsync(site_mean, dmean, (size_t)e->L * sizeof(double), hipMemcpyDeviceToHost, e->stream));
        if (site_std) HIPCHK(hipMemcpyAsync(site_std, dstd, (size_t)e->L * sizeof(double), hipMemcpyDeviceToHost, e->stream));
        HIPCHK(hipStreamSynchronize(e->stream));
        return 0;
    };
    rc = body();
    hipFree(sc); hipFree(dmean); hipFree(dstd); hipFree(part); hipFree(G); hipFree(dpairs); hipFree(dbad);
    return rc;
}

int pvae_snapshot_best(pvae_engine* e) {
    int rc = 0;
    for_each_buf(e, [&](Buf3& b) {
        if (rc) return;
        if (!b.best && hipMalloc(&b.best, b.n * 4) != hipSuccess) { rc = -10; return; }
        if (copy_params(e, b, b.best, true) != hipSuccess) rc = -10;
    });
    if (rc) return set_err(e, rc, "snapshot failed: %s", hipGetErrorString(hipGetLastError()));
    e->has_best = true;
    return 0;
}

int pvae_restore_best(pvae_engine* e) {
    e->narrow_pack_dirty = true;
    e->promised_rows.clear();
    if (!e->has_best) return set_err(e, -6, "no snapshot taken");
    int rc = 0;
    for_each_buf(e, [&](Buf3& b) {
        if (copy_params(e, b, b.best, false) != hipSuccess) rc = -10;
    });
    if (rc) return set_err(e, rc, "restore failed");
    HIPCHK(hipStreamSynchronize(e->stream));
    return 0;
}

int pvae_exchange_buffer(pvae_engine* e, int which, void** device_ptr, int64_t* n_floats) {
    if (which == 0) { *device_ptr = e->P; *n_floats = 32 * e->Wp; return 0; }
    if (which == 1) { *device_ptr = e->xch1; *n_floats = 32 * e->Wp + 32; return 0; }
    return set_err(e, -2, "exchange buffer %d", which);
}

int pvae_exchange_copy(pvae_engine* e, int which, float* host, int64_t n_floats, int to_device) {
    void* d;
    int64_t n;
    int rc;
    if ((rc = pvae_exchange_buffer(e, which, &d, &n))) return rc;
    if (!host || n_floats != n) return set_err(e, -2, "exchange buffer %d holds %lld floats, caller passed %lld", which, (long long)n, (long long)n_floats);
    if (to_device) HIPCHK(hipMemcpyAsync(d, host, (size_t)n * sizeof(float), hipMemcpyHostToDevice, e->stream));
    else HIPCHK(hipMemcpyAsync(host, d, (size_t)n * sizeof(float), hipMemcpyDeviceToHost, e->stream));
    HIPCHK(hipStreamSynchronize(e->stream));
    return 0;
}

int pvae_stage_a(pvae_engine* e, const int32_t* rows, int32_t n, const float* eps, int mode) {
    int rc;
    if (mode == 3) {
        const int save = e->cfg.batch_size;
        e->cfg.batch_size = 32;
        rc = begin_batch(e, rows, n, eps, false);
        e->cfg.batch_size = save;
    } else {
        rc = begin_batch(e, rows, n, eps, mode == 0);
    }
    if (rc) return rc;
    // the previous pvae_stage_c_next already formed this batch's partial first-layer sums on the updated weights
    const bool fwd_done = !e->promised_rows.empty() && (int32_t)e->promised_rows.size() == n &&
                          !memcmp(e->promised_rows.data(), rows, (size_t)n * sizeof(int32_t));
    e->promised_rows.clear();
    return stage_a(e, 0, mode != 3, fwd_done);
}
int pvae_stage_b(pvae_engine* e, int mode) { return stage_b(e, mode); }
int pvae_stage_c(pvae_engine* e, int mode, float* out3) {
    int rc;
    e->promised_rows.clear();
    if ((rc = ensure_loss(e, 1))) return rc;
    if ((rc = stage_c(e, mode, e->d_loss))) return rc;
    return fetch_out3(e, out3);
}
int pvae_stage_c_next(pvae_engine* e, int mode, float* out3, const int32_t* next_rows, int32_t next_n) {
    int rc;
    e->promised_rows.clear();
    if (mode != 0 || !e->fuse_fwd || !next_rows || next_n <= 0) return pvae_stage_c(e, mode, out3);
    if (next_n > 32) return set_err(e, -2, "a device pass takes at most 32 rows (got %d)", next_n);
    if ((rc = check_rows(e, next_rows, next_n))) return rc;
    if ((rc = ensure_loss(e, 1))) return rc;
    pvae_engine::Slot& sl = e->slots[e->slot_next];
    e->slot_next = (e->slot_next + 1) % 8;
    if (sl.pending) HIPCHK(hipEventSynchronize(sl.ev));
    memcpy(sl.host, next_rows, (size_t)next_n * sizeof(int));
    HIPCHK(hipMemcpyAsync(e->d_rows + 32, sl.host, (size_t)next_n * sizeof(int), hipMemcpyHostToDevice, e->stream));
    HIPCHK(hipEventRecord(sl.ev, e->stream));
    sl.pending = true;
    if ((rc = stage_c(e, 0, e->d_loss, true, 32, next_n))) return rc;
    e->promised_rows.assign(next_rows, next_rows + next_n);
    return fetch_out3(e, out3);
}

int pvae_comm_unique_id(char* id128) {
    PvRccl* r = rccl();
    if (!r->lib || !id128) return -12;
    PvNcclId id;
    if (r->GetUniqueId(&id)) return -12;
    memcpy(id128, id.internal, 128);
    return 0;
}
int pvae_comm_destroy(pvae_engine* e);
int pvae_comm_info(pvae_engine* e, int32_t* nranks, int32_t* rank);
int pvae_comm_init(pvae_engine* e, const char* id128) {
    PvRccl* r = rccl();
    if (!r->lib) return set_err(e, -12, "%s", r->err.c_str());
    if (e->comm) return set_err(e, -12, "communicator already initialised");
    if (e->cfg.world < 1 || e->cfg.rank < 0 || e->cfg.rank >= e->cfg.world) return set_err(e, -2, "rank %d of %d", e->cfg.rank, e->cfg.world);
    HIPCHK(hipSetDevice(e->dev));
    PvNcclId id;
    memcpy(id.internal, id128, 128);
    void* comm = nullptr;
    const int rc = r->CommInitRank(&comm, e->cfg.world, id, e->cfg.rank);
    if (rc) return set_err(e, -12, "ncclCommInitRank: %s", r->GetErrorString(rc));
    e->comm = comm;
    // what RCCL itself saw must be what the configuration says (the struct travels by value through a dlsym'd pointer)
    int32_t n = 0, me = -1;
    if (pvae_comm_info(e, &n, &me)) { pvae_comm_destroy(e); return -12; }
    if (n != e->cfg.world || me != e->cfg.rank) {
        pvae_comm_destroy(e);
        return set_err(e, -12, "communicator reports rank %d of %d, configuration says %d of %d", me, n, e->cfg.rank, e->cfg.world);
    }
    return 0;
}
int pvae_comm_available(void) { return rccl()->lib ? 1 : 0; }
int pvae_comm_info(pvae_engine* e, int32_t* nranks, int32_t* rank) {
    if (!e->comm) return set_err(e, -12, "no communicator (pvae_comm_init)");
    int n = 0, r = -1, rc;
    if ((rc = rccl()->CommCount(e->comm, &n))) return set_err(e, -12, "ncclCommCount: %s", rccl()->GetErrorString(rc));
    if ((rc = rccl()->CommUserRank(e->comm, &r))) return set_err(e, -12, "ncclCommUserRank: %s", rccl()->GetErrorString(rc));
    if (nranks) *nranks = n;
    if (rank) *rank = r;
    return 0;
}
int pvae_comm_allreduce(pvae_engine* e, int which) {
    if (which != 0 && which != 1) return set_err(e, -2, "exchange buffer %d", which);
    return shard_allreduce(e, which);
}
// One-shot exchange among the ranks' processes (opt-in alternative to the RCCL all-reduces; kernel: pvk_p2p_allreduce).
int pvae_p2p_prepare(pvae_engine* e, char* handle64) {
    static_assert(sizeof(hipIpcMemHandle_t) == 64, "the handle travels as 64 bytes");
    if (e->cfg.world < 1 || e->cfg.world > PV_P2P_MAX_WORLD) return set_err(e, -2, "one-shot exchange: world %d (at most %d)", e->cfg.world, PV_P2P_MAX_WORLD);
    HIPCHK(hipSetDevice(e->dev));
    if (const char* sp = getenv("PVAE_P2P_SPIN_LOG2")) {  // how long an exchange waits for a peer's words before it gives up
        const int lg = atoi(sp);
        if (lg < 4 || lg > 30) return set_err(e, -2, "PVAE_P2P_SPIN_LOG2 = %s (4 .. 30)", sp);
        e->p2p_max_spin = 1 << lg;
    }
    if (!e->p2p_inbox) {
        e->p2p_words = (int64_t)32 * e->T * e->Wp + 32;  // the largest exchange: dg of a pass + the loss sums
        const size_t bytes = (size_t)2 * e->cfg.world * e->p2p_words * sizeof(unsigned long long);
        HIPCHK(hipExtMallocWithFlags((void**)&e->p2p_inbox, bytes, hipDeviceMallocFinegrained));
        HIPCHK(hipMemset(e->p2p_inbox, 0, bytes));
        HIPCHK(hipMalloc(&e->p2p_status, sizeof(int)));
        HIPCHK(hipMemset(e->p2p_status, 0, sizeof(int)));
        HIPCHK(hipDeviceSynchronize());
    }
    hipIpcMemHandle_t h;
    HIPCHK(hipIpcGetMemHandle(&h, e->p2p_inbox));
    memcpy(handle64, &h, 64);
    return 0;
}
int pvae_p2p_connect(pvae_engine* e, const char* handles) {
    if (!e->p2p_inbox) return set_err(e, -12, "pvae_p2p_connect before pvae_p2p_prepare");
    if (e->p2p) return set_err(e, -12, "one-shot exchange already connected");
    HIPCHK(hipSetDevice(e->dev));
    for (int r = 0; r < e->cfg.world; ++r) {
        if (r == e->cfg.rank) continue;
        hipIpcMemHandle_t h;
        memcpy(&h, handles + (size_t)64 * r, 64);
        const hipError_t rc = hipIpcOpenMemHandle(&e->p2p_peer[r], h, hipIpcMemLazyEnablePeerAccess);
        if (rc != hipSuccess) {
            for (int q = 0; q < r; ++q)
                if (e->p2p_peer[q]) { hipIpcCloseMemHandle(e->p2p_peer[q]); e->p2p_peer[q] = nullptr; }
            return set_err(e, -12, "hipIpcOpenMemHandle (rank %d's inbox): %s", r, hipGetErrorString(rc));
        }
    }
    e->p2p = true;
    return 0;
}
int pvae_p2p_status(pvae_engine* e, int32_t* gave_up) {
    int st = 0;
    if (e->p2p_status) {
        HIPCHK(hipStreamSynchronize(e->stream));
        HIPCHK(hipMemcpy(&st, e->p2p_status, sizeof(int), hipMemcpyDeviceToHost));
    }
    if (gave_up) *gave_up = st;
    return 0;
}
int pvae_p2p_disconnect(pvae_engine* e) {
    if (e->stream) hipStreamSynchronize(e->stream);
    for (int r = 0; r < PV_P2P_MAX_WORLD; ++r)
        if (e->p2p_peer[r]) { hipIpcCloseMemHandle(e->p2p_peer[r]); e->p2p_peer[r] = nullptr; }
    e->p2p = false;
    // (the inbox stays allocated until pvae_destroy: a peer may still have it mapped)
    return 0;
}

int pvae_comm_destroy(pvae_engine* e) {
    pvae_p2p_disconnect(e);
    if (e->comm) {
        hipStreamSynchronize(e->stream);
        rccl()->CommDestroy(e->comm);
        e->comm = nullptr;
    }
    return 0;
}

int pvae_set_timing(pvae_engine* e, int enabled) {
    flush_timing(e);
    e->timing = enabled != 0;
    e->timing_mask = ~0u;
    for (int i = 0; i < C_NCAT; ++i) { e->cat_ms[i] = 0; e->cat_n[i] = 0; }
    return 0;
}
int pvae_set_timing_only(pvae_engine* e, const char* name) {
    int rc = pvae_set_timing(e, 1);
    if (rc) return rc;
    for (int i = 0; i < C_NCAT; ++i)
        if (!strcmp(name, kCatNames[i])) {
            e->timing_mask = 1u << i;
            return 0;
        }
    e->timing = false;
    return set_err(e, -2, "unknown timing category %s", name);
}
int pvae_get_timing(pvae_engine* e, const char* name, float* ms_total, int64_t* launches) {
    flush_timing(e);
    for (int i = 0; i < C_NCAT; ++i)
        if (!strcmp(name, kCatNames[i])) {
            *ms_total = (float)e->cat_ms[i];
            *launches = e->cat_n[i];
            return 0;
        }
    return set_err(e, -2, "unknown timing category %s", name);
}

}  // extern "C"
