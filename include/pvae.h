/* pvae.h -- C ABI of libpvae.so: the MI355X (gfx950) training engine behind popvae.py.
 *
 * The reference (/root/reference/popvae.py) has no plugin or FFI surface: its hot path is a sequence of
 * direct Keras calls.  Each entry point below replaces one of those call sites (cited per function);
 * INTEGRATION.md shows the ctypes binding a maintainer of popvae.py would add.
 *
 * Conventions: one engine handle per GPU rank, single host thread per handle; every function returns 0
 * on success or a negative code (pvae_last_error gives the text); all host buffers are caller-owned and
 * copied in/out; no callbacks.  Tensors cross the boundary in Keras layout (Dense kernel = [in][out]
 * row-major float32) and Keras order: encoder [W1,b1,...,W_D,b_D,Wmu,bmu,Ws,bs], decoder
 * [V1,c1,...,V_D,c_D,Vout,cout]  (= vae.get_weights()).  Tensors with an SNP axis hold only this rank's
 * SNP shard [snp_offset, snp_offset + n_snps_local).
 *
 * There is no CPU fallback: without a usable HIP device pvae_create fails.
 */
#ifndef PVAE_H
#define PVAE_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct pvae_engine pvae_engine;

typedef struct pvae_config {
    int64_t n_snps_local;  /* SNPs held by this rank */
    int64_t n_snps_global; /* SNPs of the whole model (== local when world == 1) */
    int64_t snp_offset;    /* first global SNP of this rank's shard */
    int32_t width;         /* --width  (popvae.py:78-80) */
    int32_t depth;         /* --depth  (popvae.py:75-77) */
    int32_t latent_dim;    /* --latent_dim (popvae.py:104-106) */
    int32_t batch_size;    /* --batch_size (popvae.py:26); above 32 rows at a padded width of exactly 128 (widths 97..128): one pass over the wide weights per 128 rows; otherwise (widths <= 96, 256, 512) 32-row tiles; gradients accumulated, one Adam update either way */
    float lr, beta1, beta2, eps; /* Keras Adam defaults 1e-3, .9, .999, 1e-7 (popvae.py:426) */
    int32_t device_id;
    int32_t rank, world;   /* SNP-axis shard id / count */
    uint64_t seed;         /* reparameterisation noise stream */
} pvae_config;

/* library-level */
const char* pvae_version(void);
int pvae_device_count(void);

/* model build: encoder/decoder/vae + compile("adam")  -- popvae.py:392-426 */
int pvae_create(const pvae_config* cfg, pvae_engine** out);
/* K.clear_session() -- popvae.py:369 */
int pvae_destroy(pvae_engine* e);
const char* pvae_last_error(pvae_engine* e);
/* run all engine work on this hipStream_t (e.g. torch's current stream, so that a torch.distributed all-reduce
 * on the exchange buffers is ordered with the stages); hip_stream == NULL is the legacy default stream.
 * use_own != 0 returns to the engine's own non-blocking stream (the state after pvae_create). */
int pvae_set_stream(pvae_engine* e, void* hip_stream, int use_own);
int pvae_synchronize(pvae_engine* e);

/* genotype matrix: traingen/testgen/dc (popvae.py:237-238,259-267) as ONE resident 2-bit matrix of all N
 * samples, this rank's SNP columns only; SNP l of a row at bits 2*(l%4) of byte l/4, code = 2*dc value. */
int pvae_load_genotypes(pvae_engine* e, const uint8_t* packed, int64_t n_samples, int64_t row_bytes);

/* tensors in Keras order -- vae.get_weights()/set_weights(), load_weights (popvae.py:508) */
int pvae_num_tensors(pvae_engine* e);
int pvae_tensor_shape(pvae_engine* e, int index, int64_t* rows, int64_t* cols); /* bias: rows = n, cols = 0 */
int pvae_set_tensor(pvae_engine* e, int index, const float* host, int64_t count);
int pvae_get_tensor(pvae_engine* e, int index, float* host, int64_t count);
/* glorot_uniform kernels + zero biases (Keras Dense defaults, popvae.py:394-409); resets Adam state.
 * The draw for an SNP-axis tensor depends only on (seed, global element index): shards agree. */
int pvae_init_weights(pvae_engine* e, uint64_t seed);
/* Adam slot variables (m, v) and iteration count; which: 0 = m, 1 = v */
int pvae_get_moment(pvae_engine* e, int which, int index, float* host, int64_t count);
int pvae_reset_optimizer(pvae_engine* e);

/* optimizer.lr (ReduceLROnPlateau, popvae.py:441-450) */
int pvae_set_lr(pvae_engine* e, float lr);
int pvae_get_lr(pvae_engine* e, float* lr);
/* K5: which form of keras.losses.binary_crossentropy (popvae.py:417-418) the fused decoder pass computes:
 * 0 (default) from the logits -- what Keras 2.15 does for a Sigmoid output in graph mode -- or 1 from probabilities clipped to
 * [1e-7, 1 - 1e-7] (its other path; differs where |logit| > ~16: no gradient beyond the clip).  No counterpart flag in the
 * reference: which path Keras takes is not visible there, so both are offered and tested against the oracle. */
int pvae_set_bce_mode(pvae_engine* e, int mode);

/* one train_step of vae.fit (popvae.py:490): rows = sample indices into the resident matrix (n <= batch_size),
 * eps = injected N(0,1) [n][latent_dim] or NULL for the engine's own stream.
 * out3 = {loss, mean reconstruction, mean KL} or NULL (no host sync when NULL). */
int pvae_train_step(pvae_engine* e, const int32_t* rows, int32_t n, const float* eps, float* out3);
/* No counterpart in the reference: pvae_train_step for callers that cannot allocate (tight serving loops, harnesses that
 * forbid hipMalloc; under HIP graph capture only with out3 == NULL and with rows / eps re-uploaded outside the graph: the call
 * copies them into a pinned slot on the host at call time and, with out3 != NULL, synchronises the stream).  Same arguments,
 * same result bit for bit; the wide kernels'
 * split-K / split-N partial sums live in a file-scope device array of the library (32 MiB per GPU, held until the
 * process ends) instead of hipMalloc'ed buffers, and the call performs no hipMalloc / hipHostMalloc / hipFree / event
 * creation.  Limits: 1 <= n <= 32 rows (one device pass), world == 1, event timing off, and the model's partial-sum
 * buffers must fit the array (every width up to 512 on a 256-CU device does; otherwise error -8).  All calls on one GPU
 * share that GPU's array: calls that overlap on one GPU (two engines on two host threads) race. */
int pvae_train_step_noalloc(pvae_engine* e, const int32_t* rows, int32_t n, const float* eps, float* out3);
/* one test_step (validation; forward + loss, no update) */
int pvae_eval_step(pvae_engine* e, const int32_t* rows, int32_t n, const float* eps, float* out3);
/* same forward/backward as pvae_train_step but gradients are stored instead of applied (test hook for
 * per-step gradient parity); read them back with pvae_get_grad in Keras layout. */
int pvae_grad_step(pvae_engine* e, const int32_t* rows, int32_t n, const float* eps, float* out3);
int pvae_get_grad(pvae_engine* e, int index, float* host, int64_t count);

/* one epoch of vae.fit(shuffle=True, validation_data=...) (popvae.py:490-498): perm = this epoch's order of
 * the training rows (K8), batches of batch_size with the short last batch kept, then the validation rows in
 * order.  loss/val_loss = unweighted means over batches (K9). */
int pvae_train_epoch(pvae_engine* e, const int32_t* perm, int32_t n_train, const int32_t* val_rows, int32_t n_val,
                     float* loss, float* val_loss);

/* encoder.predict (popvae.py:455,509): z_mean/z_log_var/z for the given rows; each [n][latent_dim]; any
 * output may be NULL.  eps as in pvae_train_step ([n][latent_dim] or NULL). */
int pvae_encode(pvae_engine* e, const int32_t* rows, int32_t n, const float* eps, float* z_mean, float* z_log_var,
                float* z);
/* decoder.predict (scripts/popvae_decoder_tests_HGDP.py:156): probs [n][n_snps_local] from z [n][latent_dim] */
int pvae_decode(pvae_engine* e, const float* z, int32_t n, float* probs);

/* popvae.py:528-533: allel.pca(dc.T, scaler, n_components) = per-site scaler + SVD of the scaled N x L matrix.
 * This entry computes, in float64 on the GPU, the per-site scaler over ALL loaded samples and the N x N Gram matrix
 * G = Xs Xs^T of the scaled matrix (v_mfma_f64_16x16x4_f64 on the resident 2-bit matrix); the host takes
 * G = U S^2 U^T and keeps U S (popvae_amd/pca.py).  scaler: 0 = centre only, 1 = Patterson, 2 = standard
 * (allel get_scaler; the reference feeds VALUES code/2 with allel's default ploidy 2, reproduced as is).
 * gram: host, n_samples^2 doubles, row-major.  site_mean / site_std: host, n_snps_local doubles, or NULL.
 * Under SNP-axis sharding each rank gets the Gram matrix of its own columns; their sum is the global one. */
int pvae_pca_gram(pvae_engine* e, int scaler, double* gram, double* site_mean, double* site_std);

/* ModelCheckpoint(save_best_only) + load_weights (popvae.py:429-435,508): device-to-device copy of the
 * parameters (not the Adam slots) */
int pvae_snapshot_best(pvae_engine* e);
int pvae_restore_best(pvae_engine* e);

/* SNP-axis sharding (world > 1): the step is cut at its two exchange points so that the host can run the
 * all-reduce (RCCL through torch.distributed) on the exchange buffers between the stages.
 *   stage A: gather/transposes the batch, partial first-layer sums      -> exchange buffer 0 [32][width_padded]
 *   stage B: narrow forward, fused wide decoder                        -> exchange buffer 1 [32*width_padded + 32]
 *   stage C: loss, narrow backward (+Adam), wide encoder backward (+Adam)
 * mode: 0 = train, 1 = eval (stage C only finalises the loss), 2 = gradients only, 3 = encode (A then
 * pvae_stage_encode).  With world == 1 pvae_train_step is exactly A,B,C back to back. */
int pvae_exchange_buffer(pvae_engine* e, int which, void** device_ptr, int64_t* n_floats);
/* host-staged exchange for transports that cannot take device pointers (MPI without GPU support, gloo, tests):
 * copies exchange buffer `which` to the host (to_device == 0) or back (to_device != 0), ordered with the stages on
 * the engine stream; n_floats must equal the buffer's size.  No counterpart in the reference (it has no multi-GPU path). */
int pvae_exchange_copy(pvae_engine* e, int which, float* host, int64_t n_floats, int to_device);
int pvae_stage_a(pvae_engine* e, const int32_t* rows, int32_t n, const float* eps, int mode);
int pvae_stage_b(pvae_engine* e, int mode);
int pvae_stage_c(pvae_engine* e, int mode, float* out3);
/* stage C of a TRAIN step that also knows the next batch: the wide encoder backward, which has every updated first-layer
 * tile in registers anyway, forms the next batch's partial first-layer sums on the way (as pvae_train_epoch does between
 * its steps), so that the next pvae_stage_a -- if it is called with exactly these rows -- only gathers and reduces and
 * the first-layer kernel is read once per step instead of twice.  Any other call in between drops the promise (the
 * next stage A then runs the ordinary forward).  Other modes, or next_rows == NULL: same as pvae_stage_c. */
int pvae_stage_c_next(pvae_engine* e, int mode, float* out3, const int32_t* next_rows, int32_t next_n);
int pvae_stage_encode(pvae_engine* e, int32_t n, float* z_mean, float* z_log_var, float* z);

/* SNP-axis sharding without host code in the step: the engine's own RCCL communicator.  One rank calls
 * pvae_comm_unique_id (ncclGetUniqueId, 128 bytes), the host distributes the bytes (any transport: torch.distributed
 * object broadcast, MPI_Bcast, a file), every rank then calls pvae_comm_init (ncclCommInitRank with the rank / world of
 * pvae_config; collective).  From then on pvae_train_step / eval_step / grad_step / train_epoch / encode work under
 * world > 1 exactly as on one GPU: the two SUM all-reduces of a step are enqueued on the engine's stream between the
 * stages (in place on the exchange buffers; next-batch forward fused as on one GPU).  librccl is bound with dlopen at the
 * first call, so a process that already holds one (PyTorch's) keeps using that one.  pvae_comm_allreduce runs one such
 * all-reduce by itself (self-test; host-driven staging).  No counterpart in the reference (it has no multi-GPU path). */
int pvae_comm_available(void); /* 1 when librccl could be bound (checked by every rank BEFORE the collective pvae_comm_init) */
int pvae_comm_unique_id(char* id128);
int pvae_comm_init(pvae_engine* e, const char* id128);
/* what the communicator itself reports (ncclCommCount / ncclCommUserRank): bench.py prints it on the --gpus N line */
int pvae_comm_info(pvae_engine* e, int32_t* nranks, int32_t* rank);
int pvae_comm_allreduce(pvae_engine* e, int which);
int pvae_comm_destroy(pvae_engine* e);

/* One-shot exchange among the ranks' processes instead of the two RCCL all-reduces of a step (SURVEY 8e "tuned path"; opt-in:
 * PVAE_COMM=p2p in popvae_amd/dist.py; no counterpart in the reference, which is single-GPU).  Every rank owns an inbox in
 * fine-grained device memory; pvae_p2p_prepare allocates it and returns its 64-byte hipIpc handle, the host carries the handles
 * of all ranks to every rank (rank order, 64 bytes each), pvae_p2p_connect maps the peers' inboxes; from then on the ordinary
 * calls (pvae_train_epoch, pvae_train_step, pvae_encode, pvae_comm_allreduce ...) exchange through one kernel per exchange point
 * (pvk_p2p_allreduce: 8-byte {sequence, value} words written into every rank's inbox, summed in rank order).  Waits are bounded:
 * an exchange that gives up (2^PVAE_P2P_SPIN_LOG2 polls of a word, default 2^26, about a minute) fills the buffer with NaN and
 * raises the status word pvae_p2p_status reads -- the host checks it once per epoch and per encode, agrees on it across the
 * ranks and ends the run (popvae_amd/dist.py check_exchange).  At most 16 ranks.  Before pvae_destroy frees an inbox every
 * rank must have left its last exchange (the host barriers first).  Rehearsed with two processes on one GPU
 * (tests/test_gpu_dist.py); between devices it has not run yet. */
int pvae_p2p_prepare(pvae_engine* e, char* handle64);
int pvae_p2p_connect(pvae_engine* e, const char* handles /* world x 64 bytes */);
int pvae_p2p_status(pvae_engine* e, int32_t* gave_up);
int pvae_p2p_disconnect(pvae_engine* e);

/* measurement hook: average device time (ms) of the last call's kernels by name, via hipEvents on the
 * engine stream; names: "enc_fwd", "dec", "enc_bwd", "narrow_fwd", "narrow_bwd", "gather", and "xch0" / "xch1" = the two
 * all-reduces of a step on the engine-owned communicator */
int pvae_set_timing(pvae_engine* e, int enabled);
int pvae_get_timing(pvae_engine* e, const char* name, float* ms_total, int64_t* launches);
/* as pvae_set_timing(e, 1), but only the named category gets events: every event pair costs the stream ~5-10 us per
 * launch, which is why a run that measures throughput times its dominant kernel only */
int pvae_set_timing_only(pvae_engine* e, const char* name);

#ifdef __cplusplus
}
#endif
#endif
