/* subspace_hip.h -- C ABI of libsubspace_hip.so (MI355X / gfx950).
 *
 * Drop-in boundary for the ONE hot path of efmanu/SubspaceInference.jl: subspace construction
 * (SWA mean -> deviation matrix -> tall-skinny Gram/eigen -> projection P) and subspace sampling
 * (W_swa + P*z -> Dense-chain forward -> Gaussian log-likelihood -> random-walk Metropolis).
 *
 * The reference is inline Julia with no FFI seam of its own; each entry point below cites the
 * reference expression it replaces (paths relative to the reference repository root).  A Julia
 * `ccall` wrapper and a Python `ctypes` binding over exactly these symbols are shown in
 * INTEGRATION.md.
 *
 * Conventions
 *   - every function returns int32 status: 0 = SI_OK, negative = error; si_last_error(ctx) gives the
 *     message (owned by the library, valid until the next call on that ctx).  The host wrappers turn a
 *     non-zero status into the reference's error convention, `throw(::String)`.
 *   - host pointers are caller-owned, contiguous, COLUMN-MAJOR (Julia layout); they are only read or
 *     written during the call and never retained.  "_dev" variants take device pointers instead.
 *   - device memory is owned by the opaque si_ctx and released by si_destroy.
 *   - one ctx drives ONE GPU (one process per GPU); a ctx is used by one host thread at a time.  More than one GPU =
 *     one process (and ctx) per GPU joined by an RCCL communicator: the si_comm_* section below.
 *   - there is NO CPU backend: si_create fails when no gfx950 device is usable.
 */
#ifndef SUBSPACE_HIP_H
#define SUBSPACE_HIP_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct si_ctx si_ctx;

enum {
  SI_OK = 0,
  SI_ERR_INVALID = -1,  /* bad argument                                         */
  SI_ERR_STATE = -2,    /* call out of order (e.g. push before begin)           */
  SI_ERR_HIP = -3,      /* HIP runtime error, message has the hipError string   */
  SI_ERR_NOMEM = -4,    /* device or host allocation failed                     */
  SI_ERR_BOUNDS = -5,   /* M > rank(A): the reference's BoundsError at U[:,1:M] */
  SI_ERR_NODEVICE = -6, /* no usable GPU                                        */
  SI_ERR_COMM = -7      /* RCCL error / RCCL not loadable, message has the detail */
};

enum { SI_F32 = 0, SI_F64 = 1 };                 /* dtype of a weight snapshot; compute_dtype of si_infer_setup */
/* Flux 0.11.2 / NNlib 0.7.23 definitions [upstream]: leakyrelu(x) = max(0.01 x, x); elu(x) = x >= 0 ? x : exp(x) - 1;
 * softplus(x) = log(1 + exp(x)); selu(x) = 1.0507009873554805 * (x > 0 ? x : 1.6732632423543772 * (exp(x) - 1)).
 * (gelu / swish are not monotone: their derivative cannot be rebuilt from the stored output, so they are not offered.) */
enum { SI_ACT_IDENTITY = 0, SI_ACT_RELU = 1, SI_ACT_TANH = 2, SI_ACT_SIGMOID = 3, SI_ACT_LEAKYRELU = 4, SI_ACT_ELU = 5,
       SI_ACT_SOFTPLUS = 6, SI_ACT_SELU = 7, SI_ACT_COUNT = 8 };
enum { SI_LAYER_DENSE = 0, SI_LAYER_CONV = 1, SI_LAYER_MAXPOOL = 2, SI_LAYER_FLATTEN = 3 };

/* One layer of a Flux Chain.  Layout contract of the whole path inside the flat weight vector (src/libs.jl:19-22
 * extract_params, src/libs.jl:55-57 Flux.destructure/re -- `model_re` restructures ANY Chain):
 *   Dense  [vec(W) (out x in, column-major); b]          Conv  [vec(weight) (kw x kh x cin x cout, column-major); bias]
 * in the order of the layers; MaxPool and flatten own no parameters.  Activations between layers are (features x B)
 * matrices whose feature index runs in Julia's (W, H, C) column-major order -- the (W, H, C, N) arrays of Flux.
 *   SI_LAYER_DENSE    in, out, act, w_off, b_off                                  (geometry fields unused: 0)
 *   SI_LAYER_CONV     Flux 0.11.2 Conv((kw, kh), cin => cout, act; stride, pad, dilation) with NNlib's default TRUE
 *                     convolution (flipped kernel); in = wi*hi*cin, out = wo*ho*cout,
 *                     wo = (wi + 2*pw - dw*(kw-1) - 1) / sw + 1
 *   SI_LAYER_MAXPOOL  MaxPool((kw, kh); stride = (sw, sh), pad = 0); cin == cout channels
 *   SI_LAYER_FLATTEN  (W, H, C, N) -> (W*H*C, N); in == out                                                          */
typedef struct {
  int32_t kind;  /* SI_LAYER_*      */
  int32_t in;    /* input features  */
  int32_t out;   /* output features */
  int32_t act;   /* SI_ACT_*        */
  int64_t w_off; /* offset of vec(W) / vec(weight) in the flat vector (elements) */
  int64_t b_off; /* offset of the bias                                           */
  int32_t kw, kh;         /* Conv kernel / MaxPool window                         */
  int32_t cin, cout;      /* channels                                             */
  int32_t wi, hi;         /* input width, height                                  */
  int32_t sw, sh;         /* stride                                               */
  int32_t pw, ph;         /* Conv zero padding (each side)                        */
  int32_t dw, dh;         /* Conv dilation                                        */
} si_layer;

/* kernel classes timed by the library's own hipEvents (si_set_profiling) */
enum {
  SI_K_PUSH = 0,     /* K1 swa_dev_push                       */
  SI_K_GRAM = 1,     /* K2 gram A'A partials (MFMA f64)       */
  SI_K_GRAM_RED = 2, /* K2 partial-slab reduction             */
  SI_K_PROJECT = 3,  /* K3 P = A*V_M                          */
  SI_K_RECON = 4,    /* K4 w = W_swa + P*z                    */
  SI_K_DENSE = 5,    /* K5 dense layer GEMM (MFMA f64), all layers */
  SI_K_SSE = 6,      /* K5 sum of squared errors              */
  SI_K_RWMH = 7,     /* K6 propose / accept kernels           */
  SI_K_DENSE_MAIN = 8, /* K5 the largest layer only (dominant kernel) */
  SI_K_EIG_HOST = 9,   /* H1 K x K symmetric eigensolve: HOST wall time, not a device kernel */
  SI_K_BACKWARD = 10,  /* reverse sweep of si_logdensity_grad: delta, dW (split-K MFMA), W'delta, P'g */
  SI_K_CONV = 11,      /* K5 Conv layer: implicit-GEMM convolution (MFMA f64), forward                */
  SI_K_CONV_AUX = 12,  /* K5 Conv stacks: weight re-pack, MaxPool, layout changes (HBM-bound)         */
  SI_K_COUNT = 13
};

typedef struct {
  double ms[SI_K_COUNT];        /* accumulated device time per class (hipEvent), milliseconds */
  int64_t launches[SI_K_COUNT]; /* launches per class                                          */
  double flops[SI_K_COUNT];     /* accumulated ALGORITHMIC flops per class                     */
  double bytes[SI_K_COUNT];     /* accumulated ALGORITHMIC HBM bytes per class                 */
} si_stats;

int32_t si_version(void); /* 500: + narrow-chain fused density / grid loop (specialised at run time), si_train_setup_ex, SI_F32 on Conv chains; 400: + compute_dtype = SI_F32 (300: RCCL communicator, streamed output map, pipelined host push) */

/* ---- context ------------------------------------------------------------------------------- */
int32_t si_create(si_ctx** out, int32_t device_id);
int32_t si_destroy(si_ctx* ctx);
const char* si_last_error(si_ctx* ctx); /* ctx may be NULL: message of the last failed si_create */
/* run on a caller-provided hipStream_t (e.g. torch's current stream); NULL = the ctx's own stream */
int32_t si_set_stream(si_ctx* ctx, void* hip_stream);
int32_t si_synchronize(si_ctx* ctx);
int32_t si_set_profiling(si_ctx* ctx, int32_t on); /* hipEvent pair around every kernel launch */
/* restrict the event pairs to the classes whose bit (1u << SI_K_*) is set (default: all); launches / flops / bytes are
 * counted for every class regardless.  An event pair costs ~5 us of stream time: a timed region that wants the duration
 * of one kernel class only should say so. */
int32_t si_set_profiling_classes(si_ctx* ctx, uint32_t class_mask);
int32_t si_get_stats(si_ctx* ctx, si_stats* out);  /* synchronizes; resolves pending events   */
int32_t si_reset_stats(si_ctx* ctx);
int32_t si_device_name(si_ctx* ctx, char* buf, int32_t buflen);

/* ---- subspace construction: replaces src/subspace_construction.jl:31-33,45-52,61-65 ---------- */
/* :31-33  W_swa = zeros(N) (Float64), A = empty.  K_capacity = number of pushes that will follow.
 * max_cols = 0 keeps every deviation column (the reference's behaviour: the shift at :48-50 is
 * commented out); max_cols = M keeps only the newest M columns (the paper's column shift).        */
int32_t si_construct_begin(si_ctx* ctx, int64_t N, int64_t K_capacity, int32_t max_cols);
/* NON-DEFAULT option: start the running mean at the given weights instead of zeros -- what the
 * reference's docs describe ("Initialize ... W_swa = W_0", docs/src/nn_example.md:44) but its code does not do (:31).
 * Call right after si_construct_begin.                                                                             */
int32_t si_construct_set_mean(si_ctx* ctx, const void* w_host, int32_t w_dtype);
/* NON-DEFAULT option: store the deviation matrix in fp32.  The reference keeps A in Float64
 * (src/subspace_construction.jl:33,51-52); with a_dtype = SI_F32 every column w - W_swa is still FORMED in fp64 and rounded
 * once on the way to memory, while W_swa, the Gram matrix, its eigen-decomposition and P stay fp64.  Halves the memory of A
 * (BASELINE cfg5: 52 -> 26 GB) and the bytes the Gram / projection kernels stream.  Measured against the fp64 oracle
 * (tests/test_gpu_a32.py): W_swa bit-exact, singular values rtol 1e-6, P up to sign 1e-5 of its scale (north_star: 1e-4).
 * Call right after si_construct_begin, before the first push; si_construct_get_A returns the stored (rounded) columns.   */
int32_t si_construct_set_storage(si_ctx* ctx, int32_t a_dtype);
/* :45-52  W = extract_params(ps); n = i/c; W_swa = (n.*W_swa + W)./(n+1); W_dev = W - W_swa;
 * append!(A, W_dev).  `n` is supplied by the caller (it is the EPOCH counter i/c, repeated for every
 * batch of the epoch).  w has N elements of w_dtype.
 * si_construct_push (host snapshot, pageable is fine) is PIPELINED: the call copies w into one of two pinned staging
 * buffers (host threads, SI_HOST_COPY_THREADS), queues H2D + K1 on the ctx's stream and returns -- w_host may be reused
 * at once, the transfer and the kernel overlap the caller's next gradient / update!.  Errors of the queued work surface
 * at the next synchronising call (si_construct_finish, si_synchronize).                                              */
int32_t si_construct_push(si_ctx* ctx, const void* w_host, int32_t w_dtype, double n);
int32_t si_construct_push_dev(si_ctx* ctx, const void* w_dev, int32_t w_dtype, double n);
/* `count` pushes in one pass over snapshots already on the device: snapshot j starts at w_dev + j*ld elements and is
 * pushed with n_host[j].  Bit-identical to `count` calls of si_construct_push_dev, with W_swa held in registers
 * (count*(s_w+8)+16 bytes per element instead of count*(s_w+24)) and one launch instead of `count`.               */
int32_t si_construct_push_batch_dev(si_ctx* ctx, const void* w_dev, int32_t w_dtype, int64_t ld, int32_t count,
                                    const double* n_host);
/* Gram matrix G = A'A (K x K, fp64) of the columns pushed so far, computed on the device.
 * get/set exist so that a row-sharded construction (each rank holds a row block of w, W_swa, A, P) can
 * all-reduce G across ranks (RCCL via the host wrapper) before si_construct_finish.                  */
int32_t si_construct_gram(si_ctx* ctx);
int32_t si_construct_gram_get(si_ctx* ctx, double* G_host /* K*K */, int64_t* K_out);
int32_t si_construct_gram_set(si_ctx* ctx, const double* G_host /* K*K */);
/* The same hook without the host: device address of G (K x K fp64, contiguous, valid until the next si_construct_gram /
 * si_construct_begin).  A row-sharded construction all-reduces it IN PLACE over RCCL (stream-ordered after
 * si_construct_gram when the caller runs the collective on the stream given to si_set_stream) and then calls
 * si_construct_finish: no D2H / H2D of G, no extra synchronisation.                                                   */
int32_t si_construct_gram_ptr(si_ctx* ctx, double** G_dev_out, int64_t* K_out);
/* Ill-conditioned deviation matrices (s_M below ~3e-5 s_1, where A'A no longer resolves the trailing singular values;
 * psvd at src/subspace_construction.jl:63 resolves them down to 5 eps): si_construct_finish switches by itself to a
 * two-stage route -- B = A*V_full on the device, second Gram matrix G2 = B'B, scaled-criterion Jacobi on the host,
 * P = B*W_M.  A ROW-SHARDED construction has to all-reduce G2 as well, so the stage is exposed:
 *     si_construct_gram; all-reduce G;  si_construct_needs_refine(M, &flag)   -- same G => same flag on every rank
 *     if flag:  si_construct_refine  (leaves G2 where G was: si_construct_gram_ptr / _get / _set now address G2);
 *               all-reduce G2;
 *     si_construct_finish(M, ...).
 * After a refine, si_construct_gram_get returns G2 until the next si_construct_gram.                              */
int32_t si_construct_needs_refine(si_ctx* ctx, int32_t M, int32_t* flag_out);
int32_t si_construct_refine(si_ctx* ctx);
/* :61-65  A = reshape(A, N, :); U,s,V = psvd(A); P = U[:,1:M]*Diagonal(s[1:M])  ==  A*V[:,1:M].
 * Outputs may be NULL (results stay on the device for si_infer_setup).  s_out receives the M largest
 * singular values.  Column signs of P are fixed so that the entry of largest magnitude in each column
 * of V is positive (deterministic; the reference's signs are arbitrary).  Returns SI_ERR_BOUNDS when M
 * exceeds the numerical rank of A (s_M <= ~5 eps s_1, psvd's rtol) -- the reference's BoundsError at U[:,1:M]. */
int32_t si_construct_finish(si_ctx* ctx, int32_t M, double* W_swa_out /* N */,
                            double* P_out /* N x M col-major */, double* s_out /* M */,
                            int64_t* K_out);
/* Device addresses of the finished construction's results: W_swa (ld elements, rows [N, ld) zero) and P (ld x M
 * column-major, ld = N rounded up to 64).  Valid until the next si_construct_begin / si_construct_finish / si_destroy.
 * For device-to-device hand-over: an RCCL broadcast of (W_swa, P) to the other ranks' si_infer_setup_dev (independent
 * chains, cfg3), or checks that must not stage 26 GB of P through the host (cfg5).                                  */
int32_t si_construct_result_ptr(si_ctx* ctx, double** W_swa_dev_out, double** P_dev_out, int64_t* ld_out, int32_t* M_out);
/* Copy the results of the FINISHED construction the ctx holds to the host: its own (same values si_construct_finish
 * returned), or one received from another rank (si_bcast_subspace / si_construct_allgather).  Any output may be NULL;
 * N_out / M_out report the sizes (call once with NULL arrays to size them).                                         */
int32_t si_construct_get_result(si_ctx* ctx, double* W_swa_out /* N */, double* P_out /* N x M */, double* s_out /* M */,
                                int64_t* N_out, int32_t* M_out);
/* read back deviation columns [k0, k0+nk) (N x nk col-major) -- parity tests of :51-52 */
int32_t si_construct_get_A(si_ctx* ctx, int64_t k0, int64_t nk, double* A_out);

/* ---- density + sampling: replaces src/space_inference.jl:88-95,111-116,125 ------------------- */
/* :88 split_data (full X, Y), :90-95 density closure.  W_swa / P may both be NULL: the result of the
 * ctx's finished construction is used in place (no host round trip).  X is in_dim x B, Y is out_dim x B,
 * column-major fp64.
 * compute_dtype: SI_F64 = the reference's arithmetic (src/subspace_construction.jl:31,33 and README.md:56-57 make W_swa, P,
 *   the data and therefore the whole density Float64 -- SURVEY section 0, Q6).
 * SI_F32 = the measured fp32 option of SURVEY section 0 Q6 / 8(b),(d): X is rounded to
 *   fp32 once, W_swa + P*z is formed in fp64 and rounded to fp32 once per evaluation, every Dense layer multiplies and
 *   accumulates in fp32 (v_mfma_f32_32x32x2_f32) with fp32 activations; a narrow last layer (out <= 4), its bias /
 *   activation and the sum of squared errors are fp64.  It applies to si_logdensity, si_forward, the RWMH samplers
 *   (si_sample_rwmh*, si_rwmh_*) and -- since round 5 -- si_logdensity_grad (the fp32 forward with kept activations and the
 *   fp32 reverse sweep of the training step; the pull-back P' g and the optional prior term in fp64: grad rtol 2e-5 of its
 *   scale against the fp64 oracle, tests/test_gpu_f32.py; Dense chains only: on a Conv chain set up with SI_F32 the gradient is
 *   refused with SI_ERR_INVALID); si_predict keeps computing in fp64, and the output map
 *   (si_sample_rwmh_weights, si_reconstruct) delivers the fp64 W_swa + P*z.  Stated tolerance (tests/test_gpu_f32.py,
 *   against the fp64 oracle): model outputs 2e-5 of their scale, lp rtol 1e-5 (north_star: 1e-4); at BASELINE cfg2 the
 *   measured lp difference is rtol 2e-8 or better and none of 1000 accept decisions changes.
 *   Conv / MaxPool / flatten chains (round 5): the same option -- the convolution kernels on fp32 operands
 *   (v_mfma_f32_16x16x4_f32), fp32 activations and pooling, the Dense layers behind `flatten` on the fp32 GEMM kernels, the
 *   squared errors in fp64; outputs within 5e-5 of their scale and lp rtol 1e-5 on the CNN cases of the tests (BASELINE cfg4's
 *   CNN, 4096 images: lp rtol 9e-8, no accept decision of 200 differs, 7.3 -> 4.3 ms per transition).                            */
int32_t si_infer_setup(si_ctx* ctx, const si_layer* layers, int32_t L, int64_t N, int32_t M,
                       const double* W_swa, const double* P, const double* X, const double* Y,
                       int32_t in_dim, int32_t out_dim, int64_t B, double sigma_m,
                       int32_t compute_dtype);
/* si_infer_setup with every array already on the DEVICE (fp64, column-major; P_dev has leading dimension ldP >= N).
 * X_dev / Y_dev are copied device-to-device.  W_swa_dev / P_dev: both NULL = the ctx's finished construction;
 * borrow = 0 copies them; borrow = 1 uses them IN PLACE -- the caller keeps them alive and unchanged until the next
 * set-up or si_destroy, bases 16-byte aligned, ldP even and >= N + (N mod 2), W_swa_dev readable for N + (N mod 2)
 * elements (the streaming kernels read rows in 16-byte pairs).  No host staging: at cfg5 P is 26 GB per rank.      */
int32_t si_infer_setup_dev(si_ctx* ctx, const si_layer* layers, int32_t L, int64_t N, int32_t M,
                           const double* W_swa_dev, const double* P_dev, int64_t ldP, int32_t borrow,
                           const double* X_dev, const double* Y_dev, int32_t in_dim, int32_t out_dim, int64_t B,
                           double sigma_m, int32_t compute_dtype);
/* NON-DEFAULT option: add the prior term the reference writes AFTER its `return` (dead code,
 * src/space_inference.jl:95):  + logpdf(MvNormal(zeros(N), sigma_p), new_W).  sigma_p = 0 switches it off again (the
 * reference's behaviour and the default after every si_infer_setup).  Applies to si_logdensity, its gradient and the
 * RWMH samplers.                                                                                                   */
int32_t si_infer_set_prior(si_ctx* ctx, double sigma_p);
/* :90-95  lp[c] = logpdf(MvNormal(vec(f_{W_swa+P z_c}(X)), sigma_m), vec(Y)),  Z is M x C          */
int32_t si_logdensity(si_ctx* ctx, const double* Z, int32_t C, double* lp_out /* C */);
/* lp and its gradient with respect to z: grad_out[m] = d lp / d z_m = (P' * d lp / d w)[m].
 * Replaces `l_pi_grad(theta) = (density(theta), getbackend(backend).gradient(density, theta))`
 * (src/space_inference.jl:107): one reverse sweep through the chain instead of M-wide forward duals.         */
int32_t si_logdensity_grad(si_ctx* ctx, const double* z /* M */, double* lp_out, double* grad_out /* M */);
/* same, additionally returning the model output (out_dim x B) of the LAST z -- forward-pass parity  */
int32_t si_forward(si_ctx* ctx, const double* z /* M */, double* Yhat_out /* out_dim x B */);
/* posterior predictive on NEW inputs: Yhat_out[:, :, c] = f_{W_swa + P*Z[:, c]}(Xnew), out_dim x Bn x C column-major.
 * What the reference's users compute on the host from every returned weight sample (docs/src/nn_example.md:207-217,
 * `re(chn[i])(x)`), without materialising the N-long weight vectors.                                  */
int32_t si_predict(si_ctx* ctx, const double* Z /* M x C */, int32_t C, const double* Xnew /* in_dim x Bn */,
                   int64_t Bn, double* Yhat_out);
/* :111-116  DensityModel + RWMH(MvNormal(zeros(M), sigma_z)) + sample(model, spl, itr): `itr` samples
 * per chain INCLUDING the initial draw z0 ~ proposal; accept iff -randexp() < lp' - lp.  Chains
 * chain_id0 .. chain_id0+nchains-1 use the library's Philox4x32-10 streams (seed, chain, step).
 * Z_out is M x itr x nchains, lp_out is itr x nchains (column-major), accept_rate_out nchains.        */
int32_t si_sample_rwmh(si_ctx* ctx, int64_t itr, double sigma_z, uint64_t seed, int32_t chain_id0,
                       int32_t nchains, double* Z_out, double* lp_out, double* accept_rate_out);
/* si_sample_rwmh plus the reference's output map (:125 `map(z -> W_swa + P*z.params, chm)`), delivered WHILE the chain
 * runs: W_out is N x itr x nchains (column-major; sample t of chain c at W_out + N*(t + itr*c)), may be pageable and
 * untouched.  K4 has already formed W_swa + P z' for every proposal; the library keeps the chains' current weights in a
 * small device ring (select on accept, no second K4 pass over P), a second stream DMAs sample t into pinned staging
 * during transition t+1 and host threads (SI_HOST_COPY_THREADS) move it into W_out a few transitions later -- at cfg2
 * 8.4 MB per sample against 3.2 ms of compute.  Bit-identical to si_reconstruct on the returned Z.                   */
int32_t si_sample_rwmh_weights(si_ctx* ctx, int64_t itr, double sigma_z, uint64_t seed, int32_t chain_id0,
                               int32_t nchains, double* Z_out, double* lp_out, double* accept_rate_out, double* W_out);
/* The same chain, one transition at a time, with the proposal's sum of squared errors handed to the caller between
 * evaluation and acceptance: a DATA-SHARDED density (every rank holds a column block of X, Y and the same W_swa, P)
 * all-reduces the per-rank partial SSE there (one 8*nchains-byte RCCL all-reduce per step).  d_total = out_dim * B over
 * ALL ranks (0 = this ctx's own B).  All ranks use the same seed / chain ids, hence take identical decisions.        */
int32_t si_rwmh_begin(si_ctx* ctx, int64_t itr, double sigma_z, uint64_t seed, int32_t chain_id0, int32_t nchains,
                      int64_t d_total);
int32_t si_rwmh_step_eval(si_ctx* ctx, double* sse_local_out /* nchains; NULL = leave them on the device */);
int32_t si_rwmh_step_accept(si_ctx* ctx, const double* sse_total /* nchains; NULL = the device buffer holds them */);
/* device address of the nchains partial sums between step_eval(NULL) and step_accept(NULL): the caller all-reduces it
 * in place over RCCL on the stream given to si_set_stream -- no host round trip, no synchronisation per transition.  */
int32_t si_rwmh_sse_ptr(si_ctx* ctx, double** sse_dev_out, int32_t* nchains_out);
int32_t si_rwmh_end(si_ctx* ctx, double* Z_out, double* lp_out, double* accept_rate_out);
/* drop an open session.  While a session is open, si_logdensity / _grad / si_forward / si_predict / si_sample_rwmh
 * return SI_ERR_STATE (they share the session's proposal and SSE buffers); si_rwmh_begin restarts it.              */
int32_t si_rwmh_abort(si_ctx* ctx);
/* K6 as a DEVICE-RESIDENT loop and K5 as ONE launch for narrow Dense chains -- the sizes the reference itself
 * documents.  Automatic (on = 1, the default):
 *   - weights, data and activations of a chain fit one workgroup's LDS and 2*N*B <= 3 MFLOP (README.md:52-79): si_sample_rwmh
 *     runs all `itr` transitions of all chains in ONE launch, one workgroup per chain;
 *   - fp64 Dense chains with hidden widths <= 256 (docs/src/nn_example.md:112-118, 2-200-50-50-50-1 on 1000 observations): the
 *     density runs EVERY layer in one launch, a workgroup per batch tile with the activations in LDS; si_sample_rwmh runs all
 *     transitions in one PERSISTENT launch when ceil(B / tile) * nchains workgroups are resident together (one per CU), with
 *     two bounded grid barriers per transition -- a barrier that times out (the GPU is shared and the workgroups were not
 *     all resident) ends the call with SI_ERR_HIP, never a hang; more chains than that stack in the grid of the one-launch
 *     density, one pass of launches per transition.
 * Every form gives the SAME BITS as the launch-per-step loop with one launch per layer (on = 0: what the parity tests
 * compare against).  on = 2: the one-launch density, but no device-resident loop.
 * RUN-TIME SPECIALISATION (round 5): for chains with a narrow head whose weight fragments fit a wave's registers (the
 * nn_example model does) the two narrow-chain kernels are compiled once per distinct chain and process by hiprtc from kernel text
 * embedded in the library, with the layer table as compile-time constants (csrc/chain_spec.inc: same arithmetic, same bits;
 * ~1.5 s the first time a chain shape is used; the loop then takes ONE grid barrier per transition).  When hiprtc is not usable
 * the generic kernels run; si_chain_kernel_info says which did.  on = 3 / 4: as 1 / 2 with the generic kernels only.       */
int32_t si_set_chain_loop(si_ctx* ctx, int32_t on);
/* did the last density evaluation / the last si_sample_rwmh* call run the kernels specialised at run time (1) or the generic
 * ones (0)?  si_chain_spec_message: why not (hiprtc's log, or the class limit), "" when they did.                           */
int32_t si_chain_kernel_info(si_ctx* ctx, int32_t* density_specialised, int32_t* loop_specialised);
const char* si_chain_spec_message(si_ctx* ctx);
/* :91 / :125  W_out[:, c] = W_swa + P * Z[:, c]   (N x C col-major).  Pipelined: K4 -> pinned staging (second stream) ->
 * host threads (SI_HOST_COPY_THREADS, default min(8, cpus / 2)) copy into W_out, which may be pageable and untouched.  */
int32_t si_reconstruct(si_ctx* ctx, const double* Z /* M x C */, int64_t C, double* W_out);

/* ---- on-device training step for Dense chains with the mse cost ----------------------------------
 * Replaces the caller-side body of the reference's loop, src/subspace_construction.jl:39-43
 *     gs = gradient(ps) do training_loss = cost(model, d...) end;  Flux.update!(opt, ps, gs)
 * for cost = mse(m(x), y) and opt in {Descent (0), Momentum (1), ADAM (2)} with Flux 0.11.2 semantics (Float32 weights
 * and optimiser state, Float64 arithmetic).  p1 = rho / beta1, p2 = beta2.  The whole (X, Y) is uploaded once; a step
 * takes the observation indices of its batch (the DataLoader's permutation stays with the caller).                  */
int32_t si_train_setup(si_ctx* ctx, const si_layer* layers, int32_t L, int64_t N, const float* w0 /* N */,
                       const double* X, const double* Y, int32_t in_dim, int32_t out_dim, int64_t B_total,
                       int64_t batch_max, int32_t opt_kind, double eta, double p1, double p2);
/* The same with the data in the caller's element type and the arithmetic of the step chosen by it -- WHICH PRECISION A STEP
 * COMPUTES IN.  In the reference the step is `gradient(ps) do cost(model, x, y) end` on Flux arrays (:39-43): with a Float32
 * model (Flux's default) and Float64 data (`rand(10, 100)`, every example of the reference) Julia promotes and the pass is
 * Float64; with Float32 data (ordinary Flux practice) the whole pass -- sgemm forward and reverse, the loss -- is Float32.
 *   data_dtype     SI_F32 / SI_F64: element type of X and Y as the caller holds them
 *   compute_dtype  SI_DTYPE_OF_DATA (the reference's own arithmetic for that data), or SI_F32 / SI_F64 to override
 *                  (Float64 data + SI_F32 rounds X once; Float32 data + SI_F64 widens it: what si_train_setup's callers
 *                  had to do before round 5, wider than the reference and twice the matrix time)
 * SI_F32 (Dense chains; Conv chains: SI_ERR_INVALID, they train in SI_F64): fp32 operands and activations on
 * v_mfma_f32_32x32x2_f32 forward and reverse; the loss, the narrow head's partial sums and every sum over the batch (dW, db)
 * in fp64, rounded once into the Float32 gradient; the optimiser as Flux runs it (Float64 scalars on Float32 arrays, rounded on
 * the store).  Not bit-equal to a BLAS sgemm pass (the sums are blocked differently): measured against the oracle's Float32
 * path (tests/test_gpu_train_f32.py) the weights after 12 ADAM steps agree to 2e-6 of their scale, the loss to 1e-6.
 * si_train_grad_ptr / _get / _set keep exchanging N doubles (holding the Float32 values) in either mode.                  */
enum { SI_DTYPE_OF_DATA = -1 };
int32_t si_train_setup_ex(si_ctx* ctx, const si_layer* layers, int32_t L, int64_t N, const float* w0 /* N */, const void* X,
                          const void* Y, int32_t data_dtype, int32_t in_dim, int32_t out_dim, int64_t B_total, int64_t batch_max,
                          int32_t opt_kind, double eta, double p1, double p2, int32_t compute_dtype);
int32_t si_train_compute_dtype(si_ctx* ctx, int32_t* out /* SI_F32 / SI_F64: what the steps of this set-up compute in */);
/* With loss_out = NULL the call QUEUES the step and returns: idx (caller-owned, may be reused at once) is copied into a pinned
 * staging buffer and shipped asynchronously, a batch that is 0 .. nb-1 in order is used in place; nothing waits for the
 * previous step, so the caller's work between two steps overlaps the GPU's.  Errors of queued work surface at the next
 * synchronising call (a step with loss_out, si_train_get_weights, si_construct_finish, si_synchronize).              */
int32_t si_train_step(si_ctx* ctx, const int64_t* idx /* nb observation indices, 0-based */, int64_t nb,
                      double* loss_out /* mse of the batch before the update; may be NULL (no host sync) */);
/* :45-52 with W taken in place from the device-resident Float32 weights (no extract_params, no PCIe) */
int32_t si_train_push(si_ctx* ctx, double n);
int32_t si_train_get_weights(si_ctx* ctx, float* w_out /* N */);
/* the optimiser state after the device steps, so that the caller's `opt` can be left as Flux.update! leaves it (its
 * IdDict state persists across calls in the reference): m_out = Momentum velocity / ADAM first moment, v_out = ADAM
 * second moment (both Float32, flat layout of the weights; either may be NULL), beta_pows_out[2] = ADAM's running
 * beta1^t, beta2^t.                                                                                                 */
int32_t si_train_get_opt_state(si_ctx* ctx, float* m_out /* N */, float* v_out /* N */, double* beta_pows_out /* 2 */);
/* Data-parallel form of si_train_step (one gradient all-reduce per step).  Every rank holds the same
 * weights / optimiser state and its own nb of the nb_total observations of the batch:
 *   si_train_grad   forward + reverse sweep; leaves  d mse(whole batch)/dw  restricted to this rank's observations
 *                   (i.e. scaled by 1/(out_dim*nb_total)) in a device buffer and returns the local SSE;
 *   the caller sums that buffer over the ranks IN PLACE: RCCL all-reduce on the device pointer of si_train_grad_ptr
 *                   (N doubles; valid until the next si_train_setup), or si_train_grad_get / _set through the host;
 *   si_train_apply  Flux.update!(opt, ps, gs) (:43) from the summed gradient.
 * With nb_total == nb and no exchange, grad + apply is exactly si_train_step.                                       */
int32_t si_train_grad(si_ctx* ctx, const int64_t* idx, int64_t nb, int64_t nb_total, double* sse_local_out /* may be NULL */);
int32_t si_train_grad_ptr(si_ctx* ctx, double** grad_dev_out, int64_t* n_out);
int32_t si_train_grad_get(si_ctx* ctx, double* g_out /* N */);
int32_t si_train_grad_set(si_ctx* ctx, const double* g_in /* N */);
int32_t si_train_apply(si_ctx* ctx);

/* ---- R1: multi-GPU behind the C ABI ------------------------------------------------------------
 * One process per GPU, one si_ctx per process, one RCCL communicator per ctx (RCCL over xGMI; bound with dlopen at the
 * first si_comm_* call -- a host that already carries librccl.so.1, e.g. PyTorch, shares its copy).  The reference is
 * single-process; what these collectives parallelise:  src/subspace_construction.jl:45-52,63 (row-sharded SWA /
 * deviation / Gram / projection), src/space_inference.jl:94 (log-likelihood over column blocks of the data),
 * src/subspace_construction.jl:39-43 (data-parallel training step), and independent chains (no per-step exchange).
 * Every collective is issued on the ctx's stream, in place on buffers the library owns: no host staging and no
 * synchronisation of its own.  All ranks must make the same calls in the same order (RCCL semantics).
 * Failure behaviour: the entry points that prepare something LOCALLY before their collective (si_bcast_subspace,
 * si_construct_allgather, si_sample_rwmh_sharded, si_train_step_dp) first agree on the ranks' local status with one
 * 8-byte all-reduce (max): when any rank failed its preparation, NO rank issues the data collective -- the failing rank
 * returns its own error, the others SI_ERR_COMM.  The thin in-place collectives (si_construct_allreduce_gram,
 * si_rwmh_allreduce_sse, si_train_allreduce_grad, si_comm_*_host) check only local state that is the same on every rank
 * of an SPMD caller (a communicator exists, si_construct_gram / si_rwmh_step_eval / si_train_grad ran); a rank that
 * returns SI_ERR_STATE from one of them while the others are inside it leaves those blocked -- keep the call sequences
 * identical, as with RCCL itself.
 *
 * Start-up: rank 0 calls si_comm_unique_id and ships the 128 bytes to the other ranks by whatever the host has
 * (Julia: Distributed.remotecall / a file; Python: a file or any process group); every rank then calls
 * si_comm_init_rank.                                                                                               */
#define SI_COMM_ID_BYTES 128
enum { SI_COMM_SUM = 0, SI_COMM_MAX = 1 };
int32_t si_comm_unique_id(uint8_t* id_out /* SI_COMM_ID_BYTES */);   /* needs no ctx; errors: si_last_error(NULL) */
int32_t si_comm_init_rank(si_ctx* ctx, int32_t world, int32_t rank, const uint8_t* id /* SI_COMM_ID_BYTES */);
int32_t si_comm_destroy(si_ctx* ctx);                                 /* also done by si_destroy */
/* world = 0 when the ctx has no communicator; rccl_version = ncclGetVersion (0 when RCCL could not be loaded) */
int32_t si_comm_info(si_ctx* ctx, int32_t* world_out, int32_t* rank_out, int32_t* rccl_version_out);
/* small HOST values over the communicator (timings, losses, rank counts): n <= 4096, op = SI_COMM_SUM / SI_COMM_MAX;
 * synchronous.  si_comm_barrier returns when every rank's stream has reached it.                                    */
int32_t si_comm_allreduce_host(si_ctx* ctx, double* inout, int64_t n, int32_t op);
int32_t si_comm_allgather_host(si_ctx* ctx, const double* send /* n */, int64_t n, double* recv /* n x world */);
int32_t si_comm_barrier(si_ctx* ctx);
/* The row partition every sharded entry point assumes: rank's rows [r0, r1) of n_total, boundaries on multiples of 32
 * elements (256 B) so that each shard keeps the kernels' 16-byte alignment.  Pure arithmetic (no ctx, no GPU).      */
int32_t si_row_shard(int64_t n_total, int32_t rank, int32_t world, int64_t* r0_out, int64_t* r1_out);
/* Row-sharded construction: each rank pushes ITS rows of every snapshot (si_construct_begin with N = r1 - r0), then
 *     si_construct_gram;  si_construct_allreduce_gram;                          -- G = sum of the local A'A, K x K fp64
 *     si_construct_needs_refine -> if set: si_construct_refine; si_construct_allreduce_gram   (ill-conditioned A)
 *     si_construct_finish            -- replicated K x K eigensolve, this rank's rows of P
 *     si_construct_allgather(n_total) (optional) -- the FULL (W_swa, P) on every rank, device to device; the ctx then
 *                                       holds a finished construction of n_total rows (si_infer_setup(NULL, NULL)).   */
int32_t si_construct_allreduce_gram(si_ctx* ctx);
int32_t si_construct_allgather(si_ctx* ctx, int64_t n_total);
/* Independent chains (BASELINE cfg3): (W_swa, P, s) of the construction finished on `root` -> every rank's ctx, device
 * to device, once (168 MB at cfg2); receivers then hold a finished construction.  No exchange per transition: rank r
 * runs chains with its own chain ids (si_sample_rwmh), results are gathered with si_comm_allgather_host.            */
int32_t si_bcast_subspace(si_ctx* ctx, int32_t root, int64_t N, int32_t M);
/* Data-sharded density (BASELINE cfg5): each rank's si_infer_setup holds a column block of (X, Y) and the full
 * W_swa / P; all ranks use the same seed / chain ids.  Between si_rwmh_step_eval(NULL) and si_rwmh_step_accept(NULL):
 * all-reduce of the nchains partial sums of squared errors (8 * nchains bytes).  si_sample_rwmh_sharded is the whole
 * loop in one call (eval -> all-reduce -> accept per transition on one stream, no host round trip); d_total =
 * out_dim * observations over ALL ranks.                                                                            */
int32_t si_rwmh_allreduce_sse(si_ctx* ctx);
int32_t si_sample_rwmh_sharded(si_ctx* ctx, int64_t itr, double sigma_z, uint64_t seed, int32_t chain_id0, int32_t nchains,
                               int64_t d_total, double* Z_out, double* lp_out, double* accept_rate_out);
/* Data-parallel training step: after si_train_grad on this rank's share of the batch, sum the N-double gradient and
 * the local SSE over the ranks (one grouped launch); sse_total_out may be NULL (then no host synchronisation).
 * si_train_step_dp = si_train_grad + si_train_allreduce_grad + si_train_apply; loss_out = mse of the WHOLE batch.   */
int32_t si_train_allreduce_grad(si_ctx* ctx, double* sse_total_out);
int32_t si_train_step_dp(si_ctx* ctx, const int64_t* idx, int64_t nb, int64_t nb_total, double* loss_out);

/* ---- host utility (no GPU needed): the K x K symmetric eigensolver used inside si_construct_finish.
 * a: n x n symmetric column-major, overwritten by the eigenvectors (columns); w: eigenvalues ascending. */
int si_host_sym_eig(int n, double* a, double* w);
/* The route si_construct_finish takes first: only the m largest eigenpairs (Householder reduction with the reflectors
 * kept in factored form, eigenvalues by vector-free QL, eigenvectors by inverse iteration, result verified against g).
 * g: n x n symmetric column-major, left intact; w_top: m eigenvalues DESCENDING; V: n x m eigenvectors.
 * Returns 0 = verified result, 1 = declined / failed verification (si_construct_finish then uses si_host_sym_eig). */
int si_host_sym_eig_top(int n, const double* g, int m, double* w_top, double* V);
/* The second-stage solver of the ill-conditioned route: cyclic two-sided Jacobi with the scaled stopping criterion
 * |a_pq| <= eps*sqrt(a_pp*a_qq) for a symmetric positive semi-definite matrix (eigenvalues accurate relative to
 * themselves for graded matrices).  a: n x n column-major, destroyed; w: eigenvalues DESCENDING; v: eigenvectors.   */
int si_host_jacobi_eig_psd(int n, double* a, double* w, double* v);
/* The host copy pool (si_construct_push, si_reconstruct, si_sample_rwmh_weights) sizes itself from the CPUs the process may
 * really use: the affinity mask capped by the cgroup CPU quota (si_host_cpu_budget; a container that shows 256 CPUs and
 * grants 16 gets 16).  Threads per process = SI_HOST_COPY_THREADS if set, else min(8, budget / (2 * nproc)) with nproc = the
 * world of the ctx's communicator (si_comm_init_rank shrinks the pool: 8 ranks on a 16-CPU quota copy with one thread each).
 * si_host_parse_cpu_max: "<quota> <period>" of a cgroup-v2 cpu.max file -> CPUs granted (0 = unlimited / unparsable).    */
int si_host_cpu_budget(void);
double si_host_parse_cpu_max(const char* text);
int si_host_copy_plan(int budget, int nproc, const char* env /* value of SI_HOST_COPY_THREADS or NULL */);

#ifdef __cplusplus
}
#endif
#endif /* SUBSPACE_HIP_H */
