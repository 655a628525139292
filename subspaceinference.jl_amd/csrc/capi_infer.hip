// C ABI, inference side: si_infer_setup*, the density (si_logdensity, si_forward, si_predict) and its gradient
// (si_logdensity_grad) -- reference src/space_inference.jl:88-95,107 and src/libs.jl:55-57,75-77.  Host-side orchestration only:
// every arithmetic step runs in the kernels of kernels_*.hip.  No CPU fallback anywhere in this file.
#include "capi_common.h"
#include "chain_spec_rtc.h"

using namespace si;

extern "C" {

// =================================================================================================
// density + sampling
// =================================================================================================
// ---- narrow Dense chains: every layer in one launch (kernels_chain_grid.hip) ---------------------------------------
// The class: fp64 Dense chains with the four MFMA-epilogue activations, hidden widths <= 256 (the weights of a layer stream
// from L2 per workgroup: wide layers belong on the big-tile kernel, which shares W between 128 observations), an LDS plan that
// fits at 16 observations per workgroup, and one squared error per thread in the SSE kernels (the order the fused loop
// reproduces).  docs/src/nn_example.md:112-118 is the model this is for.
static constexpr int SI_FUSED_MAX_WIDTH = 256;
static bool fused_chain_class(const si_ctx* ctx) {
  if (ctx->f32 || ctx->plan.has_conv) return false;
  const int L = (int)ctx->layers.size();
  if (L < 1 || L > SI_CHAIN_MAX_LAYERS) return false;
  for (int l = 0; l + 1 < L; ++l)
    if (ctx->layers[(size_t)l].out > SI_FUSED_MAX_WIDTH) return false;
  if (!ctx->fuse_tail && ctx->layers[(size_t)L - 1].out > SI_FUSED_MAX_WIDTH) return false;
  if ((int64_t)ctx->out_dim * ctx->B > (int64_t)256 * ctx->sse_blocks) return false;
  ChainFusedPlan fp;
  return chain_fused_plan(fp, ctx->layers.data(), L, ctx->B, 1, ctx->fuse_tail,
                          ctx->fuse_tail ? dense_fused_slot_feats(ctx->layers[(size_t)L - 2].out) : 0, ctx->fuse_slots) != 0;
}
// batch tile of the stacked launch (16 NB observations per workgroup; the 16-feature tiles of a layer dealt over its four
// waves).  Measured on docs/src/nn_example.md's model at 512 chains (profiles/r05_chain_grid_knockouts.log): 32 observations
// per workgroup 730 us, 16 per workgroup 780 us, one WAVE per 16-observation tile without any barrier (launch_chain_fused's
// wave_tiles form, kept for the harness) 1430 us -- a single wave's stream of small dependent steps leaves the SIMD idle.
void fused_fill_program(const si_ctx* ctx, ChainFusedPlan& fp) {
  fp.prog = ctx->d_cgprog;
  for (int i = 0; i < 5; ++i) {
    fp.prog_start[i] = ctx->cg_start[i];
    fp.prog_count[i] = ctx->cg_count[i];
    fp.prog_chunks[i] = ctx->cg_chunks[i];
  }
}
static size_t fused_plan_for(const si_ctx* ctx, int nchains, ChainFusedPlan& fp, int* nb_out, bool* wave_tiles) {
  fused_fill_program(ctx, fp);
  const int L = (int)ctx->layers.size();
  const int sf = ctx->fuse_tail ? dense_fused_slot_feats(ctx->layers[(size_t)L - 2].out) : 0;
  *wave_tiles = false;
  for (int nb : {2, 1}) {
    const size_t lds = chain_fused_plan(fp, ctx->layers.data(), L, ctx->B, nb, ctx->fuse_tail, sf, ctx->fuse_slots);
    const int64_t wgs = (ctx->B + 16 * nb - 1) / (16 * nb) * nchains;
    if (lds != 0 && (nb == 1 || (lds <= (size_t)80 * 1024 && wgs >= (int64_t)2 * ctx->num_cu))) {
      *nb_out = nb;
      return lds;
    }
  }
  return 0;
}

// forward workspace for `slots` chains evaluated in one launch (grid.y = chain slot)
static bool alloc_forward(si_ctx* ctx, int slots) {
  dev_free(ctx->d_w); dev_free(ctx->d_act[0]); dev_free(ctx->d_act[1]); dev_free(ctx->d_ssepart); dev_free(ctx->d_part);
  dev_free(ctx->d_yhat); dev_free(ctx->d_w32); dev_free(ctx->d_act32[0]); dev_free(ctx->d_act32[1]);
  ctx->fw_slots = 0;
  const size_t S = (size_t)slots, dB = (size_t)ctx->out_dim * (size_t)ctx->B;
  // SI_F32: fp32 weights + fp32 ping-pong activations INSTEAD of the fp64 activations (the fp64 weights stay: K4 writes
  // both, the output map / prior / gradient read them); the head partials serve both paths (the larger slot count)
  const size_t pslots = (size_t)std::max(ctx->fuse_slots, ctx->fuse_slots32);
  if (dev_alloc(&ctx->d_w, S * (size_t)pad_ld(ctx->iN)) != hipSuccess ||
      (!ctx->f32 && (dev_alloc(&ctx->d_act[0], S * (size_t)ctx->act_elems) != hipSuccess ||
                     dev_alloc(&ctx->d_act[1], S * (size_t)ctx->act_elems) != hipSuccess)) ||
      (ctx->f32 && (dev_alloc(&ctx->d_w32, S * (size_t)pad_ld(ctx->iN)) != hipSuccess ||
                    dev_alloc(&ctx->d_act32[0], S * (size_t)ctx->act_elems) != hipSuccess ||
                    dev_alloc(&ctx->d_act32[1], S * (size_t)ctx->act_elems) != hipSuccess)) ||
      dev_alloc(&ctx->d_ssepart, S * (size_t)ctx->sse_blocks) != hipSuccess ||
      (ctx->fuse_tail && dev_alloc(&ctx->d_part, S * pslots * dB) != hipSuccess) ||
      ((ctx->fuse_tail || ctx->f32 || ctx->fused_ok) && dev_alloc(&ctx->d_yhat, S * dB) != hipSuccess))
    return false;
  dev_free(ctx->d_wsqpart);
  ctx->wsq_blocks = sse_num_blocks(ctx->iN, ctx->num_cu);
  if (dev_alloc(&ctx->d_wsqpart, S * (size_t)ctx->wsq_blocks) != hipSuccess) return false;
  ctx->fw_slots = slots;
  return true;
}

// where the arrays of an inference set-up live: host (si_infer_setup), device copied (si_infer_setup_dev, borrow = 0),
// device used in place (borrow = 1)
enum SetupSrc { SRC_HOST = 0, SRC_DEV_COPY = 1, SRC_DEV_BORROW = 2 };

static int32_t infer_setup_common(si_ctx* ctx, const si_layer* layers, int32_t L, int64_t N, int32_t M, const double* W_swa,
                                  const double* P, int64_t ldP_in, const double* X, const double* Y, int32_t in_dim,
                                  int32_t out_dim, int64_t B, double sigma_m, int32_t compute_dtype, SetupSrc src) {
  CHECK_CTX(ctx);
  if (!layers || L <= 0 || N <= 0 || M <= 0 || !X || !Y || in_dim <= 0 || out_dim <= 0 || B <= 0)
    return fail(ctx, SI_ERR_INVALID, "si_infer_setup: bad argument");
  if (!(sigma_m > 0.0)) return fail(ctx, SI_ERR_INVALID, "si_infer_setup: sigma_m must be positive");
  if (compute_dtype != SI_F64 && compute_dtype != SI_F32)
    return fail(ctx, SI_ERR_INVALID, "si_infer_setup: compute_dtype must be SI_F64 (the reference's arithmetic) or SI_F32");
  if ((W_swa == nullptr) != (P == nullptr))
    return fail(ctx, SI_ERR_INVALID, "si_infer_setup: W_swa and P must both be given or both be NULL");
  // the Chain: Dense / Conv / MaxPool / flatten layers (anything else: the reference's "model_re function is not
  // available for this model", libs.jl:59)
  NetPlan plan;
  {
    const int32_t prc = net_plan(ctx, "si_infer_setup", layers, L, N, in_dim, out_dim, plan);
    if (prc != SI_OK) return prc;
  }
  int main_layer = 0;
  double main_flops = -1.0;
  for (int l = 0; l < L; ++l) {
    const LayerPlan& q = plan.L[(size_t)l];
    const double fl = q.kind == SI_LAYER_DENSE ? 2.0 * q.in_feat * (double)q.out_feat
                      : q.kind == SI_LAYER_CONV ? 2.0 * q.KW * q.KH * q.C * (double)q.Co * q.Wo * q.Ho : 0.0;
    if (fl > main_flops) {
      main_flops = fl;
      main_layer = l;
    }
  }
  BIND(ctx);
  SI_HIP(ctx, hipStreamSynchronize(ctx->stream));
  free_infer(ctx);
  if (!W_swa) {
    if (!ctx->c_finished) return fail(ctx, SI_ERR_STATE, "si_infer_setup: no finished construction to take W_swa / P from");
    if (ctx->N != N || ctx->M_built != M)
      return fail(ctx, SI_ERR_INVALID, "si_infer_setup: N / M differ from the finished construction");
    ctx->i_swa = ctx->d_swa;
    ctx->i_P = ctx->d_P;
    ctx->ldP = ctx->ldA;
  } else {
    if (src == SRC_DEV_BORROW) {
      // used in place: the caller keeps both buffers alive (and unchanged) until the next set-up / si_destroy
      ctx->ldP = ldP_in;
      ctx->i_swa = W_swa;
      ctx->i_P = P;
    } else {
      const hipMemcpyKind kind = src == SRC_HOST ? hipMemcpyHostToDevice : hipMemcpyDeviceToDevice;
      ctx->ldP = pad_ld(N);
      if (dev_alloc(&ctx->d_iswa, (size_t)ctx->ldP) != hipSuccess ||
          dev_alloc(&ctx->d_iP, (size_t)ctx->ldP * M) != hipSuccess) {
        free_infer(ctx);
        return fail(ctx, SI_ERR_NOMEM, "si_infer_setup: allocation of W_swa / P failed");
      }
      SI_HIP(ctx, hipMemsetAsync(ctx->d_iswa, 0, (size_t)ctx->ldP * sizeof(double), ctx->stream));
      SI_HIP(ctx, hipMemsetAsync(ctx->d_iP, 0, (size_t)ctx->ldP * M * sizeof(double), ctx->stream));
      SI_HIP(ctx, hipMemcpyAsync(ctx->d_iswa, W_swa, (size_t)N * sizeof(double), kind, ctx->stream));
      SI_HIP(ctx, hipMemcpy2DAsync(ctx->d_iP, (size_t)ctx->ldP * sizeof(double), P, (size_t)ldP_in * sizeof(double),
                                   (size_t)N * sizeof(double), (size_t)M, kind, ctx->stream));
      ctx->i_swa = ctx->d_iswa;
      ctx->i_P = ctx->d_iP;
    }
  }
  ctx->layers.assign(layers, layers + L);
  ctx->iN = N;
  ctx->iM = M;
  ctx->in_dim = in_dim;
  ctx->out_dim = out_dim;
  ctx->B = B;
  ctx->sigma_m = sigma_m;
  ctx->main_layer = main_layer;
  // fused tail: a narrow last layer (regression heads: out = 1) is folded into the epilogue of the layer before it
  ctx->plan = plan;
  ctx->fuse_tail = !plan.has_conv && (L >= 2) && layers[L - 1].out <= SI_FUSE_MAX_OUT &&
                   layers[L - 1].act < SI_ACT_LEAKYRELU && layers[L - 2].act < SI_ACT_LEAKYRELU;   // (kernels_gemm.h)
  ctx->fuse_slots = ctx->fuse_tail ? dense_fused_slots(layers[L - 2].out) : 0;
  ctx->f32 = compute_dtype == SI_F32;
  // (the fp32 weight vector is 256-byte aligned and its slots are pad_ld(N) apart: a layer's W is 16-byte aligned iff w_off % 4 == 0)
  ctx->fuse_slots32 = (ctx->f32 && ctx->fuse_tail) ? dense_f32_fused_slots(layers[L - 2].out, layers[L - 2].in, layers[L - 2].w_off % 4 == 0) : 0;
  int64_t maxstored = 1;
  for (int l = 0; l < (ctx->fuse_tail ? L - 2 : L); ++l) maxstored = std::max<int64_t>(maxstored, plan.L[(size_t)l].out_elems);
  ctx->max_stored = maxstored;
  ctx->act_elems = pad_ld(maxstored * B);
  ctx->sse_blocks = sse_num_blocks((int64_t)out_dim * B, ctx->num_cu);
  ctx->fused_ok = fused_chain_class(ctx);
  if (ctx->fused_ok) {   // the chain's tile program (64 bytes per 16-feature tile), uploaded once
    std::vector<CgTileD> prog;
    chain_fused_program(ctx->layers.data(), L, ctx->fuse_tail, prog, ctx->cg_start, ctx->cg_count, ctx->cg_chunks);
    if (dev_alloc(&ctx->d_cgprog, prog.size()) != hipSuccess) {
      free_infer(ctx);
      return fail(ctx, SI_ERR_NOMEM, "si_infer_setup: device allocation failed");
    }
    SI_HIP(ctx, hipMemcpy(ctx->d_cgprog, prog.data(), prog.size() * sizeof(CgTileD), hipMemcpyHostToDevice));
  }
  if (dev_alloc(&ctx->d_X, (size_t)in_dim * B) != hipSuccess || dev_alloc(&ctx->d_Y, (size_t)out_dim * B) != hipSuccess ||
      (ctx->f32 && dev_alloc(&ctx->d_X32, (size_t)pad_ld(std::max<int64_t>((int64_t)in_dim * B, plan.input_spatial ? plan.in_elems * B : 0))) != hipSuccess) ||
      (ctx->f32 && plan.has_conv && dev_alloc(&ctx->d_wpack32, plan.wpack_elems) != hipSuccess) ||
      !alloc_forward(ctx, 1) ||
      (plan.has_conv && dev_alloc(&ctx->d_wpack, plan.wpack_elems) != hipSuccess) ||
      (plan.input_spatial && dev_alloc(&ctx->d_Xc, (size_t)plan.in_elems * B) != hipSuccess)) {
    free_infer(ctx);
    return fail(ctx, SI_ERR_NOMEM, "si_infer_setup: device allocation failed");
  }
  {
    const hipMemcpyKind kind = src == SRC_HOST ? hipMemcpyHostToDevice : hipMemcpyDeviceToDevice;
    SI_HIP(ctx, hipMemcpyAsync(ctx->d_X, X, (size_t)in_dim * B * sizeof(double), kind, ctx->stream));
    SI_HIP(ctx, hipMemcpyAsync(ctx->d_Y, Y, (size_t)out_dim * B * sizeof(double), kind, ctx->stream));
  }
  if (plan.input_spatial) net_input(ctx, plan, ctx->d_X, ctx->d_Xc, B);  // (W, H, C, N) -> channel-fastest, once
  if (ctx->f32 && plan.input_spatial)   // X rounded to fp32 once, in the layout the conv kernels read (pad channels: zero)
    launch_narrow_f32(ctx->stream, ctx->d_Xc, ctx->d_X32, plan.in_elems * B);
  else if (ctx->f32)
    launch_narrow_f32(ctx->stream, ctx->d_X, ctx->d_X32, (int64_t)in_dim * B);   // X rounded to fp32 once
  SI_HIP(ctx, hipGetLastError());
  SI_HIP(ctx, hipStreamSynchronize(ctx->stream));
  ctx->i_ready = true;
  return SI_OK;
}

int32_t si_infer_setup(si_ctx* ctx, const si_layer* layers, int32_t L, int64_t N, int32_t M, const double* W_swa,
                       const double* P, const double* X, const double* Y, int32_t in_dim, int32_t out_dim,
                       int64_t B, double sigma_m, int32_t compute_dtype) {
  return infer_setup_common(ctx, layers, L, N, M, W_swa, P, N, X, Y, in_dim, out_dim, B, sigma_m, compute_dtype, SRC_HOST);
}

int32_t si_infer_setup_dev(si_ctx* ctx, const si_layer* layers, int32_t L, int64_t N, int32_t M, const double* W_swa_dev,
                           const double* P_dev, int64_t ldP, int32_t borrow, const double* X_dev, const double* Y_dev,
                           int32_t in_dim, int32_t out_dim, int64_t B, double sigma_m, int32_t compute_dtype) {
  CHECK_CTX(ctx);
  if (W_swa_dev && P_dev) {
    if (ldP < N) return fail(ctx, SI_ERR_INVALID, "si_infer_setup_dev: ldP < N");
    if (borrow) {
      // the streaming kernels read rows in 16-byte pairs: row N of an odd-length column must exist and be aligned
      if ((ldP & 1) || ldP < N + (N & 1) || (reinterpret_cast<uintptr_t>(P_dev) & 15u) ||
          (reinterpret_cast<uintptr_t>(W_swa_dev) & 15u))
        return fail(ctx, SI_ERR_INVALID,
                    "si_infer_setup_dev: borrowed W_swa / P need 16-byte aligned bases, an even ldP >= N + (N mod 2) and "
                    "N + (N mod 2) readable elements of W_swa");
    }
  }
  return infer_setup_common(ctx, layers, L, N, M, W_swa_dev, P_dev, ldP, X_dev, Y_dev, in_dim, out_dim, B, sigma_m,
                            compute_dtype, borrow ? SRC_DEV_BORROW : SRC_DEV_COPY);
}

int32_t si_construct_result_ptr(si_ctx* ctx, double** W_swa_dev_out, double** P_dev_out, int64_t* ld_out, int32_t* M_out) {
  CHECK_CTX(ctx);
  if (!ctx->c_finished) return fail(ctx, SI_ERR_STATE, "si_construct_result_ptr: no finished construction");
  if (W_swa_dev_out) *W_swa_dev_out = ctx->d_swa;
  if (P_dev_out) *P_dev_out = ctx->d_P;
  if (ld_out) *ld_out = ctx->ldA;
  if (M_out) *M_out = ctx->M_built;
  return SI_OK;
}

// How many chains one forward launch carries.  Small models leave most of the 256 CUs idle with one chain per launch
// (a 20-wide Dense layer on 1000 observations is 8 workgroups), so independent chains are stacked in grid.y; the
// workspace for that is capped so that a model whose single chain already fills the chip (cfg2: 1.5 GB of activations
// per chain) keeps one slot.
static constexpr double SI_BATCH_BYTES = 2.0 * 1024.0 * 1024.0 * 1024.0;
static int batch_width(const si_ctx* ctx, int C) {
  if (ctx->plan.has_conv) return 1;  // chains with Conv layers fill the chip one chain at a time
  const double pslots = (double)std::max(ctx->fuse_slots, ctx->fuse_slots32);
  const double per = (ctx->f32 ? 4.0 : 8.0) * 2.0 * (double)ctx->act_elems +
                     8.0 * ((pslots + 1.0) * (double)ctx->out_dim * (double)ctx->B + (ctx->f32 ? 1.5 : 1.0) * (double)pad_ld(ctx->iN) +
                            (double)ctx->sse_blocks);
  const double fit = std::floor(SI_BATCH_BYTES / per);
  return (int)std::max(1.0, std::min({(double)C, fit, 1024.0}));
}

int32_t ensure_chains(si_ctx* ctx, int32_t C) {
  const int want = batch_width(ctx, C);
  if (ctx->fw_slots < want) {
    SI_HIP(ctx, hipStreamSynchronize(ctx->stream));
    if (!alloc_forward(ctx, want)) {
      if (!alloc_forward(ctx, 1)) {
        ctx->i_ready = false;
        return fail(ctx, SI_ERR_NOMEM, "forward workspace allocation failed");
      }
    }
  }
  if (ctx->chains_cap >= C) return SI_OK;
  SI_HIP(ctx, hipStreamSynchronize(ctx->stream));
  dev_free(ctx->d_zcur);
  dev_free(ctx->d_zprop);
  dev_free(ctx->d_lpcur);
  dev_free(ctx->d_sse);
  dev_free(ctx->d_nacc);
  dev_free(ctx->d_steps);
  dev_free(ctx->d_wsq);
  ctx->chains_cap = 0;
  if (dev_alloc(&ctx->d_wsq, (size_t)C) != hipSuccess || dev_alloc(&ctx->d_zcur, (size_t)ctx->iM * C) != hipSuccess || dev_alloc(&ctx->d_zprop, (size_t)ctx->iM * C) != hipSuccess ||
      dev_alloc(&ctx->d_lpcur, (size_t)C) != hipSuccess || dev_alloc(&ctx->d_sse, (size_t)C) != hipSuccess ||
      dev_alloc(&ctx->d_nacc, (size_t)C) != hipSuccess || dev_alloc(&ctx->d_steps, (size_t)C) != hipSuccess)
    return fail(ctx, SI_ERR_NOMEM, "sampler state allocation failed");
  ctx->chains_cap = C;
  return SI_OK;
}

// compute_dtype = SI_F32: the Dense chain of eval_density on the fp32 matrix instruction (kernels_gemm_f32.hip).  K4 has
// left W_swa + P z in d_w (fp64) AND rounded once in d_w32; X32 / activations are fp32; the narrow head's partial sums, the
// last bias + activation (tail_sse_kernel, unchanged) and the sum of squared errors are fp64.  Replaces the same reference
// lines as the fp64 path, src/space_inference.jl:92-94, with the precision option of SURVEY section 0 Q6.
static int32_t eval_density_f32(si_ctx* ctx, int c0, int nc, const double** yhat_out) {
  const int64_t N = ctx->iN, B = ctx->B, ldw = pad_ld(N);
  const double dn = (double)nc;
  ChainBatch cb;
  cb.n = nc;
  cb.w = ldw;
  cb.hin = 0;  // X is shared by all chains
  cb.hout = ctx->act_elems;
  cb.part = (int64_t)ctx->fuse_slots32 * ctx->out_dim * B;
  const float* h = ctx->d_X32;
  const float* w = ctx->d_w32;
  const size_t nl = ctx->layers.size();
  const size_t nstored = ctx->fuse_tail ? nl - 2 : nl;
  for (size_t l = 0; l < nstored; ++l) {
    const si_layer& ly = ctx->layers[l];
    float* o = ctx->d_act32[l & 1];
    const double fl = 2.0 * (double)ly.in * (double)ly.out * (double)B * dn;
    const double by = ((double)ly.in * ly.out + ly.out + (double)(ly.in + ly.out) * (double)B) * 4.0 * dn;
    {
      ProfScope ps(ctx, SI_K_DENSE, fl, by);
      ProfScope pm((int)l == ctx->main_layer ? ctx : nullptr, SI_K_DENSE_MAIN, fl, by);
      launch_dense_f32(ctx->stream, w + ly.w_off, w + ly.b_off, h, o, ly.out, ly.in, B, ly.act, cb);
    }
    h = o;
    cb.hin = ctx->act_elems;
  }
  const int64_t d = (int64_t)ctx->out_dim * B;
  if (ctx->fuse_tail) {
    const si_layer& ly = ctx->layers[nl - 2];
    const si_layer& ll = ctx->layers[nl - 1];
    const double fl = (2.0 * (double)ly.in * (double)ly.out * (double)B + 2.0 * (double)ll.in * (double)ll.out * (double)B) * dn;
    const double by = (((double)ly.in * ly.out + ly.out + (double)ly.in * (double)B + (double)ll.in * ll.out) * 4.0 +
                       (double)ctx->fuse_slots32 * ll.out * (double)B * 8.0) * dn;
    {
      ProfScope ps(ctx, SI_K_DENSE, fl, by);
      ProfScope pm(((int)nl - 2 == ctx->main_layer || (int)nl - 1 == ctx->main_layer) ? ctx : nullptr, SI_K_DENSE_MAIN, fl, by);
      launch_dense_f32_fused(ctx->stream, w + ly.w_off, w + ly.b_off, h, ly.out, ly.in, B, ly.act, w + ll.w_off, ll.out,
                             ctx->d_part, cb);
    }
    {
      ProfScope ps(ctx, SI_K_SSE, (3.0 + ctx->fuse_slots32) * (double)d * dn, (16.0 + 8.0 * ctx->fuse_slots32) * (double)d * dn);
      // the head's bias is added in fp64 from the fp64 weight vector (same number the fp32 copy was rounded from)
      launch_tail_sse(ctx->stream, ctx->d_part, ctx->fuse_slots32, ll.out, B, ctx->d_w + ll.b_off, ll.act, ctx->d_Y,
                      yhat_out ? ctx->d_yhat : nullptr, ctx->d_ssepart, ctx->sse_blocks, cb);
      if (!ctx->defer_sse_final) launch_sse_final(ctx->stream, ctx->d_ssepart, ctx->sse_blocks, ctx->d_sse + c0, nc);
    }
  } else {
    ProfScope ps(ctx, SI_K_SSE, 3.0 * (double)d * dn, 12.0 * (double)d * dn);
    launch_sse_f32(ctx->stream, h, ctx->d_Y, d, ctx->d_ssepart, ctx->sse_blocks, ctx->d_sse + c0, nc, ctx->act_elems,
                   yhat_out ? ctx->d_yhat : nullptr, d, !ctx->defer_sse_final);
  }
  SI_HIP(ctx, hipGetLastError());
  if (yhat_out) *yhat_out = ctx->d_yhat;   // slot j at + j * out_dim*B
  return SI_OK;
}

// density evaluations for chain slots [c0, c0 + nc), nc <= fw_slots, in ONE pass of launches:
// d_zprop[:, c] -> d_sse[c]; optionally leaves the model outputs at *yhat_out (slot j at + j * out_dim*B after the fused
// tail, at + j * act_elems otherwise)
int32_t eval_density(si_ctx* ctx, int c0, int nc, const double** yhat_out) {
  ctx->last_density_spec = 0;   // (si_chain_kernel_info reports the last evaluation)
  const int64_t N = ctx->iN, B = ctx->B, ldw = pad_ld(N);
  const int32_t M = ctx->iM;
  const double dn = (double)nc;
  {
    ProfScope ps(ctx, SI_K_RECON, 2.0 * (double)N * M * dn, (double)N * (M + 1 + dn) * 8.0);
    launch_reconstruct(ctx->stream, ctx->i_swa, ctx->i_P, ctx->ldP, N, M, ctx->d_zprop + (size_t)c0 * M, nc, ctx->d_w, ldw,
                       ctx->num_cu, ctx->f32 ? ctx->d_w32 : nullptr, ldw);
  }
  if (ctx->sigma_p > 0.0)  // ||new_W||^2 per chain for the optional prior term (same fixed-order reduction as the SSE)
    launch_sse(ctx->stream, ctx->d_w, nullptr, N, ctx->d_wsqpart, ctx->wsq_blocks, ctx->d_wsq + c0, nc, ldw);
  if (ctx->plan.has_conv && ctx->f32) {
    // compute_dtype = SI_F32 on a Conv chain: the same pass on fp32 operands (net_forward_f32); the squared errors in fp64
    float* last32 = nullptr;
    const int32_t rc = net_forward_f32(ctx, ctx->plan, ctx->d_w32, ctx->d_X32, B, ctx->d_act32, ctx->d_wpack32, &last32);
    if (rc != SI_OK) return rc;
    const int64_t d = (int64_t)ctx->out_dim * B;
    {
      ProfScope ps(ctx, SI_K_SSE, 3.0 * (double)d, 12.0 * (double)d);
      launch_sse_f32(ctx->stream, last32, ctx->d_Y, d, ctx->d_ssepart, ctx->sse_blocks, ctx->d_sse + c0, 1, ctx->act_elems,
                     yhat_out ? ctx->d_yhat : nullptr, d, !ctx->defer_sse_final);
    }
    SI_HIP(ctx, hipGetLastError());
    if (yhat_out) *yhat_out = ctx->d_yhat;
    return SI_OK;
  }
  if (ctx->plan.has_conv) {
    // generic path (capi_net.hip): Conv / MaxPool / flatten / Dense layers one after the other, ping-pong activations
    double* last = nullptr;
    const int32_t rc = net_forward(ctx, ctx->plan, ctx->d_w, ctx->plan.input_spatial ? ctx->d_Xc : ctx->d_X, B, ctx->d_act,
                                   ctx->d_wpack, /*pingpong=*/true, &last);
    if (rc != SI_OK) return rc;
    const int64_t d = (int64_t)ctx->out_dim * B;
    {
      ProfScope ps(ctx, SI_K_SSE, 3.0 * (double)d, 16.0 * (double)d);
      launch_sse(ctx->stream, last, ctx->d_Y, d, ctx->d_ssepart, ctx->sse_blocks, ctx->d_sse + c0, 1, ctx->act_elems, !ctx->defer_sse_final);
    }
    SI_HIP(ctx, hipGetLastError());
    if (yhat_out) *yhat_out = last;
    return SI_OK;
  }
#ifdef SI_DEV_KNOBS   // development build only: measured 1 % slower, not shipped
  const size_t nl_all = ctx->layers.size();
  if (ctx->overlap_halves && nc == 1 && ctx->fuse_tail && !yhat_out && B >= 4096) {
    // EXPERIMENT: the batch in two halves on two streams -- layer 1 of half B runs beside layer 2 of
    // half A, so the output-store drain of one overlaps the MFMAs of the other inside ONE chain.  The halves meet on whole
    // 128-column tiles, so every tile is computed exactly as in the single launch; the head partials of both halves land
    // in one buffer with the full-B pitch and ONE tail_sse launch sums them in the usual fixed order: lp is bit-identical.
    // (each under its own null check: si_reconstruct / the streamed output map create stream2 by themselves -- ADVICE r2)
    if (!ctx->stream2) SI_HIP(ctx, hipStreamCreateWithFlags(&ctx->stream2, hipStreamNonBlocking));
    if (!ctx->ev_fork) SI_HIP(ctx, hipEventCreateWithFlags(&ctx->ev_fork, hipEventDisableTiming));
    if (!ctx->ev_join) SI_HIP(ctx, hipEventCreateWithFlags(&ctx->ev_join, hipEventDisableTiming));
    const int64_t b1 = ((B / 2 + 127) / 128) * 128;
    SI_HIP(ctx, hipEventRecord(ctx->ev_fork, ctx->stream));
    SI_HIP(ctx, hipStreamWaitEvent(ctx->stream2, ctx->ev_fork, 0));
    auto run_half = [&](hipStream_t st, int64_t b0, int64_t bh, bool prof) {
      const double* h = ctx->d_X + (size_t)ctx->in_dim * b0;
      const size_t nst = nl_all - 2;
      for (size_t l = 0; l < nst; ++l) {
        const si_layer& ly = ctx->layers[l];
        double* o = ctx->d_act[l & 1] + (size_t)ly.out * b0;
        ProfScope ps(prof ? ctx : nullptr, SI_K_DENSE, 2.0 * (double)ly.in * ly.out * (double)bh, 0.0);
        launch_dense_f64(st, ctx->d_w + ly.w_off, ctx->d_w + ly.b_off, h, o, ly.out, ly.in, bh, ly.act);
        h = o;
      }
      const si_layer& ly = ctx->layers[nl_all - 2];
      const si_layer& ll = ctx->layers[nl_all - 1];
      ChainBatch hb;
      hb.part_ld = B;
      const double fl = (2.0 * (double)ly.in * ly.out + 2.0 * (double)ll.in * ll.out) * (double)bh;
      ProfScope ps(prof ? ctx : nullptr, SI_K_DENSE, fl, 0.0);
      ProfScope pm(prof ? ctx : nullptr, SI_K_DENSE_MAIN, fl, 0.0);
      launch_dense_f64_fused(st, ctx->d_w + ly.w_off, ctx->d_w + ly.b_off, h, ly.out, ly.in, bh, ly.act, ctx->d_w + ll.w_off, ll.out,
                             ctx->d_part + b0, hb);
    };
    run_half(ctx->stream2, b1, B - b1, false);
    run_half(ctx->stream, 0, b1, true);
    SI_HIP(ctx, hipEventRecord(ctx->ev_join, ctx->stream2));
    SI_HIP(ctx, hipStreamWaitEvent(ctx->stream, ctx->ev_join, 0));
    const si_layer& ll = ctx->layers[nl_all - 1];
    const int64_t d = (int64_t)ctx->out_dim * B;
    ChainBatch cb1;
    cb1.part = (int64_t)ctx->fuse_slots * ctx->out_dim * B;
    cb1.w = ldw;
    {
      ProfScope ps(ctx, SI_K_SSE, (3.0 + ctx->fuse_slots) * (double)d, (16.0 + 8.0 * ctx->fuse_slots) * (double)d);
      launch_tail_sse(ctx->stream, ctx->d_part, ctx->fuse_slots, ll.out, B, ctx->d_w + ll.b_off, ll.act, ctx->d_Y, nullptr,
                      ctx->d_ssepart, ctx->sse_blocks, cb1);
      if (!ctx->defer_sse_final) launch_sse_final(ctx->stream, ctx->d_ssepart, ctx->sse_blocks, ctx->d_sse + c0, 1);
    }
    SI_HIP(ctx, hipGetLastError());
    return SI_OK;
  }
#endif  // SI_DEV_KNOBS
  if (ctx->f32) return eval_density_f32(ctx, c0, nc, yhat_out);
  if (ctx->fused_ok && ctx->chain_mode != 0 && !yhat_out) {
    // narrow chain: every layer of all nc chains in ONE launch, activations in LDS; the model outputs go through the plain
    // SSE kernels (one squared error per thread: the same partial sums as tail_sse_kernel's)
    ChainFusedPlan fp;
    int nb = 1;
    bool wave_tiles = false;
    const size_t lds = fused_plan_for(ctx, nc, fp, &nb, &wave_tiles);
    if (lds != 0) {
      const int64_t d = (int64_t)ctx->out_dim * B;
      {
        const double fl = 2.0 * (double)N * (double)B * dn;
        const double by = ((double)N * dn + (double)ctx->in_dim * (double)B + (double)d * dn) * 8.0;
        ProfScope ps(ctx, SI_K_DENSE, fl, by);
        ProfScope pm(ctx, SI_K_DENSE_MAIN, fl, by);
        const SpecKernels* sk = nullptr;
        if (ctx->chain_spec && ctx->fuse_tail)   // the same kernel compiled for this chain's shapes (chain_spec_rtc.cpp; same bits)
          sk = spec_kernels(ctx->layers.data(), (int)ctx->layers.size(), nb, dense_fused_slot_feats(ctx->layers[ctx->layers.size() - 2].out), 0, false,
                            &ctx->spec_message);
        ctx->last_density_spec = sk != nullptr;
        if (sk) {
          const double* wp = ctx->d_w;
          const double* xp = ctx->d_X;
          double* yp = ctx->d_yhat;
          long long ws = ldw, ys = d;
          int Bi = (int)B;
          void* args[] = {&wp, &ws, &xp, &yp, &ys, &Bi};
          if (sk->stack && nc >= 16) {
            // many chains: a wave per 16 observations with the activations in registers and the chain's weights in LDS, no barrier
            // between layers (329 against 495 us at 512 chains of the nn_example model, 48 against 71 at 64; 24 against 14 at 8)
            const int splits = nc >= ctx->num_cu ? 1 : std::min(8, (ctx->num_cu + nc - 1) / nc);
            SI_HIP(ctx, hipModuleLaunchKernel(sk->stack, (unsigned)splits, (unsigned)nc, 1, 512, 1, 1,
                                              (unsigned)((size_t)sk->wvec_doubles * sizeof(double)), ctx->stream, args, nullptr));
          } else {
            SI_HIP(ctx, hipModuleLaunchKernel(sk->fused, (unsigned)((B + 16 * nb - 1) / (16 * nb)), (unsigned)nc, 1, 256, 1, 1,
                                              (unsigned)((size_t)sk->lds_doubles * sizeof(double)), ctx->stream, args, nullptr));
          }
        } else {
          launch_chain_fused(ctx->stream, fp, nb, wave_tiles, lds, ctx->d_w, ldw, ctx->d_X, ctx->d_yhat, d, nc);
        }
      }
      {
        ProfScope ps(ctx, SI_K_SSE, 3.0 * (double)d * dn, 16.0 * (double)d * dn);
        launch_sse(ctx->stream, ctx->d_yhat, ctx->d_Y, d, ctx->d_ssepart, ctx->sse_blocks, ctx->d_sse + c0, nc, d, !ctx->defer_sse_final);
      }
      SI_HIP(ctx, hipGetLastError());
      return SI_OK;
    }
  }
  ChainBatch cb;
  cb.n = nc;
  cb.w = ldw;
  cb.hin = 0;  // X is shared by all chains
  cb.hout = ctx->act_elems;
  cb.part = (int64_t)ctx->fuse_slots * ctx->out_dim * B;
  const double* h = ctx->d_X;
  const size_t nl = ctx->layers.size();
  const size_t nstored = ctx->fuse_tail ? nl - 2 : nl;
  for (size_t l = 0; l < nstored; ++l) {
    const si_layer& ly = ctx->layers[l];
    double* o = ctx->d_act[l & 1];
    const double fl = 2.0 * (double)ly.in * (double)ly.out * (double)B * dn;
    const double by = ((double)ly.in * ly.out + ly.out + (double)(ly.in + ly.out) * (double)B) * 8.0 * dn;
    {
      ProfScope ps(ctx, SI_K_DENSE, fl, by);
      ProfScope pm((int)l == ctx->main_layer ? ctx : nullptr, SI_K_DENSE_MAIN, fl, by);
      launch_dense_f64(ctx->stream, ctx->d_w + ly.w_off, ctx->d_w + ly.b_off, h, o, ly.out, ly.in, B, ly.act, cb);
    }
    h = o;
    cb.hin = ctx->act_elems;
  }
  const int64_t d = (int64_t)ctx->out_dim * B;
  if (ctx->fuse_tail) {
    const si_layer& ly = ctx->layers[nl - 2];
    const si_layer& ll = ctx->layers[nl - 1];
    const double fl = (2.0 * (double)ly.in * (double)ly.out * (double)B + 2.0 * (double)ll.in * (double)ll.out * (double)B) * dn;
    const double by = (((double)ly.in * ly.out + ly.out + (double)ly.in * (double)B + (double)ll.in * ll.out) * 8.0 +
                       (double)ctx->fuse_slots * ll.out * (double)B * 8.0) * dn;
    {
      ProfScope ps(ctx, SI_K_DENSE, fl, by);
      ProfScope pm(((int)nl - 2 == ctx->main_layer || (int)nl - 1 == ctx->main_layer) ? ctx : nullptr, SI_K_DENSE_MAIN, fl, by);
      launch_dense_f64_fused(ctx->stream, ctx->d_w + ly.w_off, ctx->d_w + ly.b_off, h, ly.out, ly.in, B, ly.act,
                             ctx->d_w + ll.w_off, ll.out, ctx->d_part, cb);
    }
    {
      ProfScope ps(ctx, SI_K_SSE, (3.0 + ctx->fuse_slots) * (double)d * dn, (16.0 + 8.0 * ctx->fuse_slots) * (double)d * dn);
      launch_tail_sse(ctx->stream, ctx->d_part, ctx->fuse_slots, ll.out, B, ctx->d_w + ll.b_off, ll.act, ctx->d_Y,
                      yhat_out ? ctx->d_yhat : nullptr, ctx->d_ssepart, ctx->sse_blocks, cb);
      if (!ctx->defer_sse_final) launch_sse_final(ctx->stream, ctx->d_ssepart, ctx->sse_blocks, ctx->d_sse + c0, nc);
    }
    h = ctx->d_yhat;
  } else {
    ProfScope ps(ctx, SI_K_SSE, 3.0 * (double)d * dn, 16.0 * (double)d * dn);
    launch_sse(ctx->stream, h, ctx->d_Y, d, ctx->d_ssepart, ctx->sse_blocks, ctx->d_sse + c0, nc, ctx->act_elems, !ctx->defer_sse_final);
  }
  SI_HIP(ctx, hipGetLastError());
  if (yhat_out) *yhat_out = h;
  return SI_OK;
}

// all C chain slots, fw_slots at a time
int32_t eval_density_all(si_ctx* ctx, int C) {
  const int w = std::max(1, ctx->fw_slots);
  for (int c0 = 0; c0 < C; c0 += w) {
    const int32_t rc = eval_density(ctx, c0, std::min(w, C - c0), nullptr);
    if (rc != SI_OK) return rc;
  }
  return SI_OK;
}

double mvnormal_c0(double d, double sigma) {
  // Distributions.mvnormal_c0: -(d*log(2pi) + logdet(Sigma))/2 with logdet = d*log(sigma^2)
  return -(d * std::log(2.0 * 3.14159265358979323846) + d * std::log(sigma * sigma)) / 2.0;
}

// log N(w; 0, sigma_p^2 I) = c0p - ||w||^2 / (2 sigma_p^2): the term the reference writes AFTER its `return` (Q4)
double prior_c0(const si_ctx* ctx) { return ctx->sigma_p > 0.0 ? mvnormal_c0((double)ctx->iN, ctx->sigma_p) : 0.0; }

int32_t si_infer_set_prior(si_ctx* ctx, double sigma_p) {
  CHECK_CTX(ctx);
  if (!ctx->i_ready) return fail(ctx, SI_ERR_STATE, "si_infer_set_prior: call si_infer_setup first");
  if (ctx->sw_Z) return fail(ctx, SI_ERR_STATE, "si_infer_set_prior: a step-wise RWMH session is open");
  if (!(sigma_p >= 0.0)) return fail(ctx, SI_ERR_INVALID, "si_infer_set_prior: sigma_p must be >= 0 (0 = off, the reference's behaviour)");
  ctx->sigma_p = sigma_p;
  return SI_OK;
}

int32_t si_logdensity(si_ctx* ctx, const double* Z, int32_t C, double* lp_out) {
  CHECK_CTX(ctx);
  if (!ctx->i_ready) return fail(ctx, SI_ERR_STATE, "si_logdensity: call si_infer_setup first");
  if (!Z || C <= 0 || !lp_out) return fail(ctx, SI_ERR_INVALID, "si_logdensity: bad argument");
  if (ctx->sw_Z) return fail(ctx, SI_ERR_STATE, "si_logdensity: a step-wise RWMH session is open; its proposal / SSE buffers are shared (si_rwmh_end or si_rwmh_abort first)");
  BIND(ctx);
  int32_t rc = ensure_chains(ctx, C);
  if (rc != SI_OK) return rc;
  SI_HIP(ctx, hipMemcpyAsync(ctx->d_zprop, Z, (size_t)ctx->iM * C * sizeof(double), hipMemcpyHostToDevice, ctx->stream));
  if ((rc = eval_density_all(ctx, C)) != SI_OK) return rc;
  std::vector<double> sse((size_t)C), wsq((size_t)C, 0.0);
  SI_HIP(ctx, hipMemcpyAsync(sse.data(), ctx->d_sse, (size_t)C * sizeof(double), hipMemcpyDeviceToHost, ctx->stream));
  if (ctx->sigma_p > 0.0)
    SI_HIP(ctx, hipMemcpyAsync(wsq.data(), ctx->d_wsq, (size_t)C * sizeof(double), hipMemcpyDeviceToHost, ctx->stream));
  SI_HIP(ctx, hipStreamSynchronize(ctx->stream));
  const double d = (double)ctx->out_dim * (double)ctx->B;
  const double c0 = mvnormal_c0(d, ctx->sigma_m), s2 = ctx->sigma_m * ctx->sigma_m;
  for (int c = 0; c < C; ++c) {
    lp_out[c] = c0 - (sse[(size_t)c] / s2) / 2.0;
    if (ctx->sigma_p > 0.0) lp_out[c] += prior_c0(ctx) - (wsq[(size_t)c] / (ctx->sigma_p * ctx->sigma_p)) / 2.0;
  }
  return SI_OK;
}

static int32_t ensure_grad(si_ctx* ctx) {
  if (ctx->g_ready) return SI_OK;
  const int64_t B = ctx->B;
  if (ctx->plan.has_conv && ctx->f32)
    return fail(ctx, SI_ERR_INVALID, "si_logdensity_grad: compute_dtype = SI_F32 has a reverse sweep for Dense chains only; set a Conv chain up with SI_F64 for gradients");
  if (ctx->plan.has_conv) {
    const NetPlan& p = ctx->plan;
    size_t nb, nr, nw, nd;
    net_scratch_sizes(p, B, ctx->num_cu, &nb, &nr, &nw, &nd);
    ctx->d_hs.assign(p.L.size(), nullptr);
    ctx->d_pidx.assign(p.L.size(), nullptr);
    bool ok = true;
    for (size_t l = 0; l < p.L.size() && ok; ++l) {
      if (net_grad_fused(p, l))   // Conv + MaxPool as one kernel: a byte index instead of the un-pooled activation
        ok = dev_alloc(&ctx->d_pidx[l], net_pidx_bytes(p, l, B)) == hipSuccess;
      else
        ok = dev_alloc(&ctx->d_hs[l], (size_t)p.L[l].out_elems * B) == hipSuccess;
    }
    ctx->g_scratch.pidx = ctx->d_pidx.data();
    ok = ok && dev_alloc(&ctx->d_delta[0], (size_t)p.max_elems * B) == hipSuccess &&
         dev_alloc(&ctx->d_delta[1], (size_t)p.max_elems * B) == hipSuccess &&
         dev_alloc(&ctx->d_gw, (size_t)pad_ld(ctx->iN)) == hipSuccess && dev_alloc(&ctx->g_scratch.bwpart, nb) == hipSuccess &&
         dev_alloc(&ctx->g_scratch.rspart, nr) == hipSuccess && dev_alloc(&ctx->g_scratch.wt, nw) == hipSuccess &&
         dev_alloc(&ctx->g_scratch.dbtmp, nd) == hipSuccess &&
         dev_alloc(&ctx->d_ptgpart, (size_t)ptg_blocks() * ctx->iM) == hipSuccess && dev_alloc(&ctx->d_gz, (size_t)ctx->iM) == hipSuccess;
    if (!ok) {
      for (auto& h : ctx->d_hs) dev_free(h);
      ctx->d_hs.clear();
      for (auto& h : ctx->d_pidx) dev_free(h);
      ctx->d_pidx.clear();
      ctx->g_scratch.pidx = nullptr;
      dev_free(ctx->d_delta[0]); dev_free(ctx->d_delta[1]); dev_free(ctx->d_gw); dev_free(ctx->g_scratch.bwpart);
      dev_free(ctx->g_scratch.rspart); dev_free(ctx->g_scratch.wt); dev_free(ctx->g_scratch.dbtmp); dev_free(ctx->d_ptgpart);
      dev_free(ctx->d_gz);
      return fail(ctx, SI_ERR_NOMEM, "si_logdensity_grad: workspace allocation failed");
    }
    ctx->g_ready = true;
    return SI_OK;
  }
  if (ctx->f32) {   // compute_dtype = SI_F32: the fp32 forward + reverse sweep (kernels_bwd_f32.hip), P' g in fp64
    ctx->g_ws32 = new SweepF32Ws();
    const bool ok32 = sweep_f32_alloc(ctx, *ctx->g_ws32, ctx->layers.data(), (int)ctx->layers.size(), ctx->fuse_tail, ctx->iN, ctx->in_dim,
                                      ctx->out_dim, B) &&
                      dev_alloc(&ctx->d_gw, (size_t)pad_ld(ctx->iN)) == hipSuccess &&
                      dev_alloc(&ctx->d_ptgpart, (size_t)ptg_blocks() * ctx->iM) == hipSuccess && dev_alloc(&ctx->d_gz, (size_t)ctx->iM) == hipSuccess;
    if (!ok32) {
      sweep_f32_free(*ctx->g_ws32);
      delete ctx->g_ws32;
      ctx->g_ws32 = nullptr;
      dev_free(ctx->d_gw); dev_free(ctx->d_ptgpart); dev_free(ctx->d_gz);
      return fail(ctx, SI_ERR_NOMEM, "si_logdensity_grad: workspace allocation failed");
    }
    ctx->g_ready = true;
    return SI_OK;
  }
  int64_t maxw = 1;
  size_t maxpart = 1;
  ctx->d_hs.assign(ctx->layers.size(), nullptr);
  bool ok = true;
  for (size_t l = 0; l < ctx->layers.size() && ok; ++l) {
    const si_layer& ly = ctx->layers[l];
    maxw = std::max<int64_t>(maxw, ly.out);
    maxpart = std::max(maxpart, backward_weight_part_elems(ly.out, ly.in, B, ctx->num_cu));
    ok = dev_alloc(&ctx->d_hs[l], (size_t)ly.out * B) == hipSuccess;
  }
  if (ctx->fuse_tail) maxpart = std::max(maxpart, tail_bwd_part_elems(ctx->layers.back().out, ctx->layers.back().in));
  ok = ok && dev_alloc(&ctx->d_delta[0], (size_t)maxw * B) == hipSuccess &&
       dev_alloc(&ctx->d_delta[1], (size_t)maxw * B) == hipSuccess &&
       dev_alloc(&ctx->d_gw, (size_t)pad_ld(ctx->iN)) == hipSuccess && dev_alloc(&ctx->d_bwpart, maxpart) == hipSuccess &&
       dev_alloc(&ctx->d_rspart, (size_t)rowsum_chunks() * maxw) == hipSuccess &&
       dev_alloc(&ctx->d_ptgpart, (size_t)ptg_blocks() * ctx->iM) == hipSuccess && dev_alloc(&ctx->d_gz, (size_t)ctx->iM) == hipSuccess;
  if (!ok) {
    for (auto& h : ctx->d_hs) dev_free(h);
    ctx->d_hs.clear();
    dev_free(ctx->d_delta[0]); dev_free(ctx->d_delta[1]); dev_free(ctx->d_gw); dev_free(ctx->d_bwpart);
    dev_free(ctx->d_rspart); dev_free(ctx->d_ptgpart); dev_free(ctx->d_gz);
    return fail(ctx, SI_ERR_NOMEM, "si_logdensity_grad: workspace allocation failed");
  }
  ctx->g_ready = true;
  return SI_OK;
}

int32_t si_logdensity_grad(si_ctx* ctx, const double* z, double* lp_out, double* grad_out) {
  CHECK_CTX(ctx);
  if (!ctx->i_ready) return fail(ctx, SI_ERR_STATE, "si_logdensity_grad: call si_infer_setup first");
  if (!z || !lp_out || !grad_out) return fail(ctx, SI_ERR_INVALID, "si_logdensity_grad: bad argument");
  if (ctx->sw_Z) return fail(ctx, SI_ERR_STATE, "si_logdensity_grad: a step-wise RWMH session is open; its proposal / SSE buffers are shared (si_rwmh_end or si_rwmh_abort first)");
  BIND(ctx);
  int32_t rc = ensure_chains(ctx, 1);
  if (rc != SI_OK) return rc;
  if ((rc = ensure_grad(ctx)) != SI_OK) return rc;
  const int64_t N = ctx->iN, B = ctx->B;
  const int32_t M = ctx->iM;
  const size_t nl = ctx->layers.size();
  const double s2 = ctx->sigma_m * ctx->sigma_m;
  SI_HIP(ctx, hipMemcpyAsync(ctx->d_zprop, z, (size_t)M * sizeof(double), hipMemcpyHostToDevice, ctx->stream));
  {
    ProfScope ps(ctx, SI_K_RECON, 2.0 * (double)N * M, (double)N * (M + 2) * 8.0);
    launch_reconstruct(ctx->stream, ctx->i_swa, ctx->i_P, ctx->ldP, N, M, ctx->d_zprop, 1, ctx->d_w, pad_ld(N), ctx->num_cu,
                       ctx->f32 ? ctx->d_w32 : nullptr, pad_ld(N));
  }
  const bool prior = ctx->sigma_p > 0.0;
  if (prior) launch_sse(ctx->stream, ctx->d_w, nullptr, N, ctx->d_wsqpart, ctx->wsq_blocks, ctx->d_wsq, 1, pad_ld(N));
  if (ctx->f32) {
    // compute_dtype = SI_F32: value and gradient on the fp32 density's own arithmetic (fp32 operands and activations, fp64 head
    // partials / SSE / batch sums; W_swa + P z rounded once), the pull-back P' g and the optional prior term in fp64
    const int64_t d = (int64_t)ctx->out_dim * B;
    DenseSweepF32 sw{ctx->layers.data(), nl, ctx->fuse_tail, ctx->d_w32, ctx->d_w, ctx->d_X32, ctx->d_Y, ctx->g_ws32, ctx->d_part,
                     ctx->d_ssepart, ctx->d_sse, ctx->sse_blocks, B, N, 1.0 / s2};   // d lp / d yhat = (y - yhat) / sigma^2
    if ((rc = dense_value_and_grad_f32(ctx, ctx->stream, sw)) != SI_OK) return rc;
    launch_widen_f32_to_f64(ctx->stream, ctx->g_ws32->gw32, N, ctx->d_gw, ctx->num_cu);
    if (prior) launch_prior_grad(ctx->stream, ctx->d_gw, ctx->d_w, N, 1.0 / (ctx->sigma_p * ctx->sigma_p), ctx->num_cu);
    launch_ptg(ctx->stream, ctx->i_P, ctx->ldP, N, M, ctx->d_gw, ctx->d_ptgpart, ctx->d_gz);
    SI_HIP(ctx, hipGetLastError());
    double sse = 0.0, wsq = 0.0;
    SI_HIP(ctx, hipMemcpyAsync(&sse, ctx->d_sse, sizeof(double), hipMemcpyDeviceToHost, ctx->stream));
    SI_HIP(ctx, hipMemcpyAsync(grad_out, ctx->d_gz, (size_t)M * sizeof(double), hipMemcpyDeviceToHost, ctx->stream));
    if (prior) SI_HIP(ctx, hipMemcpyAsync(&wsq, ctx->d_wsq, sizeof(double), hipMemcpyDeviceToHost, ctx->stream));
    SI_HIP(ctx, hipStreamSynchronize(ctx->stream));
    *lp_out = mvnormal_c0((double)d, ctx->sigma_m) - (sse / s2) / 2.0;
    if (prior) *lp_out += prior_c0(ctx) - (wsq / (ctx->sigma_p * ctx->sigma_p)) / 2.0;
    return SI_OK;
  }
  if (ctx->plan.has_conv) {
    // generic path: forward with every output kept, d lp / d yhat = (y - yhat) / sigma^2, reverse sweep, P' g_w
    const NetPlan& p = ctx->plan;
    const double* xin = p.input_spatial ? ctx->d_Xc : ctx->d_X;
    if ((rc = net_forward(ctx, p, ctx->d_w, xin, B, ctx->d_hs.data(), ctx->d_wpack, false, nullptr, ctx->d_pidx.data())) != SI_OK) return rc;
    const int64_t d = (int64_t)ctx->out_dim * B;
    const double* yhat = ctx->d_hs[nl - 1];
    launch_sse(ctx->stream, yhat, ctx->d_Y, d, ctx->d_ssepart, ctx->sse_blocks, ctx->d_sse);
    {
      double bflops = 0.0;
      for (const auto& q : p.L)
        bflops += q.kind == SI_LAYER_DENSE ? 4.0 * (double)q.in_feat * q.out_feat * (double)B
                  : q.kind == SI_LAYER_CONV ? 4.0 * (double)q.KW * q.KH * q.C * q.Co * (double)q.Wo * q.Ho * (double)B : 0.0;
      ProfScope ps(ctx, SI_K_BACKWARD, bflops, 0.0);
      SI_HIP(ctx, hipMemsetAsync(ctx->d_gw, 0, (size_t)pad_ld(N) * sizeof(double), ctx->stream));
      launch_delta_out(ctx->stream, ctx->d_Y, yhat, d, 1.0 / s2, SI_ACT_IDENTITY, ctx->d_delta[0]);
      if ((rc = net_backward(ctx, p, ctx->d_w, xin, B, ctx->d_hs.data(), ctx->d_delta[0], ctx->d_delta[1], ctx->d_gw,
                             ctx->g_scratch)) != SI_OK)
        return rc;
      if (prior) launch_prior_grad(ctx->stream, ctx->d_gw, ctx->d_w, N, 1.0 / (ctx->sigma_p * ctx->sigma_p), ctx->num_cu);
      launch_ptg(ctx->stream, ctx->i_P, ctx->ldP, N, M, ctx->d_gw, ctx->d_ptgpart, ctx->d_gz);
    }
    SI_HIP(ctx, hipGetLastError());
    double sse = 0.0;
    SI_HIP(ctx, hipMemcpyAsync(&sse, ctx->d_sse, sizeof(double), hipMemcpyDeviceToHost, ctx->stream));
    SI_HIP(ctx, hipMemcpyAsync(grad_out, ctx->d_gz, (size_t)M * sizeof(double), hipMemcpyDeviceToHost, ctx->stream));
    double wsq = 0.0;
    if (prior) SI_HIP(ctx, hipMemcpyAsync(&wsq, ctx->d_wsq, sizeof(double), hipMemcpyDeviceToHost, ctx->stream));
    SI_HIP(ctx, hipStreamSynchronize(ctx->stream));
    *lp_out = mvnormal_c0((double)d, ctx->sigma_m) - (sse / s2) / 2.0;
    if (prior) *lp_out += prior_c0(ctx) - (wsq / (ctx->sigma_p * ctx->sigma_p)) / 2.0;
    return SI_OK;
  }
  // forward with every layer's output kept for the reverse sweep.  With a narrow head (fuse_tail) the layer in front
  // of it stores its output AND feeds the head from its epilogue, so the head costs no pass over that activation.
  const double* h = ctx->d_X;
  const size_t nplain = ctx->fuse_tail ? nl - 2 : nl;
  for (size_t l = 0; l < nplain; ++l) {
    const si_layer& ly = ctx->layers[l];
    ProfScope ps(ctx, SI_K_DENSE, 2.0 * (double)ly.in * ly.out * (double)B,
                 ((double)ly.in * ly.out + ly.out + (double)(ly.in + ly.out) * (double)B) * 8.0);
    launch_dense_f64(ctx->stream, ctx->d_w + ly.w_off, ctx->d_w + ly.b_off, h, ctx->d_hs[l], ly.out, ly.in, B, ly.act);
    h = ctx->d_hs[l];
  }
  const int64_t d = (int64_t)ctx->out_dim * B;
  if (ctx->fuse_tail) {
    const si_layer& ly = ctx->layers[nl - 2];
    const si_layer& ll = ctx->layers[nl - 1];
    {
      ProfScope ps(ctx, SI_K_DENSE, 2.0 * ((double)ly.in * ly.out + (double)ll.in * ll.out) * (double)B,
                   ((double)ly.in * ly.out + ly.out + (double)(ly.in + ly.out) * (double)B) * 8.0);
      launch_dense_f64_fused(ctx->stream, ctx->d_w + ly.w_off, ctx->d_w + ly.b_off, h, ly.out, ly.in, B, ly.act,
                             ctx->d_w + ll.w_off, ll.out, ctx->d_part, ChainBatch(), ctx->d_hs[nl - 2]);
    }
    ProfScope ps(ctx, SI_K_SSE, (3.0 + ctx->fuse_slots) * (double)d, (16.0 + 8.0 * ctx->fuse_slots) * (double)d);
    launch_tail_sse(ctx->stream, ctx->d_part, ctx->fuse_slots, ll.out, B, ctx->d_w + ll.b_off, ll.act, ctx->d_Y,
                    ctx->d_hs[nl - 1], ctx->d_ssepart, ctx->sse_blocks);
    launch_sse_final(ctx->stream, ctx->d_ssepart, ctx->sse_blocks, ctx->d_sse);
    h = ctx->d_hs[nl - 1];
  } else {
    ProfScope ps(ctx, SI_K_SSE, 3.0 * (double)d, 16.0 * (double)d);
    launch_sse(ctx->stream, h, ctx->d_Y, d, ctx->d_ssepart, ctx->sse_blocks, ctx->d_sse);
  }
  {
    double bflops = 0.0;
    for (const auto& ly : ctx->layers) bflops += 4.0 * (double)ly.in * ly.out * (double)B;
    ProfScope ps(ctx, SI_K_BACKWARD, bflops, 0.0);
    SI_HIP(ctx, hipMemsetAsync(ctx->d_gw, 0, (size_t)pad_ld(N) * sizeof(double), ctx->stream));
    // d lp / d yhat = (y - yhat) / sigma^2
    launch_delta_out(ctx->stream, ctx->d_Y, h, d, 1.0 / s2, ctx->layers[nl - 1].act, ctx->d_delta[0]);
    DenseSweep sw{ctx->layers.data(), nl, ctx->fuse_tail, ctx->d_w, ctx->d_X, ctx->d_hs.data(), {ctx->d_delta[0], ctx->d_delta[1]},
                  ctx->d_gw, ctx->d_rspart, ctx->d_bwpart, B};
    const int32_t rcs = dense_reverse_sweep(ctx, ctx->stream, sw);
    if (rcs != SI_OK) return rcs;
    if (prior) launch_prior_grad(ctx->stream, ctx->d_gw, ctx->d_w, N, 1.0 / (ctx->sigma_p * ctx->sigma_p), ctx->num_cu);
    launch_ptg(ctx->stream, ctx->i_P, ctx->ldP, N, M, ctx->d_gw, ctx->d_ptgpart, ctx->d_gz);
  }
  SI_HIP(ctx, hipGetLastError());
  double sse = 0.0;
  SI_HIP(ctx, hipMemcpyAsync(&sse, ctx->d_sse, sizeof(double), hipMemcpyDeviceToHost, ctx->stream));
  SI_HIP(ctx, hipMemcpyAsync(grad_out, ctx->d_gz, (size_t)M * sizeof(double), hipMemcpyDeviceToHost, ctx->stream));
  double wsq = 0.0;
  if (prior) SI_HIP(ctx, hipMemcpyAsync(&wsq, ctx->d_wsq, sizeof(double), hipMemcpyDeviceToHost, ctx->stream));
  SI_HIP(ctx, hipStreamSynchronize(ctx->stream));
  *lp_out = mvnormal_c0((double)d, ctx->sigma_m) - (sse / s2) / 2.0;
  if (prior) *lp_out += prior_c0(ctx) - (wsq / (ctx->sigma_p * ctx->sigma_p)) / 2.0;
  return SI_OK;
}

int32_t si_forward(si_ctx* ctx, const double* z, double* Yhat_out) {
  CHECK_CTX(ctx);
  if (!ctx->i_ready) return fail(ctx, SI_ERR_STATE, "si_forward: call si_infer_setup first");
  if (!z || !Yhat_out) return fail(ctx, SI_ERR_INVALID, "si_forward: bad argument");
  if (ctx->sw_Z) return fail(ctx, SI_ERR_STATE, "si_forward: a step-wise RWMH session is open; its proposal / SSE buffers are shared (si_rwmh_end or si_rwmh_abort first)");
  BIND(ctx);
  int32_t rc = ensure_chains(ctx, 1);
  if (rc != SI_OK) return rc;
  SI_HIP(ctx, hipMemcpyAsync(ctx->d_zprop, z, (size_t)ctx->iM * sizeof(double), hipMemcpyHostToDevice, ctx->stream));
  const double* yh = nullptr;
  if ((rc = eval_density(ctx, 0, 1, &yh)) != SI_OK) return rc;
  SI_HIP(ctx, hipMemcpyAsync(Yhat_out, yh, (size_t)ctx->out_dim * ctx->B * sizeof(double), hipMemcpyDeviceToHost, ctx->stream));
  SI_HIP(ctx, hipStreamSynchronize(ctx->stream));
  return SI_OK;
}

int32_t si_predict(si_ctx* ctx, const double* Z, int32_t C, const double* Xnew, int64_t Bn, double* Yhat_out) {
  CHECK_CTX(ctx);
  if (!ctx->i_ready) return fail(ctx, SI_ERR_STATE, "si_predict: call si_infer_setup first");
  if (!Z || C <= 0 || !Xnew || Bn <= 0 || !Yhat_out) return fail(ctx, SI_ERR_INVALID, "si_predict: bad argument");
  if (ctx->sw_Z) return fail(ctx, SI_ERR_STATE, "si_predict: a step-wise RWMH session is open; its proposal / SSE buffers are shared (si_rwmh_end or si_rwmh_abort first)");
  BIND(ctx);
  int32_t rc = ensure_chains(ctx, C);
  if (rc != SI_OK) return rc;
  SI_HIP(ctx, hipStreamSynchronize(ctx->stream));
  // run the ordinary forward path on temporary data buffers sized for Bn (the density's own X, Y stay untouched);
  // like the density, up to `wb` samples share one pass of launches (grid.y), within the same workspace cap
  struct Saved {
    double *X, *Y, *act0, *act1, *ssepart, *part, *yhat, *Xc;
    int64_t B, act_elems;
    int sse_blocks;
  } sv{ctx->d_X, ctx->d_Y, ctx->d_act[0], ctx->d_act[1], ctx->d_ssepart, ctx->d_part, ctx->d_yhat, ctx->d_Xc,
       ctx->B, ctx->act_elems, ctx->sse_blocks};
  // (with compute_dtype = SI_F32 the predictive forward still runs in fp64: it owns its fp64 workspace below, and the fp64
  //  weights are what K4 writes in either mode -- the fp32 option covers the density / the RWMH samplers)
  const bool f32_saved = ctx->f32;
  ctx->f32 = false;
  const int64_t act_elems = pad_ld(ctx->max_stored * Bn);   // (eval_density with yhat_out set never takes the fused launch)
  const int sse_blocks = sse_num_blocks((int64_t)ctx->out_dim * Bn, ctx->num_cu);
  const size_t dB = (size_t)ctx->out_dim * (size_t)Bn;
  const double per = 8.0 * (2.0 * (double)act_elems + ((double)ctx->fuse_slots + 1.0) * (double)dB + (double)sse_blocks);
  const size_t wb = (size_t)std::max(1.0, std::min({(double)C, (double)ctx->fw_slots, std::floor(SI_BATCH_BYTES / per)}));
  double *tX = nullptr, *tY = nullptr, *tA0 = nullptr, *tA1 = nullptr, *tS = nullptr, *tP = nullptr, *tYh = nullptr, *tXc = nullptr;
  bool ok = dev_alloc(&tX, (size_t)ctx->in_dim * Bn) == hipSuccess && dev_alloc(&tY, dB) == hipSuccess &&
            (!ctx->plan.input_spatial || dev_alloc(&tXc, (size_t)ctx->plan.in_elems * Bn) == hipSuccess) &&
            dev_alloc(&tA0, wb * (size_t)act_elems) == hipSuccess && dev_alloc(&tA1, wb * (size_t)act_elems) == hipSuccess &&
            dev_alloc(&tS, wb * (size_t)sse_blocks) == hipSuccess &&
            (!ctx->fuse_tail || (dev_alloc(&tP, wb * (size_t)ctx->fuse_slots * dB) == hipSuccess &&
                                 dev_alloc(&tYh, wb * dB) == hipSuccess));
  hipError_t e = hipSuccess;
  if (ok) {
    e = hipMemcpyAsync(tX, Xnew, (size_t)ctx->in_dim * Bn * sizeof(double), hipMemcpyHostToDevice, ctx->stream);
    if (e == hipSuccess) e = hipMemsetAsync(tY, 0, dB * sizeof(double), ctx->stream);
    if (e == hipSuccess)
      e = hipMemcpyAsync(ctx->d_zprop, Z, (size_t)ctx->iM * C * sizeof(double), hipMemcpyHostToDevice, ctx->stream);
    ctx->d_X = tX; ctx->d_Y = tY; ctx->d_act[0] = tA0; ctx->d_act[1] = tA1; ctx->d_ssepart = tS; ctx->d_part = tP;
    ctx->d_yhat = tYh; ctx->B = Bn; ctx->act_elems = act_elems; ctx->sse_blocks = sse_blocks; ctx->d_Xc = tXc;
    if (ctx->plan.input_spatial) net_input(ctx, ctx->plan, tX, tXc, Bn);
    for (int c0 = 0; c0 < C && e == hipSuccess && rc == SI_OK; c0 += (int)wb) {
      const int nc = std::min<int>((int)wb, C - c0);
      const double* yh = nullptr;
      rc = eval_density(ctx, c0, nc, &yh);
      if (rc == SI_OK) {
        // sample j of the batch: yh + j * (out_dim*Bn) after the fused tail, yh + j * act_elems otherwise
        const size_t src_pitch = (ctx->fuse_tail ? dB : (size_t)act_elems) * sizeof(double);
        e = hipMemcpy2DAsync(Yhat_out + (size_t)c0 * dB, dB * sizeof(double), yh, src_pitch, dB * sizeof(double), (size_t)nc,
                             hipMemcpyDeviceToHost, ctx->stream);
      }
      if (e == hipSuccess) e = hipStreamSynchronize(ctx->stream);  // the outputs are overwritten by the next batch
    }
    ctx->d_X = sv.X; ctx->d_Y = sv.Y; ctx->d_act[0] = sv.act0; ctx->d_act[1] = sv.act1; ctx->d_ssepart = sv.ssepart;
    ctx->d_part = sv.part; ctx->d_yhat = sv.yhat; ctx->B = sv.B; ctx->act_elems = sv.act_elems; ctx->sse_blocks = sv.sse_blocks;
    ctx->d_Xc = sv.Xc;
  }
  ctx->f32 = f32_saved;
  (void)hipStreamSynchronize(ctx->stream);
  dev_free(tX); dev_free(tY); dev_free(tA0); dev_free(tA1); dev_free(tS); dev_free(tP); dev_free(tYh); dev_free(tXc);
  if (!ok) return fail(ctx, SI_ERR_NOMEM, "si_predict: device allocation failed");
  if (rc != SI_OK) return rc;
  if (e != hipSuccess) return fail(ctx, SI_ERR_HIP, std::string("si_predict: ") + hipGetErrorString(e));
  return SI_OK;
}

}  // extern "C"
