// C ABI, sampling side: the RWMH samplers (si_sample_rwmh*, the step-wise session si_rwmh_*), the output map
// (si_sample_rwmh_weights, si_reconstruct) -- reference src/space_inference.jl:111-116,125.  Three sampler forms behind one
// entry point: the one-workgroup device-resident loop (kernels_chain.hip), the persistent grid loop (kernels_chain_grid.hip)
// and the launch-per-step loop.  Host-side orchestration only; no CPU fallback anywhere in this file.
#include <cstring>

#include "capi_common.h"
#include "chain_spec_args.h"
#include "chain_spec_rtc.h"

using namespace si;

extern "C" {

// ---- streamed output map (a13, src/space_inference.jl:125: `map(z -> W_swa + P*z.params, chm)`) ------------------------
// K4 has already produced W_swa + P z' for every proposal (d_w); the weight vector of sample t is that vector when the
// proposal was accepted and the previous sample's otherwise.  A select kernel keeps the CURRENT weights of every chain in
// a small device ring (8 bytes read + 8 written per weight, ~4 us at cfg2 -- instead of a second K4 pass over P), a DMA
// on the second stream moves ring slot t into pinned memory while transition t+1 computes, and the host copy pool
// moves it into the caller's array R-1 transitions later.  The chain never waits for PCIe.
static constexpr int SI_WRING = 4;

void free_wstream(si_ctx* ctx) {
  dev_free(ctx->d_wring);
  dev_free(ctx->d_accflag);
  for (int r = 0; r < SI_WRING; ++r) {
    if (ctx->h_wring[r]) (void)hipHostFree(ctx->h_wring[r]);
    ctx->h_wring[r] = nullptr;
    if (ctx->ev_wcomp[r]) (void)hipEventDestroy(ctx->ev_wcomp[r]);
    if (ctx->ev_wcopy[r]) (void)hipEventDestroy(ctx->ev_wcopy[r]);
    ctx->ev_wcomp[r] = ctx->ev_wcopy[r] = nullptr;
  }
  ctx->wring_N = 0;
  ctx->wring_C = 0;
}

static int32_t ensure_wstream(si_ctx* ctx, int32_t C) {
  const size_t need = (size_t)C * (size_t)pad_ld(ctx->iN);
  if (!ctx->stream2) SI_HIP(ctx, hipStreamCreateWithFlags(&ctx->stream2, hipStreamNonBlocking));
  if (ctx->wring_N == ctx->iN && ctx->wring_C >= C) return SI_OK;
  SI_HIP(ctx, hipStreamSynchronize(ctx->stream));
  free_wstream(ctx);
  bool ok = dev_alloc(&ctx->d_wring, need * SI_WRING) == hipSuccess && dev_alloc(&ctx->d_accflag, (size_t)C) == hipSuccess;
  for (int r = 0; r < SI_WRING && ok; ++r)
    ok = hipHostMalloc(reinterpret_cast<void**>(&ctx->h_wring[r]), (size_t)C * (size_t)ctx->iN * sizeof(double), hipHostMallocDefault) == hipSuccess &&
         hipEventCreateWithFlags(&ctx->ev_wcomp[r], hipEventDisableTiming) == hipSuccess &&
         hipEventCreateWithFlags(&ctx->ev_wcopy[r], hipEventDisableTiming) == hipSuccess;
  if (!ok) {
    free_wstream(ctx);
    return fail(ctx, SI_ERR_NOMEM, "si_sample_rwmh_weights: allocation of the weight ring / pinned staging failed");
  }
  ctx->wring_N = ctx->iN;
  ctx->wring_C = C;
  return SI_OK;
}

// the device-resident loop covers: Dense chains in fp64 with the head folded into the layer before it (fuse_tail), the
// four activations the MFMA epilogues carry, no prior term, and little enough arithmetic that ONE CU per chain beats ~8
// launches per transition spread over the chip (2 N B <= 3 MFLOP: the README toy is 0.14)
static constexpr size_t SI_CHAIN_LDS_LIMIT = 160 * 1024 - 256;
static bool chain_loop_applies(const si_ctx* ctx) {
  if (ctx->f32 || ctx->plan.has_conv || !ctx->fuse_tail || ctx->sigma_p > 0.0) return false;
  if (ctx->layers.size() > (size_t)SI_CHAIN_MAX_LAYERS || ctx->iN > (1 << 20) || ctx->B > (1 << 20)) return false;
  if (ctx->iM > 1024) return false;   // (rwmh_chain_kernel keeps z one element per thread of its 1024-thread workgroup)
  for (const auto& ly : ctx->layers)
    if (ly.kind != SI_LAYER_DENSE || ly.act >= SI_ACT_LEAKYRELU) return false;
  return 2.0 * (double)ctx->iN * (double)ctx->B <= 3.0e6;
}

static int32_t sample_rwmh_impl(si_ctx* ctx, const char* who, int64_t itr, double sigma_z, uint64_t seed, int32_t chain_id0,
                                int32_t nchains, double* Z_out, double* lp_out, double* accept_rate_out, double* W_out) {
  CHECK_CTX(ctx);
  ctx->last_density_spec = ctx->last_loop_spec = 0;   // (si_chain_kernel_info reports THIS call)
  if (!ctx->i_ready) return fail(ctx, SI_ERR_STATE, std::string(who) + ": call si_infer_setup first");
  if (itr <= 0 || nchains <= 0 || chain_id0 < 0 || !(sigma_z > 0.0))
    return fail(ctx, SI_ERR_INVALID, std::string(who) + ": itr, nchains, sigma_z must be positive");
  if (ctx->sw_Z) return fail(ctx, SI_ERR_STATE, std::string(who) + ": a step-wise RWMH session is open; its proposal / SSE buffers are shared (si_rwmh_end or si_rwmh_abort first)");
  BIND(ctx);
  const int32_t C = nchains, M = ctx->iM;
  const int64_t N = ctx->iN, ldw = pad_ld(N);
  int32_t rc = ensure_chains(ctx, C);
  if (rc != SI_OK) return rc;
  if (W_out && (rc = ensure_wstream(ctx, C)) != SI_OK) return rc;
  // the device-side output arrays stay with the ctx (grown on demand, released with the inference set-up)
  {
    const size_t needZ = (size_t)M * itr * C, needlp = (size_t)itr * C;
    if (ctx->outZ_cap < needZ) {
      dev_free(ctx->d_outZ);
      ctx->outZ_cap = 0;
      if (dev_alloc(&ctx->d_outZ, needZ) != hipSuccess) return fail(ctx, SI_ERR_NOMEM, std::string(who) + ": output allocation failed");
      ctx->outZ_cap = needZ;
    }
    if (ctx->outlp_cap < needlp) {
      dev_free(ctx->d_outlp);
      ctx->outlp_cap = 0;
      if (dev_alloc(&ctx->d_outlp, needlp) != hipSuccess) return fail(ctx, SI_ERR_NOMEM, std::string(who) + ": output allocation failed");
      ctx->outlp_cap = needlp;
    }
  }
  double* const dZ = ctx->d_outZ;
  double* const dlp = ctx->d_outlp;
  const double d = (double)ctx->out_dim * (double)ctx->B;
  const double c0 = mvnormal_c0(d, ctx->sigma_m), s2 = ctx->sigma_m * ctx->sigma_m;
  // ---- K6 as a device-resident loop (kernels_chain.hip): small Dense chains whose weights, data and activations fit one
  // workgroup's LDS run ALL transitions in one launch, one workgroup per chain -- the launch-per-step loop below costs ~8
  // dependent launches (25 us) per transition whatever the size.  Same bits (tests/test_gpu_chain.py).
  // (with the output map requested -- what the drop-in sub_inference call does -- the weight samples of the finished chains
  //  come from ONE K4 pass over all itr * C samples, the kernel si_reconstruct runs: same bits, no streaming needed at this size)
  const size_t wall_elems = (size_t)ldw * (size_t)itr * (size_t)C;
  if ((!W_out || wall_elems <= ((size_t)512 << 20) / sizeof(double)) && chain_loop_applies(ctx) && ctx->chain_mode == 1) {
    ChainLoopArgs a{};
    const int L = (int)ctx->layers.size();
    for (int l = 0; l < L; ++l) a.lay[l] = ctx->layers[(size_t)l];
    a.swa = ctx->i_swa; a.P = ctx->i_P; a.X = ctx->d_X; a.Y = ctx->d_Y;
    a.Z_out = dZ; a.lp_out = dlp; a.nacc_out = ctx->d_nacc;
    a.ldP = ctx->ldP; a.itr = itr; a.seed = seed; a.sigma_z = sigma_z; a.c0 = c0; a.sigma2 = s2;
    a.N = (int)N; a.M = M; a.B = (int)ctx->B; a.L = L; a.chain_id0 = chain_id0;
    a.slot_feats = dense_fused_slot_feats(ctx->layers[(size_t)L - 2].out);
    a.fuse_slots = ctx->fuse_slots;
    const size_t lds = chain_loop_plan(a, SI_CHAIN_LDS_LIMIT);
    if (lds != 0) {
      {
        const double fl = 2.0 * (double)N * (double)ctx->B * (double)itr * C;
        ProfScope ps(ctx, SI_K_RWMH, fl, 0.0);
        launch_chain_loop(ctx->stream, a, C, lds);
      }
      hipError_t e = hipGetLastError();
      std::vector<int64_t> nacc((size_t)C);
      if (e == hipSuccess && Z_out) e = hipMemcpyAsync(Z_out, dZ, (size_t)M * itr * C * sizeof(double), hipMemcpyDeviceToHost, ctx->stream);
      if (e == hipSuccess && lp_out) e = hipMemcpyAsync(lp_out, dlp, (size_t)itr * C * sizeof(double), hipMemcpyDeviceToHost, ctx->stream);
      if (e == hipSuccess) e = hipMemcpyAsync(nacc.data(), ctx->d_nacc, (size_t)C * sizeof(int64_t), hipMemcpyDeviceToHost, ctx->stream);
      double* dW = nullptr;
      if (e == hipSuccess && W_out) {   // src/space_inference.jl:125 for every sample of every chain
        if (dev_alloc(&dW, wall_elems) != hipSuccess) e = hipErrorOutOfMemory;
        if (e == hipSuccess) {
          ProfScope ps(ctx, SI_K_RECON, 2.0 * (double)N * M * (double)itr * C, (double)N * (M + 1 + (double)itr * C) * 8.0);
          launch_reconstruct(ctx->stream, ctx->i_swa, ctx->i_P, ctx->ldP, N, M, dZ, (int32_t)(itr * C), dW, ldw, ctx->num_cu);
          e = hipGetLastError();
        }
        if (e == hipSuccess)
          e = hipMemcpy2DAsync(W_out, (size_t)N * sizeof(double), dW, (size_t)ldw * sizeof(double), (size_t)N * sizeof(double),
                               (size_t)itr * C, hipMemcpyDeviceToHost, ctx->stream);
      }
      const hipError_t e2 = hipStreamSynchronize(ctx->stream);
      dev_free(dW);
      if (e != hipSuccess) return fail(ctx, SI_ERR_HIP, std::string(who) + ": " + hipGetErrorString(e));
      if (e2 != hipSuccess) return fail(ctx, SI_ERR_HIP, std::string(who) + ": " + hipGetErrorString(e2));
      if (accept_rate_out)
        for (int c = 0; c < C; ++c) accept_rate_out[c] = itr > 1 ? (double)nacc[(size_t)c] / (double)(itr - 1) : 0.0;
      return SI_OK;
    }
  }
  // ---- K6 as a persistent loop over a GRID of workgroups (kernels_chain_grid.hip): a narrow chain too large for one
  // workgroup (docs/src/nn_example.md's MLP) runs all transitions in one launch, G = ceil(B / tile) resident workgroups per
  // chain, two bounded grid barriers per transition.  Same bits as the loop below (tests/test_gpu_chain_grid.py).
  if ((!W_out || wall_elems <= ((size_t)512 << 20) / sizeof(double)) && ctx->fused_ok && ctx->chain_mode == 1 && !(ctx->sigma_p > 0.0) &&
      C <= ctx->fw_slots && itr < ((int64_t)1 << 24)) {
    ChainGridArgs a{};
    const int L = (int)ctx->layers.size();
    const int sf = ctx->fuse_tail ? dense_fused_slot_feats(ctx->layers[(size_t)L - 2].out) : 0;
    int nb = 0;
    size_t lds = 0;
    for (int cand : {1, 2, 4}) {   // the smallest tile whose grid is resident: one workgroup per CU
      const int64_t G = (ctx->B + 16 * cand - 1) / (16 * cand);
      if (G * C > ctx->num_cu) continue;
      const size_t lf = chain_fused_plan(a.p, ctx->layers.data(), L, ctx->B, cand, ctx->fuse_tail, sf, ctx->fuse_slots);
      fused_fill_program(ctx, a.p);
      a.M = M;
      a.nblocks = ctx->sse_blocks;
      a.G = (int)G;
      lds = chain_grid_plan(a, lf);
      if (lds != 0) {
        nb = cand;
        break;
      }
    }
    // The loop specialised to this chain's shapes (chain_spec.inc through hiprtc; same bits): ONE barrier per transition, the
    // weights handed over in fragment order.  Class: a narrow head, at most 1024 model outputs (four blocks of the SSE tree),
    // every weight fragment in registers (spec_applies).
    const SpecKernels* sk = nullptr;
    SiSpecGridArgs sa{};
    size_t lds_spec = 0;
    if (nb != 0 && ctx->chain_spec && ctx->fuse_tail && (int64_t)ctx->out_dim * ctx->B <= 1024 && M <= 256) {
      const int G = a.G;
      const int rs = (int)((((N + G - 1) / G) + 15) & ~(int64_t)15);
      sk = spec_kernels(ctx->layers.data(), L, nb, sf, M, rs <= 256 && M <= 32, &ctx->spec_message);
      if (sk) {
        auto even = [](int64_t v) { return (v + 1) & ~(int64_t)1; };
        const int64_t d = (int64_t)ctx->out_dim * ctx->B;
        int64_t off = sk->lds_doubles;
        sa.y_in_lds = 1;
        sa.o_y = (int)off;
        off += even(d);
        sa.o_blk = (int)off;
        off += even(ctx->sse_blocks);
        sa.o_z = (int)off;
        off += even(5 * (int64_t)M);
        sa.o_red = (int)off;
        off += 24;
        sa.o_flag = (int)off;
        off += 2;
        lds_spec = std::max((size_t)off * sizeof(double), (size_t)(81 * 1024));   // (more than half a CU's LDS: one workgroup per CU)
        if (lds_spec > (size_t)160 * 1024 - 256) sk = nullptr;
      }
      if (sk && (ctx->spec_chains < C || ctx->spec_fo != sk->fo_total)) {
        dev_free(ctx->d_specw);
        dev_free(ctx->d_specy);
        ctx->spec_chains = 0;
        if (dev_alloc(&ctx->d_specw, (size_t)4 * (size_t)sk->fo_total * (size_t)C) != hipSuccess ||
            dev_alloc(&ctx->d_specy, (size_t)2 * (size_t)ctx->out_dim * (size_t)ctx->B * (size_t)C) != hipSuccess)
          return fail(ctx, SI_ERR_NOMEM, std::string(who) + ": allocation failed");
        // (the padding elements of the fragment-ordered vectors are never written by K4: they must be finite)
        if (hipMemsetAsync(ctx->d_specw, 0, (size_t)4 * (size_t)sk->fo_total * (size_t)C * sizeof(double), ctx->stream) != hipSuccess)
          return fail(ctx, SI_ERR_HIP, std::string(who) + ": hipMemsetAsync failed");
        ctx->spec_chains = C;
        ctx->spec_fo = sk->fo_total;
      }
      if (sk && ctx->specperm_for != (const void*)sk) {
        dev_free(ctx->d_specperm);
        ctx->specperm_for = nullptr;
        if (dev_alloc(&ctx->d_specperm, (size_t)N) != hipSuccess) return fail(ctx, SI_ERR_NOMEM, std::string(who) + ": allocation failed");
        int* pp = ctx->d_specperm;
        void* pargs[] = {&pp};
        if (hipModuleLaunchKernel(sk->perm, (unsigned)((sk->max_wn + 255) / 256), 1, 1, 256, 1, 1, 0, ctx->stream, pargs, nullptr) != hipSuccess)
          return fail(ctx, SI_ERR_HIP, std::string(who) + ": launch of the fragment-order permutation failed");
        ctx->specperm_for = (const void*)sk;
      }
    }
    const size_t sync_lines = sk ? (size_t)8 * (size_t)C + 1 : (size_t)C + 1;   // (the specialised loop shards each chain's counter over 8 lines)
    if (nb != 0) {
      if ((size_t)ctx->gridsync_chains < sync_lines) {
        dev_free(ctx->d_gridsync);
        ctx->gridsync_chains = 0;
        if (dev_alloc(&ctx->d_gridsync, (size_t)32 * sync_lines) != hipSuccess) return fail(ctx, SI_ERR_NOMEM, std::string(who) + ": allocation failed");
        ctx->gridsync_chains = (int)sync_lines;
      }
      a.swa = ctx->i_swa; a.P = ctx->i_P; a.X = ctx->d_X; a.Y = ctx->d_Y;
      a.wbuf = ctx->d_w; a.w_stride = ldw;
      a.ybuf = ctx->d_yhat; a.y_stride = (int64_t)ctx->out_dim * ctx->B;
      a.cnt = ctx->d_gridsync; a.status = ctx->d_gridsync + (size_t)32 * (sync_lines - 1);
      a.Z_out = dZ; a.lp_out = dlp; a.nacc_out = ctx->d_nacc;
      a.ldP = ctx->ldP; a.itr = itr; a.seed = seed; a.sigma_z = sigma_z; a.c0 = c0; a.sigma2 = s2;
      a.N = (int)N; a.chain_id0 = chain_id0;
      hipError_t e = hipMemsetAsync(ctx->d_gridsync, 0, (size_t)32 * sync_lines * sizeof(unsigned), ctx->stream);
      if (e == hipSuccess) {
        const double fl = 2.0 * (double)N * (double)ctx->B * (double)itr * C;
        ProfScope ps(ctx, SI_K_RWMH, fl, 0.0);
        if (sk) {
          sa.swa = a.swa; sa.P = a.P; sa.X = a.X; sa.Y = a.Y; sa.perm = ctx->d_specperm;
          sa.wbuf = ctx->d_specw; sa.w_stride = sk->fo_total;
          sa.ybuf = ctx->d_specy; sa.y_stride = a.y_stride;
          sa.cnt = a.cnt; sa.status = a.status;
          sa.Z_out = dZ; sa.lp_out = dlp; sa.nacc_out = (long long*)ctx->d_nacc;
          sa.ldP = a.ldP; sa.itr = itr; sa.seed = seed; sa.sigma_z = sigma_z; sa.c0 = c0; sa.sigma2 = s2;
          sa.N = (int)N; sa.M = M; sa.G = a.G; sa.B = (int)ctx->B; sa.chain_id0 = chain_id0; sa.nblocks = ctx->sse_blocks;
          void* gargs[] = {&sa};
          e = hipModuleLaunchKernel(sk->grid, (unsigned)(a.G * C), 1, 1, 256, 1, 1, (unsigned)lds_spec, ctx->stream, gargs, nullptr);
          ctx->last_loop_spec = e == hipSuccess;
        } else {
          e = launch_chain_grid(ctx->stream, a, nb, C, lds);
        }
      }
      std::vector<int64_t> nacc((size_t)C);
      unsigned status = 0;
      if (e == hipSuccess) e = hipMemcpyAsync(&status, a.status, sizeof(unsigned), hipMemcpyDeviceToHost, ctx->stream);
      // the samples and lp go through pinned staging: a device-to-host copy straight into the caller's array pins its pages on the
      // fly, and for an array nobody has touched yet (np.empty, Matrix{Float64}(undef, ...)) that costs ~9 us per page -- 7 ms for the
      // 3.4 MB of a 20 000-transition chain, 0.35 us per transition of a 9.8 us loop
      const size_t zbytes = Z_out ? (size_t)M * itr * C * sizeof(double) : 0, lbytes = lp_out ? (size_t)itr * C * sizeof(double) : 0;
      bool staged = zbytes + lbytes >= ((size_t)128 << 10) && zbytes + lbytes <= ((size_t)128 << 20);
      if (staged && ctx->h_outpin_cap < zbytes + lbytes) {
        if (ctx->h_outpin) (void)hipHostFree(ctx->h_outpin);
        ctx->h_outpin = nullptr;
        ctx->h_outpin_cap = 0;
        if (hipHostMalloc(reinterpret_cast<void**>(&ctx->h_outpin), zbytes + lbytes, hipHostMallocDefault) == hipSuccess)
          ctx->h_outpin_cap = zbytes + lbytes;
        else
          staged = false;   // (the direct copy is always possible)
      }
      if (e == hipSuccess && Z_out) e = hipMemcpyAsync(staged ? (void*)ctx->h_outpin : (void*)Z_out, dZ, zbytes, hipMemcpyDeviceToHost, ctx->stream);
      if (e == hipSuccess && lp_out)
        e = hipMemcpyAsync(staged ? (void*)(ctx->h_outpin + zbytes) : (void*)lp_out, dlp, lbytes, hipMemcpyDeviceToHost, ctx->stream);
      if (e == hipSuccess) e = hipMemcpyAsync(nacc.data(), ctx->d_nacc, (size_t)C * sizeof(int64_t), hipMemcpyDeviceToHost, ctx->stream);
      double* dW = nullptr;
      if (e == hipSuccess && W_out) {   // src/space_inference.jl:125 for every sample of every chain (one K4 pass, as above)
        if (dev_alloc(&dW, wall_elems) != hipSuccess) e = hipErrorOutOfMemory;
        if (e == hipSuccess) {
          ProfScope ps(ctx, SI_K_RECON, 2.0 * (double)N * M * (double)itr * C, (double)N * (M + 1 + (double)itr * C) * 8.0);
          launch_reconstruct(ctx->stream, ctx->i_swa, ctx->i_P, ctx->ldP, N, M, dZ, (int32_t)(itr * C), dW, ldw, ctx->num_cu);
          e = hipGetLastError();
        }
        if (e == hipSuccess)
          e = hipMemcpy2DAsync(W_out, (size_t)N * sizeof(double), dW, (size_t)ldw * sizeof(double), (size_t)N * sizeof(double),
                               (size_t)itr * C, hipMemcpyDeviceToHost, ctx->stream);
      }
      const hipError_t e2 = hipStreamSynchronize(ctx->stream);
      dev_free(dW);
      if (e != hipSuccess) return fail(ctx, SI_ERR_HIP, std::string(who) + ": " + hipGetErrorString(e));
      if (e2 != hipSuccess) return fail(ctx, SI_ERR_HIP, std::string(who) + ": " + hipGetErrorString(e2));
      if (staged) {
        if (Z_out) std::memcpy(Z_out, ctx->h_outpin, zbytes);
        if (lp_out) std::memcpy(lp_out, ctx->h_outpin + zbytes, lbytes);
      }
      if (status != 0)
        return fail(ctx, SI_ERR_HIP, std::string(who) + ": the grid barrier of the device-resident loop timed out (its workgroups were not all resident: "
                                     "is another process holding compute units of this GPU?); si_set_chain_loop(ctx, 2) runs the launch-per-step loop");
      if (accept_rate_out)
        for (int c = 0; c < C; ++c) accept_rate_out[c] = itr > 1 ? (double)nacc[(size_t)c] / (double)(itr - 1) : 0.0;
      return SI_OK;
    }
  }
  {
    ProfScope ps(ctx, SI_K_RWMH, 0, 0);
    launch_rwmh_init(ctx->stream, ctx->d_zcur, ctx->d_lpcur, ctx->d_nacc, ctx->d_steps, M, C);
  }
  // the proposal weights of ALL chains are still in d_w at accept time only when one pass of launches carries them all
  const bool select_path = W_out && C <= ctx->fw_slots;
  // one transition for all chains; the transition index is a device-side counter, so the launches are identical.
  // When one pass of launches carries all chains (and the prior term is off) the tail of a transition is ONE launch: the last
  // stage of the SSE reduction, the accept step and the next transition's proposal (rwmh_tail_kernel) -- same functions, same
  // order, same bits; otherwise sse_final / accept / propose stay separate kernels.
  const bool fused_tail = C <= ctx->fw_slots && !(ctx->sigma_p > 0.0) && ctx->chain_loop_enabled;
  int64_t tdone = 0;
  auto transition = [&]() -> int32_t {
    if (!fused_tail || tdone == 0) {
      ProfScope ps(ctx, SI_K_RWMH, 0, 0);
      launch_rwmh_propose(ctx->stream, ctx->d_zcur, ctx->d_zprop, M, C, sigma_z, seed, chain_id0, ctx->d_steps);
    }
    ctx->defer_sse_final = fused_tail;
    const int32_t r = eval_density_all(ctx, C);
    ctx->defer_sse_final = false;
    if (r != SI_OK) return r;
    ProfScope ps(ctx, SI_K_RWMH, 0, 0);
    if (fused_tail)
      launch_rwmh_tail(ctx->stream, ctx->d_ssepart, ctx->sse_blocks, ctx->d_sse, ctx->d_zcur, ctx->d_zprop, ctx->d_lpcur, ctx->d_nacc, M,
                       C, c0, s2, sigma_z, seed, chain_id0, ctx->d_steps, dZ, dlp, itr, select_path ? ctx->d_accflag : nullptr,
                       tdone + 1 < itr);
    else
      launch_rwmh_accept(ctx->stream, ctx->d_zcur, ctx->d_zprop, ctx->d_lpcur, ctx->d_sse, ctx->d_nacc, M, C, c0, s2, seed,
                         chain_id0, ctx->d_steps, dZ, dlp, itr, ctx->sigma_p > 0.0 ? ctx->d_wsq : nullptr, prior_c0(ctx),
                         ctx->sigma_p * ctx->sigma_p, select_path ? ctx->d_accflag : nullptr);
    ++tdone;
    return SI_OK;
  };
  hipError_t e = hipSuccess;
  auto ring = [&](int64_t t) { return ctx->d_wring + (size_t)(t % SI_WRING) * (size_t)C * (size_t)ldw; };
  auto drain = [&](int64_t u) {   // sample u of every chain: pinned slot -> the caller's (pageable) N x itr x C array
    const int r = (int)(u % SI_WRING);
    hipError_t w = hipEventSynchronize(ctx->ev_wcopy[r]);
    for (int c = 0; c < C && w == hipSuccess; ++c)
      host_copy(W_out + (size_t)N * ((size_t)u + (size_t)itr * c), ctx->h_wring[r] + (size_t)c * N, (size_t)N * sizeof(double));
    return w;
  };
  // Replaying one captured transition as a hipGraph was measured and dropped: the README-toy transition takes 27.8 us
  // graphed vs 25.7 us eager -- it is bound by the serial latency of its 8 dependent small kernels, not by host
  // launches -- and at cfg2 a transition is 3.3 ms of kernel time.
  for (int64_t t = 0; t < itr && rc == SI_OK && e == hipSuccess; ++t) {
    rc = transition();
    if (rc != SI_OK || !W_out) continue;
    const int r = (int)(t % SI_WRING);
    if (t >= SI_WRING) e = hipStreamWaitEvent(ctx->stream, ctx->ev_wcopy[r], 0);  // the DMA of sample t - R has read this slot
    if (e != hipSuccess) break;
    {
      ProfScope ps(ctx, SI_K_RECON, 0.0, 16.0 * (double)N * C);
      if (select_path)
        launch_weights_select(ctx->stream, ctx->d_accflag, ctx->d_w, ldw, t > 0 ? ring(t - 1) : nullptr, ring(t), ldw, N, C, ctx->num_cu);
      else  // more chains than one pass of launches carries: K4 on the current states (same kernel, same bits)
        launch_reconstruct(ctx->stream, ctx->i_swa, ctx->i_P, ctx->ldP, N, M, ctx->d_zcur, C, ring(t), ldw, ctx->num_cu);
    }
    e = hipEventRecord(ctx->ev_wcomp[r], ctx->stream);
    if (e == hipSuccess) e = hipStreamWaitEvent(ctx->stream2, ctx->ev_wcomp[r], 0);
    if (e == hipSuccess)
      e = hipMemcpy2DAsync(ctx->h_wring[r], (size_t)N * sizeof(double), ring(t), (size_t)ldw * sizeof(double), (size_t)N * sizeof(double),
                           (size_t)C, hipMemcpyDeviceToHost, ctx->stream2);
    if (e == hipSuccess) e = hipEventRecord(ctx->ev_wcopy[r], ctx->stream2);
    if (e == hipSuccess && t >= SI_WRING - 1) e = drain(t - (SI_WRING - 1));   // frees the pinned slot sample t + 1 will use
  }
  if (rc == SI_OK && e == hipSuccess && W_out)
    for (int64_t u = std::max<int64_t>(0, itr - (SI_WRING - 1)); u < itr && e == hipSuccess; ++u) e = drain(u);
  if (W_out) (void)hipStreamSynchronize(ctx->stream2);
  if (rc != SI_OK || e != hipSuccess) {
    (void)hipStreamSynchronize(ctx->stream);
    if (rc != SI_OK) return rc;
    return fail(ctx, SI_ERR_HIP, std::string(who) + ": " + hipGetErrorString(e));
  }
  e = hipGetLastError();
  std::vector<int64_t> nacc((size_t)C);
  if (e == hipSuccess && Z_out)
    e = hipMemcpyAsync(Z_out, dZ, (size_t)M * itr * C * sizeof(double), hipMemcpyDeviceToHost, ctx->stream);
  if (e == hipSuccess && lp_out)
    e = hipMemcpyAsync(lp_out, dlp, (size_t)itr * C * sizeof(double), hipMemcpyDeviceToHost, ctx->stream);
  if (e == hipSuccess)
    e = hipMemcpyAsync(nacc.data(), ctx->d_nacc, (size_t)C * sizeof(int64_t), hipMemcpyDeviceToHost, ctx->stream);
  hipError_t e2 = hipStreamSynchronize(ctx->stream);
  if (e != hipSuccess) return fail(ctx, SI_ERR_HIP, std::string(who) + ": " + hipGetErrorString(e));
  if (e2 != hipSuccess) return fail(ctx, SI_ERR_HIP, std::string(who) + ": " + hipGetErrorString(e2));
  if (accept_rate_out)
    for (int c = 0; c < C; ++c) accept_rate_out[c] = itr > 1 ? (double)nacc[(size_t)c] / (double)(itr - 1) : 0.0;
  return SI_OK;
}

int32_t si_sample_rwmh(si_ctx* ctx, int64_t itr, double sigma_z, uint64_t seed, int32_t chain_id0, int32_t nchains,
                       double* Z_out, double* lp_out, double* accept_rate_out) {
  return sample_rwmh_impl(ctx, "si_sample_rwmh", itr, sigma_z, seed, chain_id0, nchains, Z_out, lp_out, accept_rate_out, nullptr);
}

int32_t si_sample_rwmh_weights(si_ctx* ctx, int64_t itr, double sigma_z, uint64_t seed, int32_t chain_id0, int32_t nchains,
                               double* Z_out, double* lp_out, double* accept_rate_out, double* W_out) {
  if (ctx && !W_out) return fail(ctx, SI_ERR_INVALID, "si_sample_rwmh_weights: W_out is NULL (use si_sample_rwmh)");
  return sample_rwmh_impl(ctx, "si_sample_rwmh_weights", itr, sigma_z, seed, chain_id0, nchains, Z_out, lp_out, accept_rate_out, W_out);
}

// ---- step-wise RWMH: the same chain as si_sample_rwmh, but the SSE of every proposal passes through the caller
// between evaluation and acceptance, so that a DATA-SHARDED density (each rank holds B/world observations of X, Y and
// the same W_swa, P) can all-reduce the per-rank partial sums.  Every rank draws the same Philox
// stream (same seed / chain ids), so all ranks take identical accept decisions and keep identical chains.
int32_t si_set_chain_loop(si_ctx* ctx, int32_t on) {
  CHECK_CTX(ctx);
  if (on < 0 || on > 4)
    return fail(ctx, SI_ERR_INVALID, "si_set_chain_loop: 0 (one launch per layer and step), 1 (automatic), 2 (fused launches, no device-resident loop), "
                                     "3 / 4 (1 / 2 without the run-time specialised kernels)");
  ctx->chain_mode = on >= 3 ? on - 2 : on;
  ctx->chain_spec = on < 3;
  ctx->chain_loop_enabled = on != 0;
  return SI_OK;
}

int32_t si_chain_kernel_info(si_ctx* ctx, int32_t* density_specialised, int32_t* loop_specialised) {
  CHECK_CTX(ctx);
  if (density_specialised) *density_specialised = ctx->last_density_spec;
  if (loop_specialised) *loop_specialised = ctx->last_loop_spec;
  return SI_OK;
}

const char* si_chain_spec_message(si_ctx* ctx) { return ctx ? ctx->spec_message.c_str() : ""; }

int32_t si_rwmh_begin(si_ctx* ctx, int64_t itr, double sigma_z, uint64_t seed, int32_t chain_id0, int32_t nchains,
                      int64_t d_total) {
  CHECK_CTX(ctx);
  if (!ctx->i_ready) return fail(ctx, SI_ERR_STATE, "si_rwmh_begin: call si_infer_setup first");
  if (itr <= 0 || nchains <= 0 || chain_id0 < 0 || !(sigma_z > 0.0) || d_total < 0)
    return fail(ctx, SI_ERR_INVALID, "si_rwmh_begin: itr, nchains, sigma_z must be positive");
  BIND(ctx);
  SI_HIP(ctx, hipStreamSynchronize(ctx->stream));
  int32_t rc = ensure_chains(ctx, nchains);
  if (rc != SI_OK) return rc;
  dev_free(ctx->sw_Z);
  dev_free(ctx->sw_lp);
  if (dev_alloc(&ctx->sw_Z, (size_t)ctx->iM * itr * nchains) != hipSuccess ||
      dev_alloc(&ctx->sw_lp, (size_t)itr * nchains) != hipSuccess) {
    dev_free(ctx->sw_Z);
    dev_free(ctx->sw_lp);
    return fail(ctx, SI_ERR_NOMEM, "si_rwmh_begin: output allocation failed");
  }
  ctx->sw_itr = itr; ctx->sw_sigma_z = sigma_z; ctx->sw_seed = seed; ctx->sw_chain0 = chain_id0; ctx->sw_C = nchains;
  ctx->sw_d = d_total > 0 ? (double)d_total : (double)ctx->out_dim * (double)ctx->B;
  ctx->sw_next = 0;
  ctx->sw_evaluated = false;
  launch_rwmh_init(ctx->stream, ctx->d_zcur, ctx->d_lpcur, ctx->d_nacc, ctx->d_steps, ctx->iM, nchains);
  SI_HIP(ctx, hipGetLastError());
  return SI_OK;
}

int32_t si_rwmh_step_eval(si_ctx* ctx, double* sse_local_out) {
  CHECK_CTX(ctx);
  if (!ctx->sw_Z || ctx->sw_next >= ctx->sw_itr || ctx->sw_evaluated)
    return fail(ctx, SI_ERR_STATE, "si_rwmh_step_eval: call si_rwmh_begin first / accept the pending step / chain finished");
  BIND(ctx);
  const int32_t C = ctx->sw_C;
  launch_rwmh_propose(ctx->stream, ctx->d_zcur, ctx->d_zprop, ctx->iM, C, ctx->sw_sigma_z, ctx->sw_seed, ctx->sw_chain0,
                      ctx->d_steps);
  {
    const int32_t rc = eval_density_all(ctx, C);
    if (rc != SI_OK) return rc;
  }
  if (sse_local_out) {  // NULL: the partial sums stay on the device (si_rwmh_sse_ptr) -- no copy, no synchronisation
    SI_HIP(ctx, hipMemcpyAsync(sse_local_out, ctx->d_sse, (size_t)C * sizeof(double), hipMemcpyDeviceToHost, ctx->stream));
    SI_HIP(ctx, hipStreamSynchronize(ctx->stream));
  }
  ctx->sw_evaluated = true;
  return SI_OK;
}

int32_t si_rwmh_step_accept(si_ctx* ctx, const double* sse_total) {
  CHECK_CTX(ctx);
  if (!ctx->sw_Z || !ctx->sw_evaluated) return fail(ctx, SI_ERR_STATE, "si_rwmh_step_accept: no evaluated step pending");
  BIND(ctx);
  const int32_t C = ctx->sw_C;
  if (sse_total) {  // NULL: the device buffer of si_rwmh_sse_ptr already holds the totals (all-reduced in place)
    SI_HIP(ctx, hipMemcpyAsync(ctx->d_sse, sse_total, (size_t)C * sizeof(double), hipMemcpyHostToDevice, ctx->stream));
    SI_HIP(ctx, hipStreamSynchronize(ctx->stream));  // sse_total is caller-owned
  }
  const double c0 = mvnormal_c0(ctx->sw_d, ctx->sigma_m), s2 = ctx->sigma_m * ctx->sigma_m;
  launch_rwmh_accept(ctx->stream, ctx->d_zcur, ctx->d_zprop, ctx->d_lpcur, ctx->d_sse, ctx->d_nacc, ctx->iM, C, c0, s2,
                     ctx->sw_seed, ctx->sw_chain0, ctx->d_steps, ctx->sw_Z, ctx->sw_lp, ctx->sw_itr,
                     ctx->sigma_p > 0.0 ? ctx->d_wsq : nullptr, prior_c0(ctx), ctx->sigma_p * ctx->sigma_p);
  SI_HIP(ctx, hipGetLastError());
  ctx->sw_next += 1;
  ctx->sw_evaluated = false;
  return SI_OK;
}

int32_t si_rwmh_sse_ptr(si_ctx* ctx, double** sse_dev_out, int32_t* nchains_out) {
  CHECK_CTX(ctx);
  if (!ctx->sw_Z) return fail(ctx, SI_ERR_STATE, "si_rwmh_sse_ptr: call si_rwmh_begin first");
  if (sse_dev_out) *sse_dev_out = ctx->d_sse;
  if (nchains_out) *nchains_out = ctx->sw_C;
  return SI_OK;
}

int32_t si_rwmh_abort(si_ctx* ctx) {
  CHECK_CTX(ctx);
  BIND(ctx);
  SI_HIP(ctx, hipStreamSynchronize(ctx->stream));
  dev_free(ctx->sw_Z);
  dev_free(ctx->sw_lp);
  ctx->sw_evaluated = false;
  ctx->sw_next = ctx->sw_itr = 0;
  return SI_OK;
}

int32_t si_rwmh_end(si_ctx* ctx, double* Z_out, double* lp_out, double* accept_rate_out) {
  CHECK_CTX(ctx);
  if (!ctx->sw_Z || ctx->sw_next != ctx->sw_itr || ctx->sw_evaluated)
    return fail(ctx, SI_ERR_STATE, "si_rwmh_end: the chain is not complete");
  BIND(ctx);
  const int32_t C = ctx->sw_C, M = ctx->iM;
  const int64_t itr = ctx->sw_itr;
  std::vector<int64_t> nacc((size_t)C);
  if (Z_out) SI_HIP(ctx, hipMemcpyAsync(Z_out, ctx->sw_Z, (size_t)M * itr * C * sizeof(double), hipMemcpyDeviceToHost, ctx->stream));
  if (lp_out) SI_HIP(ctx, hipMemcpyAsync(lp_out, ctx->sw_lp, (size_t)itr * C * sizeof(double), hipMemcpyDeviceToHost, ctx->stream));
  SI_HIP(ctx, hipMemcpyAsync(nacc.data(), ctx->d_nacc, (size_t)C * sizeof(int64_t), hipMemcpyDeviceToHost, ctx->stream));
  SI_HIP(ctx, hipStreamSynchronize(ctx->stream));
  dev_free(ctx->sw_Z);
  dev_free(ctx->sw_lp);
  if (accept_rate_out)
    for (int c = 0; c < C; ++c) accept_rate_out[c] = itr > 1 ? (double)nacc[(size_t)c] / (double)(itr - 1) : 0.0;
  return SI_OK;
}

// Output map a13 (space_inference.jl:125: one W_swa + P z per sample, itr x N doubles on the host -- 8.4 GB at cfg2).
// A three-stage pipeline so that the PCIe link, not a single host thread, sets the pace: K4 writes a group of samples
// into one of two device buffers (compute stream) -> DMA into one of two PINNED staging buffers (copy stream) -> a few
// host threads move the previous group from the staging buffer into the caller's (pageable, usually never-touched)
// array (its first-touch page faults are spread over those threads too; a MADV_HUGEPAGE hint was tried and gained nothing).  A plain hipMemcpy into pageable memory does the last two steps on one thread: 0.79 ms per 8.4 MB sample.
int32_t si_reconstruct(si_ctx* ctx, const double* Z, int64_t C, double* W_out) {
  CHECK_CTX(ctx);
  if (!ctx->i_ready) return fail(ctx, SI_ERR_STATE, "si_reconstruct: call si_infer_setup first");
  if (!Z || C <= 0 || !W_out) return fail(ctx, SI_ERR_INVALID, "si_reconstruct: bad argument");
  BIND(ctx);
  const int64_t N = ctx->iN, ldw = pad_ld(N);
  const int32_t M = ctx->iM;
  // samples per pipeline stage: ~32 MB of output, at most 64 samples, and at least four stages when C allows it
  const int64_t group_cap = std::max<int64_t>(1, std::min<int64_t>(64, ((int64_t)32 << 20) / (N * 8)));   // sizes the buffers once per N
  const int64_t group = std::max<int64_t>(1, std::min<int64_t>((C + 3) / 4, group_cap));
  hipEvent_t ev_comp[2] = {nullptr, nullptr}, ev_copy[2] = {nullptr, nullptr};
  hipError_t e = hipSuccess;
  if (!ctx->stream2) e = hipStreamCreateWithFlags(&ctx->stream2, hipStreamNonBlocking);
  if (e == hipSuccess && (ctx->d_stage_cap < (size_t)ldw * (size_t)group_cap || ctx->d_zstage_cap < (size_t)M * (size_t)group_cap)) {
    (void)hipStreamSynchronize(ctx->stream);
    for (int b = 0; b < 2; ++b) {
      dev_free(ctx->d_stage[b]);
      dev_free(ctx->d_zstage[b]);
      ctx->d_stage[b] = ctx->d_zstage[b] = nullptr;
    }
    ctx->d_stage_cap = ctx->d_zstage_cap = 0;
    for (int b = 0; b < 2 && e == hipSuccess; ++b)
      if (dev_alloc(&ctx->d_stage[b], (size_t)ldw * group_cap) != hipSuccess || dev_alloc(&ctx->d_zstage[b], (size_t)M * group_cap) != hipSuccess)
        e = hipErrorOutOfMemory;
    if (e == hipSuccess) {
      ctx->d_stage_cap = (size_t)ldw * (size_t)group_cap;
      ctx->d_zstage_cap = (size_t)M * (size_t)group_cap;
    }
  }
  double* const* dW = ctx->d_stage;
  double* const* dZ = ctx->d_zstage;
  if (e == hipSuccess && ctx->h_stage_cap < (size_t)N * (size_t)group_cap) {   // the staging buffers stay with the context
    for (int b = 0; b < 2; ++b) {
      if (ctx->h_stage[b]) (void)hipHostFree(ctx->h_stage[b]);
      ctx->h_stage[b] = nullptr;
    }
    ctx->h_stage_cap = 0;
    for (int b = 0; b < 2 && e == hipSuccess; ++b)
      e = hipHostMalloc(reinterpret_cast<void**>(&ctx->h_stage[b]), (size_t)N * group_cap * sizeof(double), hipHostMallocDefault);
    if (e == hipSuccess) ctx->h_stage_cap = (size_t)N * (size_t)group_cap;
  }
  double* const* hp = ctx->h_stage;
  for (int b = 0; b < 2 && e == hipSuccess; ++b) {
    e = hipEventCreateWithFlags(&ev_comp[b], hipEventDisableTiming);
    if (e == hipSuccess) e = hipEventCreateWithFlags(&ev_copy[b], hipEventDisableTiming);
  }
  auto drain = [&](int64_t g) {   // group g has been DMA'd into its staging buffer: move it into the caller's array
    const int b = (int)(g & 1);
    const int64_t c0 = g * group, nc = std::min(group, C - c0);
    hipError_t w = hipEventSynchronize(ev_copy[b]);
    if (w == hipSuccess) host_copy(W_out + c0 * N, hp[b], (size_t)N * (size_t)nc * sizeof(double));
    return w;
  };
  const int64_t ngroups = (C + group - 1) / group;
  for (int64_t g = 0; g < ngroups && e == hipSuccess; ++g) {
    const int b = (int)(g & 1);
    const int64_t c0 = g * group, nc = std::min(group, C - c0);
    // dW[b] / dZ[b] are free once the DMA of group g - 2 has read them; hp[b] once that group has been drained (below)
    if (g >= 2) e = hipStreamWaitEvent(ctx->stream, ev_copy[b], 0);
    if (e == hipSuccess) e = hipMemcpyAsync(dZ[b], Z + c0 * M, (size_t)M * nc * sizeof(double), hipMemcpyHostToDevice, ctx->stream);
    if (e != hipSuccess) break;
    {
      ProfScope ps(ctx, SI_K_RECON, 2.0 * (double)N * M * nc, (double)N * (M + 1 + nc) * 8.0);
      launch_reconstruct(ctx->stream, ctx->i_swa, ctx->i_P, ctx->ldP, N, M, dZ[b], (int32_t)nc, dW[b], ldw, ctx->num_cu);
    }
    e = hipEventRecord(ev_comp[b], ctx->stream);
    if (e == hipSuccess) e = hipStreamWaitEvent(ctx->stream2, ev_comp[b], 0);
    if (e == hipSuccess)
      e = hipMemcpy2DAsync(hp[b], (size_t)N * sizeof(double), dW[b], (size_t)ldw * sizeof(double), (size_t)N * sizeof(double), (size_t)nc,
                           hipMemcpyDeviceToHost, ctx->stream2);
    if (e == hipSuccess) e = hipEventRecord(ev_copy[b], ctx->stream2);
    if (e == hipSuccess && g >= 1) e = drain(g - 1);   // overlaps the DMA of group g
  }
  if (e == hipSuccess) e = drain(ngroups - 1);
  (void)hipStreamSynchronize(ctx->stream2);
  (void)hipStreamSynchronize(ctx->stream);
  for (int b = 0; b < 2; ++b) {
    if (ev_comp[b]) (void)hipEventDestroy(ev_comp[b]);
    if (ev_copy[b]) (void)hipEventDestroy(ev_copy[b]);
  }
  if (e != hipSuccess) return fail(ctx, e == hipErrorOutOfMemory ? SI_ERR_NOMEM : SI_ERR_HIP, std::string("si_reconstruct: ") + hipGetErrorString(e));
  return SI_OK;
}

}  // extern "C"
