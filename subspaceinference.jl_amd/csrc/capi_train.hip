// On-device training step for Dense chains with the mse cost.  Replaces the caller-side body of the
// reference's loop, src/subspace_construction.jl:39-43
//     gs = gradient(ps) do training_loss = cost(model, d...) end ;  Flux.update!(opt, ps, gs)
// for `cost = (m, x, y) -> Flux.Losses.mse(m(x), y)` and opt in {Descent, Momentum, ADAM} (Flux 0.11.2 semantics:
// Float32 parameters and Float32 optimiser state, Float64 arithmetic because the data are Float64).  The weights never
// leave the GPU: si_train_push feeds K1 from the device-resident Float32 vector.
#include <algorithm>
#include <cmath>
#include <cstring>
#include <string>

#include "si_internal.h"

#ifdef SI_DEV_KNOBS   // development build: the library's own buffers through the guard-page allocator (guard_alloc.hip; SI_GUARD_ALLOC=end|begin)
namespace si {
hipError_t guard_malloc(void** out, size_t bytes);
hipError_t guard_free(void* p);
}
#define hipMalloc(p, n) si::guard_malloc((void**)(p), (n))
#define hipFree(p) si::guard_free((void*)(p))
#endif

namespace si {

__global__ __launch_bounds__(256) void gather_cols_kernel(const double* __restrict__ src, int rows, const int64_t* __restrict__ idx,
                                                          int64_t nb, double* __restrict__ dst) {
  const int64_t total = (int64_t)rows * nb;
  const int64_t stride = (int64_t)gridDim.x * blockDim.x;
  for (int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; e < total; e += stride) {
    const int64_t j = e / rows;
    const int r = (int)(e - j * rows);
    dst[e] = src[r + (int64_t)rows * idx[j]];
  }
}

__global__ __launch_bounds__(256) void widen_kernel(const float* __restrict__ w32, int64_t n, double* __restrict__ w64) {
  const int64_t stride = (int64_t)gridDim.x * blockDim.x;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) w64[i] = (double)w32[i];
}

__global__ __launch_bounds__(256) void gather_cols_f32_kernel(const float* __restrict__ src, int rows, const int64_t* __restrict__ idx,
                                                              int64_t nb, float* __restrict__ dst) {
  const int64_t total = (int64_t)rows * nb;
  const int64_t stride = (int64_t)gridDim.x * blockDim.x;
  for (int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; e < total; e += stride) {
    const int64_t j = e / rows;
    const int r = (int)(e - j * rows);
    dst[e] = src[r + (int64_t)rows * idx[j]];
  }
}
__global__ __launch_bounds__(256) void narrow_kernel(const double* __restrict__ src, int64_t n, float* __restrict__ dst) {
  const int64_t stride = (int64_t)gridDim.x * blockDim.x;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) dst[i] = (float)src[i];
}

// Flux 0.11.2 apply! + update!:  x .-= apply!(opt, x, g), state zero(x) (Float32), arithmetic in Float64
// g32: the gradient arrays are Float32 (compute_dtype = SI_F32: g holds Float32 values) -- then `apply!` writes its step back
// into the Float32 array `delta` (rounded) and `x .-= delta` is a Float32 subtraction; with a Float64 gradient the step stays
// Float64 and x - step is rounded once on the store into x
__global__ __launch_bounds__(256) void optimiser_kernel(float* __restrict__ w, float* __restrict__ m, float* __restrict__ v,
                                                        const double* __restrict__ g, int64_t n, int kind, double eta,
                                                        double p1, double p2, double bp1, double bp2, int g32) {
#pragma clang fp contract(off)
  const int64_t stride = (int64_t)gridDim.x * blockDim.x;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) {
    const double gi = g[i];
    double step;
    if (kind == 0) {  // Descent: delta .*= eta
      step = gi * eta;
    } else if (kind == 1) {  // Momentum: v = rho*v - eta*delta; delta = -v
      const float vn = (float)(p1 * (double)m[i] - eta * gi);
      m[i] = vn;
      step = -(double)vn;
    } else {  // ADAM
      const float mt = (float)(p1 * (double)m[i] + (1.0 - p1) * gi);
      const float vt = (float)(p2 * (double)v[i] + (1.0 - p2) * (gi * gi));
      m[i] = mt;
      v[i] = vt;
      step = (double)mt / (1.0 - bp1) / (sqrt((double)vt / (1.0 - bp2)) + 1e-8) * eta;
    }
    if (g32)
      w[i] = w[i] - (float)step;
    else
      w[i] = (float)((double)w[i] - step);
  }
}

static int grid_for(int64_t n, int num_cu) {
  int64_t b = (n + 255) / 256;
  if (b > (int64_t)num_cu * 8) b = (int64_t)num_cu * 8;
  return (int)(b < 1 ? 1 : b);
}

template <typename T>
static bool alloc(T** p, size_t count) {
  *p = nullptr;
  return hipMalloc(reinterpret_cast<void**>(p), (count ? count : 1) * sizeof(T)) == hipSuccess;
}
template <typename T>
static void release(T*& p) {
  if (p) (void)hipFree((void*)p);
  p = nullptr;
}

void free_train(Ctx* c) {
  TrainState* t = c->train;
  if (!t) return;
  release(t->X); release(t->Y); release(t->Xb); release(t->Yb); release(t->idx); release(t->w32); release(t->m32);
  release(t->v32); release(t->w64); release(t->gw); release(t->delta[0]); release(t->delta[1]); release(t->bwpart);
  release(t->rspart); release(t->ssepart); release(t->sse); release(t->part); release(t->Xc); release(t->wpack);
  release(t->scratch.bwpart); release(t->scratch.rspart); release(t->scratch.wt); release(t->scratch.dbtmp);
  for (auto& h : t->hs) release(h);
  release(t->X32); release(t->Xb32);
  if (t->ws32) {
    sweep_f32_free(*t->ws32);
    delete t->ws32;
    t->ws32 = nullptr;
  }
  for (auto& h : t->pidx) release(h);
  for (int b = 0; b < 2; ++b) {
    if (t->idx_pin[b]) (void)hipHostFree(t->idx_pin[b]);
    if (t->idx_ev[b]) (void)hipEventDestroy(t->idx_ev[b]);
  }
  delete t;
  c->train = nullptr;
}

// ---- the fp32 forward + reverse sweep shared by the training step and si_logdensity_grad (compute_dtype = SI_F32) --------------
void sweep_f32_free(SweepF32Ws& ws) {
  for (auto& h : ws.hs32) release(h);
  ws.hs32.clear();
  release(ws.delta32[0]); release(ws.delta32[1]); release(ws.gw32); release(ws.wt32); release(ws.zero32); release(ws.part32);
  release(ws.rspart64); release(ws.tailpart64); release(ws.yhat64);
}
bool sweep_f32_alloc(Ctx* ctx, SweepF32Ws& ws, const si_layer* layers, int L, bool fuse_tail, int64_t N, int32_t in_dim, int32_t out_dim,
                     int64_t Bmax) {
  size_t maxpart32 = 1, maxwt = 1, maxin = (size_t)in_dim, maxw = 1;
  for (int l = 0; l < L; ++l) {
    maxpart32 = std::max(maxpart32, backward_weight_f32_part_elems(layers[l].out, layers[l].in, Bmax, ctx->num_cu));
    maxwt = std::max(maxwt, (size_t)layers[l].out * (size_t)layers[l].in);
    maxin = std::max(maxin, (size_t)layers[l].in);
    maxw = std::max(maxw, (size_t)layers[l].out);
  }
  const size_t wide = std::max(maxin, maxw);
  ws.hs32.assign((size_t)L, nullptr);
  bool ok = alloc(&ws.delta32[0], maxw * (size_t)Bmax) && alloc(&ws.delta32[1], maxw * (size_t)Bmax) && alloc(&ws.gw32, (size_t)pad_ld(N)) &&
            alloc(&ws.wt32, maxwt) && alloc(&ws.zero32, wide) && alloc(&ws.part32, maxpart32) &&
            alloc(&ws.rspart64, rowsum_f32_part_elems((int)wide)) && alloc(&ws.yhat64, (size_t)out_dim * (size_t)Bmax) &&
            (!fuse_tail || alloc(&ws.tailpart64, tail_bwd_f32_part_elems(layers[L - 1].out, layers[L - 1].in)));
  for (int l = 0; l < L && ok; ++l)   // (a fused head's own output lives in yhat64)
    if (!(fuse_tail && l == L - 1)) ok = alloc(&ws.hs32[(size_t)l], (size_t)layers[l].out * (size_t)Bmax);
  if (ok) ok = hipMemsetAsync(ws.zero32, 0, wide * sizeof(float), ctx->stream) == hipSuccess;
  if (!ok) sweep_f32_free(ws);
  return ok;
}

int32_t dense_value_and_grad_f32(Ctx* ctx, hipStream_t st, const DenseSweepF32& s) {
  SweepF32Ws& ws = *s.ws;
  const size_t nl = s.nl;
  const int64_t nb = s.B;
  const float* w = s.w32;
  const float* h = s.X32;
  const size_t nplain = s.fuse_tail ? nl - 2 : nl;
  for (size_t l = 0; l < nplain; ++l) {
    const si_layer& ly = s.layers[l];
    ProfScope ps(ctx, SI_K_DENSE, 2.0 * (double)ly.in * ly.out * (double)nb, 0.0);
    launch_dense_f32(st, w + ly.w_off, w + ly.b_off, h, ws.hs32[l], ly.out, ly.in, nb, ly.act);
    h = ws.hs32[l];
  }
  const si_layer& ll = s.layers[nl - 1];
  const int64_t d = (int64_t)ll.out * nb;
  if (s.fuse_tail) {
    const si_layer& ly = s.layers[nl - 2];
    const int slots = dense_f32_fused_slots(ly.out, ly.in, ly.w_off % 4 == 0);
    {
      ProfScope ps(ctx, SI_K_DENSE, 2.0 * ((double)ly.in * ly.out + (double)ll.in * ll.out) * (double)nb, 0.0);
      launch_dense_f32_fused(st, w + ly.w_off, w + ly.b_off, h, ly.out, ly.in, nb, ly.act, w + ll.w_off, ll.out, s.part, ChainBatch(),
                             ws.hs32[nl - 2]);
    }
    launch_tail_sse(st, s.part, slots, ll.out, nb, s.w64 + ll.b_off, ll.act, s.Y, ws.yhat64, s.ssepart, s.sse_blocks);
    launch_sse_final(st, s.ssepart, s.sse_blocks, s.sse);
  } else {
    launch_sse_f32(st, h, s.Y, d, s.ssepart, s.sse_blocks, s.sse);
  }
  double bflops = 0.0;
  for (size_t l = 0; l < nl; ++l) bflops += 4.0 * (double)s.layers[l].in * s.layers[l].out * (double)nb;
  ProfScope ps(ctx, SI_K_BACKWARD, bflops, 0.0);
  SI_HIP(ctx, hipMemsetAsync(ws.gw32, 0, (size_t)pad_ld(s.N) * sizeof(float), st));
  int cur = 0;
  launch_delta_out_f32(st, s.Y, s.fuse_tail ? ws.yhat64 : nullptr, s.fuse_tail ? nullptr : h, d, s.scale, ll.act, ws.delta32[cur]);
  size_t top = nl;
  bool have_db = false;   // db of layer top-1 already produced (by the pass that formed its Delta)
  if (s.fuse_tail) {
    const si_layer& lp = s.layers[nl - 2];
    launch_mul_dact_rowsum_f32(st, ws.delta32[cur], nullptr, ll.out, nb, SI_ACT_IDENTITY, nullptr, ws.rspart64, ws.gw32 + ll.b_off);
    launch_tail_bwd_f32(st, w + ll.w_off, ws.delta32[cur], ws.hs32[nl - 2], ll.out, ll.in, nb, lp.act, ws.delta32[cur ^ 1], ws.tailpart64,
                        ws.gw32 + ll.w_off, ws.gw32 + lp.b_off);
    cur ^= 1;
    top = nl - 1;
    have_db = true;
  }
  for (size_t li = top; li-- > 0;) {
    const si_layer& ly = s.layers[li];
    const float* hprev = li > 0 ? ws.hs32[li - 1] : s.X32;
    launch_backward_weight_f32(st, ws.delta32[cur], hprev, ws.part32, ly.out, ly.in, nb, ctx->num_cu, ws.gw32 + ly.w_off);
    if (!have_db)
      launch_mul_dact_rowsum_f32(st, ws.delta32[cur], nullptr, ly.out, nb, SI_ACT_IDENTITY, nullptr, ws.rspart64, ws.gw32 + ly.b_off);
    if (li > 0) {
      // Delta_{l-1} = (W_l' Delta_l) .* act'(H_{l-1}): the forward kernel on W_l' (out' = in, in' = out, zero bias) with the
      // multiply in its store, then the row sums (db of layer l-1); shapes the LDS-DMA kernel does not take: one elementwise pass
      // behind the GEMM does both
      const si_layer& lq = s.layers[li - 1];
      launch_transpose_f32(st, w + ly.w_off, ly.out, ly.in, ws.wt32);
      if (launch_dense_f32_dx(st, ws.wt32, ws.zero32, ws.delta32[cur], ws.delta32[cur ^ 1], ly.in, ly.out, nb, ws.hs32[li - 1], lq.act))
        launch_mul_dact_rowsum_f32(st, ws.delta32[cur ^ 1], nullptr, ly.in, nb, SI_ACT_IDENTITY, nullptr, ws.rspart64, ws.gw32 + lq.b_off);
      else
        launch_mul_dact_rowsum_f32(st, ws.delta32[cur ^ 1], ws.hs32[li - 1], ly.in, nb, lq.act, ws.delta32[cur ^ 1], ws.rspart64,
                                   ws.gw32 + lq.b_off);
      cur ^= 1;
      have_db = true;
    }
  }
  SI_HIP(ctx, hipGetLastError());
  return SI_OK;
}

void launch_widen_f32_to_f64(hipStream_t st, const float* src, int64_t n, double* dst, int num_cu) {
  hipLaunchKernelGGL(widen_kernel, dim3(grid_for(n, num_cu)), dim3(256), 0, st, src, n, dst);
}

}  // namespace si

using namespace si;

extern "C" {

static int32_t train_setup_impl(si_ctx* ctx, const si_layer* layers, int32_t L, int64_t N, const float* w0, const void* Xv,
                                const void* Yv, int32_t data_dtype, int32_t in_dim, int32_t out_dim, int64_t B_total, int64_t batch_max,
                                int32_t opt_kind, double eta, double p1, double p2, int32_t compute_dtype) {
  if (!ctx) return SI_ERR_INVALID;
  if (!layers || L <= 0 || N <= 0 || !w0 || !Xv || !Yv || in_dim <= 0 || out_dim <= 0 || B_total <= 0 || batch_max <= 0 ||
      batch_max > B_total || opt_kind < 0 || opt_kind > 2)
    return fail(ctx, SI_ERR_INVALID, "si_train_setup: bad argument");
  if ((data_dtype != SI_F32 && data_dtype != SI_F64) || (compute_dtype != SI_F32 && compute_dtype != SI_F64 && compute_dtype != SI_DTYPE_OF_DATA))
    return fail(ctx, SI_ERR_INVALID, "si_train_setup_ex: data_dtype is SI_F32 / SI_F64, compute_dtype SI_F32 / SI_F64 / SI_DTYPE_OF_DATA");
  const bool f32 = (compute_dtype == SI_DTYPE_OF_DATA ? data_dtype : compute_dtype) == SI_F32;
  NetPlan plan;
  {
    const int32_t prc = net_plan(ctx, "si_train_setup", layers, L, N, in_dim, out_dim, plan);
    if (prc != SI_OK) return prc;
  }
  int64_t maxw = plan.max_elems;
  size_t maxpart = 1;
  if (!plan.has_conv) {
    maxw = 1;
    for (int l = 0; l < L; ++l) {
      const si_layer& ly = layers[l];
      maxw = std::max<int64_t>(maxw, ly.out);
      maxpart = std::max(maxpart, backward_weight_part_elems(ly.out, ly.in, batch_max, ctx->num_cu));
    }
  }
  const bool fuse_tail = !plan.has_conv && (L >= 2) && layers[L - 1].out <= SI_FUSE_MAX_OUT &&
                         layers[L - 1].act < SI_ACT_LEAKYRELU && layers[L - 2].act < SI_ACT_LEAKYRELU;
  if (fuse_tail) maxpart = std::max(maxpart, tail_bwd_part_elems(layers[L - 1].out, layers[L - 1].in));
  if (f32 && plan.has_conv)
    return fail(ctx, SI_ERR_INVALID, "si_train_setup_ex: compute_dtype = SI_F32 is implemented for Dense chains; Conv / MaxPool / flatten chains train in SI_F64");
  SI_HIP(ctx, hipSetDevice(ctx->device));
  SI_HIP(ctx, hipStreamSynchronize(ctx->stream));
  free_train(ctx);
  TrainState* t = new TrainState();
  ctx->train = t;
  t->layers.assign(layers, layers + L);
  t->N = N; t->Btot = B_total; t->Bmax = batch_max; t->in_dim = in_dim; t->out_dim = out_dim;
  t->opt = opt_kind; t->eta = eta; t->p1 = p1; t->p2 = p2; t->bp1 = p1; t->bp2 = p2;  // ADAM: beta powers start at beta
  t->sse_blocks = sse_num_blocks((int64_t)out_dim * batch_max, ctx->num_cu);
  t->fuse_tail = fuse_tail;
  t->fuse_slots = fuse_tail ? dense_fused_slots(layers[L - 2].out) : 0;
  t->f32 = f32;
  if (f32 && fuse_tail)   // (w32 is 256-byte aligned: a layer's W is 16-byte aligned iff w_off % 4 == 0)
    t->fuse_slots = std::max(t->fuse_slots, dense_f32_fused_slots(layers[L - 2].out, layers[L - 2].in, layers[L - 2].w_off % 4 == 0));
  t->hs.assign((size_t)L, nullptr);
  t->plan = plan;
  size_t nb = 1, nr = 1, nw = 1, nd = 1;
  if (plan.has_conv) net_scratch_sizes(plan, batch_max, ctx->num_cu, &nb, &nr, &nw, &nd);
  bool ok = (!plan.has_conv || (alloc(&t->wpack, plan.wpack_elems) && alloc(&t->scratch.bwpart, nb) && alloc(&t->scratch.rspart, nr) &&
                                alloc(&t->scratch.wt, nw) && alloc(&t->scratch.dbtmp, nd))) &&
            (!plan.input_spatial || alloc(&t->Xc, (size_t)plan.in_elems * batch_max)) &&
            (f32 || (alloc(&t->X, (size_t)in_dim * B_total) && alloc(&t->Xb, (size_t)in_dim * batch_max))) &&
            alloc(&t->Y, (size_t)out_dim * B_total) && alloc(&t->Yb, (size_t)out_dim * batch_max) &&
            alloc(&t->idx, (size_t)batch_max) &&
            hipHostMalloc((void**)&t->idx_pin[0], (size_t)batch_max * sizeof(int64_t), hipHostMallocDefault) == hipSuccess &&
            hipHostMalloc((void**)&t->idx_pin[1], (size_t)batch_max * sizeof(int64_t), hipHostMallocDefault) == hipSuccess &&
            hipEventCreateWithFlags(&t->idx_ev[0], hipEventDisableTiming) == hipSuccess &&
            hipEventCreateWithFlags(&t->idx_ev[1], hipEventDisableTiming) == hipSuccess &&
            alloc(&t->w32, (size_t)N) && alloc(&t->m32, (size_t)N) &&
            alloc(&t->v32, (size_t)N) && alloc(&t->w64, (size_t)pad_ld(N)) && alloc(&t->gw, (size_t)pad_ld(N)) &&
            (f32 || (alloc(&t->delta[0], (size_t)maxw * batch_max) && alloc(&t->delta[1], (size_t)maxw * batch_max) &&
                     alloc(&t->bwpart, maxpart) && alloc(&t->rspart, (size_t)rowsum_chunks() * maxw))) &&
            alloc(&t->ssepart, (size_t)t->sse_blocks) && alloc(&t->sse, 1) &&
            (!fuse_tail || alloc(&t->part, (size_t)t->fuse_slots * out_dim * batch_max));
  t->pidx.assign((size_t)L, nullptr);
  if (f32 && ok) {
    t->ws32 = new SweepF32Ws();
    ok = alloc(&t->X32, (size_t)in_dim * B_total) && alloc(&t->Xb32, (size_t)in_dim * batch_max) &&
         sweep_f32_alloc(ctx, *t->ws32, layers, L, fuse_tail, N, in_dim, out_dim, batch_max);
  }
  for (int l = 0; l < L && ok && !f32; ++l) {
    if (plan.has_conv && net_grad_fused(plan, (size_t)l))   // Conv + MaxPool as one kernel: a byte index instead of the activation
      ok = alloc(&t->pidx[(size_t)l], net_pidx_bytes(plan, (size_t)l, batch_max));
    else
      ok = alloc(&t->hs[(size_t)l], (size_t)plan.L[(size_t)l].out_elems * batch_max);
  }
  t->scratch.pidx = t->pidx.data();
  if (!ok) {
    free_train(ctx);
    return fail(ctx, SI_ERR_NOMEM, "si_train_setup: device allocation failed");
  }
  {
    // the data in the element type the step computes in (X) / sums the loss in (Y: fp64 always); one conversion at set-up where
    // the caller's type differs (Float32 -> Float64 is exact; Float64 -> Float32 rounds once, as `Float32.(X)` would)
    const size_t nx = (size_t)in_dim * B_total, ny = (size_t)out_dim * B_total;
    const size_t esz = data_dtype == SI_F32 ? 4 : 8;
    void* stage = nullptr;
    const bool x_conv = (data_dtype == SI_F32) != f32, y_conv = data_dtype == SI_F32;
    if ((x_conv || y_conv) && hipMalloc(&stage, std::max(nx, ny) * esz) != hipSuccess) {
      free_train(ctx);
      return fail(ctx, SI_ERR_NOMEM, "si_train_setup: device allocation failed");
    }
    hipError_t e = hipSuccess;
    const int gx = grid_for((int64_t)nx, ctx->num_cu), gy = grid_for((int64_t)ny, ctx->num_cu);
    if (!x_conv) {
      e = hipMemcpyAsync(f32 ? (void*)t->X32 : (void*)t->X, Xv, nx * esz, hipMemcpyHostToDevice, ctx->stream);
    } else {
      e = hipMemcpyAsync(stage, Xv, nx * esz, hipMemcpyHostToDevice, ctx->stream);
      if (e == hipSuccess) {
        if (f32) hipLaunchKernelGGL(narrow_kernel, dim3(gx), dim3(256), 0, ctx->stream, static_cast<const double*>(stage), (int64_t)nx, t->X32);
        else hipLaunchKernelGGL(widen_kernel, dim3(gx), dim3(256), 0, ctx->stream, static_cast<const float*>(stage), (int64_t)nx, t->X);
      }
    }
    if (e == hipSuccess) {
      if (!y_conv) {
        e = hipMemcpyAsync(t->Y, Yv, ny * 8, hipMemcpyHostToDevice, ctx->stream);
      } else {
        e = hipStreamSynchronize(ctx->stream);   // (the staging buffer is reused)
        if (e == hipSuccess) e = hipMemcpyAsync(stage, Yv, ny * 4, hipMemcpyHostToDevice, ctx->stream);
        if (e == hipSuccess) hipLaunchKernelGGL(widen_kernel, dim3(gy), dim3(256), 0, ctx->stream, static_cast<const float*>(stage), (int64_t)ny, t->Y);
      }
    }
    const hipError_t e2 = hipStreamSynchronize(ctx->stream);
    if (stage) (void)hipFree(stage);
    if (e != hipSuccess || e2 != hipSuccess) {
      free_train(ctx);
      return fail(ctx, SI_ERR_HIP, std::string("si_train_setup: ") + hipGetErrorString(e != hipSuccess ? e : e2));
    }
  }
  SI_HIP(ctx, hipMemcpyAsync(t->w32, w0, (size_t)N * sizeof(float), hipMemcpyHostToDevice, ctx->stream));
  SI_HIP(ctx, hipMemsetAsync(t->m32, 0, (size_t)N * sizeof(float), ctx->stream));
  SI_HIP(ctx, hipMemsetAsync(t->v32, 0, (size_t)N * sizeof(float), ctx->stream));
  SI_HIP(ctx, hipMemsetAsync(t->w64, 0, (size_t)pad_ld(N) * sizeof(double), ctx->stream));
  SI_HIP(ctx, hipStreamSynchronize(ctx->stream));
  return SI_OK;
}

int32_t si_train_setup(si_ctx* ctx, const si_layer* layers, int32_t L, int64_t N, const float* w0, const double* X,
                       const double* Y, int32_t in_dim, int32_t out_dim, int64_t B_total, int64_t batch_max,
                       int32_t opt_kind, double eta, double p1, double p2) {
  return train_setup_impl(ctx, layers, L, N, w0, X, Y, SI_F64, in_dim, out_dim, B_total, batch_max, opt_kind, eta, p1, p2, SI_F64);
}

int32_t si_train_setup_ex(si_ctx* ctx, const si_layer* layers, int32_t L, int64_t N, const float* w0, const void* X, const void* Y,
                          int32_t data_dtype, int32_t in_dim, int32_t out_dim, int64_t B_total, int64_t batch_max, int32_t opt_kind,
                          double eta, double p1, double p2, int32_t compute_dtype) {
  return train_setup_impl(ctx, layers, L, N, w0, X, Y, data_dtype, in_dim, out_dim, B_total, batch_max, opt_kind, eta, p1, p2,
                          compute_dtype);
}

int32_t si_train_compute_dtype(si_ctx* ctx, int32_t* out) {
  if (!ctx) return SI_ERR_INVALID;
  if (!ctx->train || !out) return fail(ctx, SI_ERR_STATE, "si_train_compute_dtype: no training state / NULL output");
  *out = ctx->train->f32 ? SI_F32 : SI_F64;
  return SI_OK;
}

// forward + reverse sweep of the mse cost on the observations idx[0..nb): the gradient w.r.t. the flat weights, scaled
// as if the batch were part of d_total = out_dim * (observations of the WHOLE batch), is left in t->gw; t->sse holds the
// local sum of squared errors
static int32_t train_gradient(si_ctx* ctx, const char* who, const int64_t* idx, int64_t nb, double d_total) {
  TrainState* t = ctx->train;
  if (!t) return fail(ctx, SI_ERR_STATE, std::string(who) + ": call si_train_setup first");
  if (!idx || nb <= 0 || nb > t->Bmax) return fail(ctx, SI_ERR_INVALID, std::string(who) + ": bad batch");
  bool in_order = true;   // idx = 0, 1, ..., nb-1 (a full batch of an unshuffled DataLoader): the data are used in place
  for (int64_t j = 0; j < nb; ++j) {
    if (idx[j] < 0 || idx[j] >= t->Btot)
      return fail(ctx, SI_ERR_INVALID, std::string(who) + ": BoundsError: batch index out of range");
    in_order = in_order && idx[j] == j;
  }
  SI_HIP(ctx, hipSetDevice(ctx->device));
  hipStream_t st = ctx->stream;
  const int64_t N = t->N;
  const size_t nl = t->layers.size();
  const double *Xb = t->X, *Yb = t->Y;
  const float* Xb32 = t->X32;
  if (!in_order) {
    // idx is caller-owned: copied into a pinned buffer, shipped asynchronously; the buffer is free again once its event has
    // passed (two steps later at the earliest) -- no synchronisation of the stream, the call returns while the GPU works
    const int b = t->idx_slot;
    t->idx_slot ^= 1;
    if (t->idx_busy[b]) SI_HIP(ctx, hipEventSynchronize(t->idx_ev[b]));
    std::memcpy(t->idx_pin[b], idx, (size_t)nb * sizeof(int64_t));
    SI_HIP(ctx, hipMemcpyAsync(t->idx, t->idx_pin[b], (size_t)nb * sizeof(int64_t), hipMemcpyHostToDevice, st));
    SI_HIP(ctx, hipEventRecord(t->idx_ev[b], st));
    t->idx_busy[b] = true;
    if (t->f32) {
      hipLaunchKernelGGL(gather_cols_f32_kernel, dim3(grid_for((int64_t)t->in_dim * nb, ctx->num_cu)), dim3(256), 0, st, t->X32,
                         t->in_dim, t->idx, nb, t->Xb32);
      Xb32 = t->Xb32;
    } else {
      hipLaunchKernelGGL(gather_cols_kernel, dim3(grid_for((int64_t)t->in_dim * nb, ctx->num_cu)), dim3(256), 0, st, t->X,
                         t->in_dim, t->idx, nb, t->Xb);
    }
    hipLaunchKernelGGL(gather_cols_kernel, dim3(grid_for((int64_t)t->out_dim * nb, ctx->num_cu)), dim3(256), 0, st, t->Y,
                       t->out_dim, t->idx, nb, t->Yb);
    Xb = t->Xb;
    Yb = t->Yb;
  }
  hipLaunchKernelGGL(widen_kernel, dim3(grid_for(N, ctx->num_cu)), dim3(256), 0, st, t->w32, N, t->w64);
  if (t->f32) {
    // ---- the step in the caller's precision: a Float32 model on Float32 data is a Float32 Zygote pass in the reference
    // (src/subspace_construction.jl:39-43).  fp32 operands and activations on v_mfma_f32_32x32x2_f32; fp64 for the head's
    // partial sums, the loss and every sum over the batch (rounded once into the Float32 gradient).
    const int64_t d = (int64_t)t->out_dim * nb;
    DenseSweepF32 sw{t->layers.data(), nl, t->fuse_tail, t->w32, t->w64, Xb32, Yb, t->ws32, t->part, t->ssepart, t->sse,
                     sse_num_blocks(d, ctx->num_cu), nb, N, -2.0 / d_total};   // d mse / d yhat = 2 (yhat - y) / d
    const int32_t rcs = dense_value_and_grad_f32(ctx, st, sw);
    if (rcs != SI_OK) return rcs;
    // the gradient as the optimiser and the data-parallel all-reduce see it: fp64 words holding the Float32 values
    hipLaunchKernelGGL(widen_kernel, dim3(grid_for(N, ctx->num_cu)), dim3(256), 0, st, t->ws32->gw32, N, t->gw);
    SI_HIP(ctx, hipGetLastError());
    t->grad_ready = true;
    return SI_OK;
  }
  if (t->plan.has_conv) {
    // chains with Conv / MaxPool / flatten layers: the generic forward / reverse sweep of capi_net.hip
    const NetPlan& p = t->plan;
    const double* xin = Xb;
    if (p.input_spatial) {
      net_input(ctx, p, Xb, t->Xc, nb);
      xin = t->Xc;
    }
    int32_t rc = net_forward(ctx, p, t->w64, xin, nb, t->hs.data(), t->wpack, false, nullptr, t->pidx.data());
    if (rc != SI_OK) return rc;
    const int64_t d = (int64_t)t->out_dim * nb;
    const double* yhat = t->hs[nl - 1];
    launch_sse(st, yhat, Yb, d, t->ssepart, sse_num_blocks(d, ctx->num_cu), t->sse);
    ProfScope ps(ctx, SI_K_BACKWARD, 0.0, 0.0);
    SI_HIP(ctx, hipMemsetAsync(t->gw, 0, (size_t)pad_ld(N) * sizeof(double), st));
    launch_delta_out(st, Yb, yhat, d, -2.0 / d_total, SI_ACT_IDENTITY, t->delta[0]);   // d mse / d yhat = 2 (yhat - y) / d
    if ((rc = net_backward(ctx, p, t->w64, xin, nb, t->hs.data(), t->delta[0], t->delta[1], t->gw, t->scratch)) != SI_OK) return rc;
    SI_HIP(ctx, hipGetLastError());
    t->grad_ready = true;
    return SI_OK;
  }
  // forward with every layer's output kept; a narrow head is fed from the epilogue of the layer in front of it
  const double* h = Xb;
  const size_t nplain = t->fuse_tail ? nl - 2 : nl;
  for (size_t l = 0; l < nplain; ++l) {
    const si_layer& ly = t->layers[l];
    ProfScope ps(ctx, SI_K_DENSE, 2.0 * (double)ly.in * ly.out * (double)nb, 0.0);
    launch_dense_f64(st, t->w64 + ly.w_off, t->w64 + ly.b_off, h, t->hs[l], ly.out, ly.in, nb, ly.act);
    h = t->hs[l];
  }
  const int64_t d = (int64_t)t->out_dim * nb;
  const int sse_blocks = sse_num_blocks(d, ctx->num_cu);
  if (t->fuse_tail) {
    const si_layer& ly = t->layers[nl - 2];
    const si_layer& ll = t->layers[nl - 1];
    {
      ProfScope ps(ctx, SI_K_DENSE, 2.0 * ((double)ly.in * ly.out + (double)ll.in * ll.out) * (double)nb, 0.0);
      launch_dense_f64_fused(st, t->w64 + ly.w_off, t->w64 + ly.b_off, h, ly.out, ly.in, nb, ly.act, t->w64 + ll.w_off, ll.out,
                             t->part, ChainBatch(), t->hs[nl - 2]);
    }
    launch_tail_sse(st, t->part, t->fuse_slots, ll.out, nb, t->w64 + ll.b_off, ll.act, Yb, t->hs[nl - 1], t->ssepart,
                    sse_blocks);
    launch_sse_final(st, t->ssepart, sse_blocks, t->sse);
    h = t->hs[nl - 1];
  } else {
    launch_sse(st, h, Yb, d, t->ssepart, sse_blocks, t->sse);
  }
  {
    double bflops = 0.0;
    for (const auto& ly : t->layers) bflops += 4.0 * (double)ly.in * ly.out * (double)nb;
    ProfScope ps(ctx, SI_K_BACKWARD, bflops, 0.0);
    SI_HIP(ctx, hipMemsetAsync(t->gw, 0, (size_t)pad_ld(N) * sizeof(double), st));
    // d mse / d yhat = 2 (yhat - y) / d
    launch_delta_out(st, Yb, h, d, -2.0 / d_total, t->layers[nl - 1].act, t->delta[0]);
    DenseSweep sw{t->layers.data(), nl, t->fuse_tail, t->w64, Xb, t->hs.data(), {t->delta[0], t->delta[1]}, t->gw, t->rspart,
                  t->bwpart, nb};
    const int32_t rcs = dense_reverse_sweep(ctx, st, sw);
    if (rcs != SI_OK) return rcs;
  }
  SI_HIP(ctx, hipGetLastError());
  t->grad_ready = true;
  return SI_OK;
}

// Flux.update!(opt, ps, gs) from the gradient in t->gw
static int32_t train_apply(si_ctx* ctx) {
  TrainState* t = ctx->train;
  hipLaunchKernelGGL(optimiser_kernel, dim3(grid_for(t->N, ctx->num_cu)), dim3(256), 0, ctx->stream, t->w32, t->m32, t->v32,
                     t->gw, t->N, t->opt, t->eta, t->p1, t->p2, t->bp1, t->bp2, t->f32 ? 1 : 0);
  SI_HIP(ctx, hipGetLastError());
  if (t->opt == 2) {
    t->bp1 *= t->p1;
    t->bp2 *= t->p2;
  }
  t->grad_ready = false;
  return SI_OK;
}

int32_t si_train_step(si_ctx* ctx, const int64_t* idx, int64_t nb, double* loss_out) {
  if (!ctx) return SI_ERR_INVALID;
  TrainState* t = ctx->train;
  const double d = t ? (double)t->out_dim * (double)nb : 1.0;
  int32_t rc = train_gradient(ctx, "si_train_step", idx, nb, d);
  if (rc != SI_OK) return rc;
  ProfScope ps(ctx, SI_K_BACKWARD, 0.0, 0.0);
  if ((rc = train_apply(ctx)) != SI_OK) return rc;
  if (loss_out) {
    double sse = 0.0;
    SI_HIP(ctx, hipMemcpyAsync(&sse, t->sse, sizeof(double), hipMemcpyDeviceToHost, ctx->stream));
    SI_HIP(ctx, hipStreamSynchronize(ctx->stream));
    *loss_out = sse / d;  // mse of the batch BEFORE the update, as Zygote's forward value
  }
  return SI_OK;
}

// ---- data-parallel form of the same step ("add one gradient all-reduce per step"): every rank holds the
// same weights / optimiser state and its own share of the batch.  si_train_grad leaves d(mse over the WHOLE batch)/dw
// restricted to this rank's observations in a device buffer, the caller sums that buffer over the ranks IN PLACE
// (RCCL all-reduce on the pointer from si_train_grad_ptr; or _get/_set through the host), si_train_apply updates.
int32_t si_train_grad(si_ctx* ctx, const int64_t* idx, int64_t nb, int64_t nb_total, double* sse_local_out) {
  if (!ctx) return SI_ERR_INVALID;
  if (nb_total < nb) return fail(ctx, SI_ERR_INVALID, "si_train_grad: nb_total < nb");
  TrainState* t = ctx->train;
  const double d_total = t ? (double)t->out_dim * (double)nb_total : 1.0;
  if (t && nb == 0 && nb_total > 0) {  // a rank without a share of a small batch contributes a zero gradient
    SI_HIP(ctx, hipSetDevice(ctx->device));
    SI_HIP(ctx, hipMemsetAsync(t->gw, 0, (size_t)pad_ld(t->N) * sizeof(double), ctx->stream));
    SI_HIP(ctx, hipMemsetAsync(t->sse, 0, sizeof(double), ctx->stream));
    t->grad_ready = true;
    if (sse_local_out) *sse_local_out = 0.0;
    SI_HIP(ctx, hipStreamSynchronize(ctx->stream));
    return SI_OK;
  }
  const int32_t rc = train_gradient(ctx, "si_train_grad", idx, nb, d_total);
  if (rc != SI_OK) return rc;
  if (sse_local_out) SI_HIP(ctx, hipMemcpyAsync(sse_local_out, t->sse, sizeof(double), hipMemcpyDeviceToHost, ctx->stream));
  SI_HIP(ctx, hipStreamSynchronize(ctx->stream));  // the gradient is complete when the caller starts its collective
  return SI_OK;
}

int32_t si_train_grad_ptr(si_ctx* ctx, double** grad_dev_out, int64_t* n_out) {
  if (!ctx) return SI_ERR_INVALID;
  if (!ctx->train || !grad_dev_out || !n_out) return fail(ctx, SI_ERR_STATE, "si_train_grad_ptr: no training state / NULL output");
  *grad_dev_out = ctx->train->gw;
  *n_out = ctx->train->N;
  return SI_OK;
}

int32_t si_train_grad_get(si_ctx* ctx, double* g_out) {
  if (!ctx) return SI_ERR_INVALID;
  if (!ctx->train || !ctx->train->grad_ready || !g_out) return fail(ctx, SI_ERR_STATE, "si_train_grad_get: no gradient pending / NULL output");
  SI_HIP(ctx, hipSetDevice(ctx->device));
  SI_HIP(ctx, hipMemcpyAsync(g_out, ctx->train->gw, (size_t)ctx->train->N * sizeof(double), hipMemcpyDeviceToHost, ctx->stream));
  SI_HIP(ctx, hipStreamSynchronize(ctx->stream));
  return SI_OK;
}

int32_t si_train_grad_set(si_ctx* ctx, const double* g_in) {
  if (!ctx) return SI_ERR_INVALID;
  if (!ctx->train || !ctx->train->grad_ready || !g_in) return fail(ctx, SI_ERR_STATE, "si_train_grad_set: no gradient pending / NULL input");
  SI_HIP(ctx, hipSetDevice(ctx->device));
  SI_HIP(ctx, hipMemcpyAsync(ctx->train->gw, g_in, (size_t)ctx->train->N * sizeof(double), hipMemcpyHostToDevice, ctx->stream));
  SI_HIP(ctx, hipStreamSynchronize(ctx->stream));
  return SI_OK;
}

int32_t si_train_apply(si_ctx* ctx) {
  if (!ctx) return SI_ERR_INVALID;
  if (!ctx->train || !ctx->train->grad_ready) return fail(ctx, SI_ERR_STATE, "si_train_apply: no gradient pending (call si_train_grad)");
  SI_HIP(ctx, hipSetDevice(ctx->device));
  return train_apply(ctx);
}

int32_t si_train_push(si_ctx* ctx, double n) {
  if (!ctx) return SI_ERR_INVALID;
  if (!ctx->train) return fail(ctx, SI_ERR_STATE, "si_train_push: call si_train_setup first");
  if (ctx->train->N != ctx->N) return fail(ctx, SI_ERR_INVALID, "si_train_push: construction and training sizes differ");
  return si_construct_push_dev(ctx, ctx->train->w32, SI_F32, n);
}

int32_t si_train_get_weights(si_ctx* ctx, float* w_out) {
  if (!ctx) return SI_ERR_INVALID;
  if (!ctx->train || !w_out) return fail(ctx, SI_ERR_STATE, "si_train_get_weights: no training state / NULL output");
  SI_HIP(ctx, hipSetDevice(ctx->device));
  SI_HIP(ctx, hipMemcpyAsync(w_out, ctx->train->w32, (size_t)ctx->train->N * sizeof(float), hipMemcpyDeviceToHost, ctx->stream));
  SI_HIP(ctx, hipStreamSynchronize(ctx->stream));
  return SI_OK;
}

int32_t si_train_get_opt_state(si_ctx* ctx, float* m_out, float* v_out, double* beta_pows_out) {
  if (!ctx) return SI_ERR_INVALID;
  if (!ctx->train) return fail(ctx, SI_ERR_STATE, "si_train_get_opt_state: no training state");
  SI_HIP(ctx, hipSetDevice(ctx->device));
  TrainState* t = ctx->train;
  if (m_out && t->m32) SI_HIP(ctx, hipMemcpyAsync(m_out, t->m32, (size_t)t->N * sizeof(float), hipMemcpyDeviceToHost, ctx->stream));
  if (v_out && t->v32) SI_HIP(ctx, hipMemcpyAsync(v_out, t->v32, (size_t)t->N * sizeof(float), hipMemcpyDeviceToHost, ctx->stream));
  SI_HIP(ctx, hipStreamSynchronize(ctx->stream));
  if (beta_pows_out) {
    beta_pows_out[0] = t->bp1;
    beta_pows_out[1] = t->bp2;
  }
  return SI_OK;
}

}  // extern "C"
