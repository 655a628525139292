// C ABI of libsubspace_hip.so (see include/subspace_hip.h for the contract and the reference lines each
// entry point replaces): context, profiling and the subspace CONSTRUCTION (reference src/subspace_construction.jl:31-33,45-52,
// 61-65).  The inference side lives in capi_infer.hip, the samplers and the output map in capi_sample.hip, the on-device
// training step in capi_train.hip, Conv chains in capi_net.hip, the communicator in comm.hip.
// Host-side orchestration only: every arithmetic step of the hot path runs in the HIP kernels of kernels_*.hip, except the
// K x K symmetric eigensolve (eig.cpp, host, K ~ 100).  There is no CPU fallback anywhere in these files.
#include <algorithm>
#include <cmath>
#include <chrono>
#include <cstdlib>
#include <cstring>
#include <new>
#include <vector>

#include "capi_common.h"



namespace si {

static thread_local std::string g_create_err;

int32_t fail(Ctx* c, int32_t code, const std::string& msg) {
  if (c)
    c->err = msg;
  else
    g_create_err = msg;
  return code;
}

// ---- profiling -------------------------------------------------------------------------------
static hipEvent_t get_event(Ctx* c) {
  if (!c->event_pool.empty()) {
    hipEvent_t e = c->event_pool.back();
    c->event_pool.pop_back();
    return e;
  }
  hipEvent_t e = nullptr;
  if (hipEventCreate(&e) != hipSuccess) return nullptr;
  return e;
}

ProfScope::ProfScope(Ctx* c_, int cls_, double flops, double bytes) : c(c_), cls(cls_) {
  if (!c) return;
  c->stats.launches[cls] += 1;
  c->stats.flops[cls] += flops;
  c->stats.bytes[cls] += bytes;
  if (!c->profiling || !((c->prof_mask >> cls) & 1u) || c->pending.size() >= (1u << 20)) return;
  a = get_event(c);
  b = get_event(c);
  if (a && b) (void)hipEventRecord(a, c->stream);
}
ProfScope::~ProfScope() {
  if (!c || !a || !b) return;
  (void)hipEventRecord(b, c->stream);
  c->pending.push_back({a, b, cls});
}

static void resolve_events(Ctx* c) {
  for (auto& p : c->pending) {
    float ms = 0.f;
    if (hipEventElapsedTime(&ms, p.a, p.b) == hipSuccess) c->stats.ms[p.cls] += (double)ms;
    c->event_pool.push_back(p.a);
    c->event_pool.push_back(p.b);
  }
  c->pending.clear();
}


static void free_construct(Ctx* c) {
  // an inference set up on the construction's own W_swa / P dies with it
  if (c->i_ready && c->i_swa == c->d_swa && c->d_swa != nullptr) free_infer(c);
  dev_free(c->d_swa);
  dev_free(c->d_A);
  dev_free(c->d_G);
  c->g_cap = 0;
  c->v_cap = 0;
  dev_free(c->d_Gpart);
  dev_free(c->d_V);
  dev_free(c->d_P);
  dev_free(c->d_B);
  dev_free(c->d_At);
  c->at_cap = 0;
  c->refine_stage = 0;
  c->gpart_bytes = 0;
  c->c_active = c->c_finished = c->gram_valid = false;
  c->K = c->Kcap = c->N = c->ldA = 0;
  c->a_cols_alloc = 0;
  c->a_bytes = 0;
  c->a_zero_dtype = -1;
  c->a_zero_cols = 0;
  c->a_pending = false;
  c->a_dtype = SI_F64;
  c->npush = 0;
  c->M_built = 0;
}

void free_infer(Ctx* c) {
  dev_free(c->d_iswa);
  dev_free(c->d_iP);
  dev_free(c->d_X);
  dev_free(c->d_Y);
  dev_free(c->d_w);
  dev_free(c->d_act[0]);
  dev_free(c->d_act[1]);
  dev_free(c->d_ssepart);
  dev_free(c->d_part);
  dev_free(c->d_outZ);
  dev_free(c->d_outlp);
  c->outZ_cap = c->outlp_cap = 0;
  dev_free(c->d_gridsync);
  c->gridsync_chains = 0;
  dev_free(c->d_cgprog);
  c->fused_ok = false;
  dev_free(c->d_specw);
  dev_free(c->d_specy);
  dev_free(c->d_specperm);
  c->specperm_for = nullptr;
  c->spec_chains = c->spec_fo = 0;
  c->last_density_spec = c->last_loop_spec = 0;
  dev_free(c->d_yhat);
  dev_free(c->d_X32);
  dev_free(c->d_w32);
  dev_free(c->d_act32[0]);
  dev_free(c->d_act32[1]);
  c->f32 = false;
  c->fuse_slots32 = 0;
  for (auto& h : c->d_hs) dev_free(h);
  c->d_hs.clear();
  for (auto& h : c->d_pidx) dev_free(h);
  c->d_pidx.clear();
  c->g_scratch.pidx = nullptr;
  dev_free(c->d_delta[0]);
  dev_free(c->d_delta[1]);
  dev_free(c->d_gw);
  dev_free(c->d_bwpart);
  dev_free(c->d_rspart);
  dev_free(c->d_ptgpart);
  dev_free(c->d_gz);
  if (c->g_ws32) {
    sweep_f32_free(*c->g_ws32);
    delete c->g_ws32;
    c->g_ws32 = nullptr;
  }
  c->g_ready = false;
  c->fuse_tail = false;
  dev_free(c->d_Xc);
  dev_free(c->d_wpack);
  dev_free(c->d_wpack32);
  dev_free(c->d_wsq);
  dev_free(c->d_wsqpart);
  c->sigma_p = 0.0;
  dev_free(c->g_scratch.bwpart);
  dev_free(c->g_scratch.rspart);
  dev_free(c->g_scratch.wt);
  dev_free(c->g_scratch.dbtmp);
  c->plan = NetPlan();
  dev_free(c->d_zcur);
  dev_free(c->d_zprop);
  dev_free(c->d_lpcur);
  dev_free(c->d_sse);
  dev_free(c->d_nacc);
  dev_free(c->d_steps);
  dev_free(c->sw_Z);
  dev_free(c->sw_lp);
  c->chains_cap = 0;
  c->fw_slots = 0;
  c->i_ready = false;
  c->i_swa = c->i_P = nullptr;
}

int32_t construct_adopt(Ctx* c, int64_t N, int32_t M) {
  const int64_t ld = pad_ld(N);
  double *w = nullptr, *p = nullptr;
  if (dev_alloc(&w, (size_t)ld) != hipSuccess || dev_alloc(&p, (size_t)ld * (size_t)M) != hipSuccess) {
    dev_free(w);
    dev_free(p);
    return fail(c, SI_ERR_NOMEM, "allocation of W_swa / P for a received subspace failed");
  }
  SI_HIP(c, hipMemsetAsync(w, 0, (size_t)ld * sizeof(double), c->stream));
  SI_HIP(c, hipMemsetAsync(p, 0, (size_t)ld * (size_t)M * sizeof(double), c->stream));
  construct_install(c, N, M, w, p);
  return SI_OK;
}

void construct_install(Ctx* c, int64_t N, int32_t M, double* w_swa, double* P) {
  (void)hipStreamSynchronize(c->stream);
  free_construct(c);  // also drops an inference bound to the old W_swa / P
  c->N = N;
  c->ldA = pad_ld(N);
  c->d_swa = w_swa;
  c->d_P = P;
  c->M_built = M;
  c->svals.assign((size_t)M, 0.0);
  c->c_active = false;  // no deviation matrix: nothing can be pushed or re-finished
  c->c_finished = true;
}

// The reverse sweep of a Dense chain, shared by si_logdensity_grad and the training step (capi_train.hip).
// Measured and not kept (round 3, tools/bwd_side_ab.py): the weight-gradient GEMM of a layer on a side stream beside the
// data-gradient GEMM of the critical path (nothing in the sweep reads dW; the split-K dW launch fills 480 of 512 slots)
// -- 3 % SLOWER at cfg2 (10.2-10.9 -> 10.5-11.1 ms per value + gradient): two GEMMs sharing the CUs lose more in the
// caches than the idle slots and the fill / drain phases return.
int32_t dense_reverse_sweep(Ctx* ctx, hipStream_t st, const DenseSweep& s) {
  int cur = 0;
  size_t top = s.nl;          // layers [0, top) go through the generic sweep
  bool have_db = false;       // db of layer top-1 already produced by the fused tail
  if (s.fuse_tail) {
    // narrow head: Delta_{L-1}, dW_L and db_{L-1} in one pass over H_{L-1}
    const si_layer& ll = s.layers[s.nl - 1];
    const si_layer& lp = s.layers[s.nl - 2];
    launch_rowsum(st, s.delta[cur], ll.out, s.B, s.rspart, s.gw + ll.b_off);
    launch_tail_bwd(st, s.w + ll.w_off, s.delta[cur], s.hs[s.nl - 2], ll.out, ll.in, s.B, lp.act, s.delta[cur ^ 1], s.bwpart,
                    s.gw + ll.w_off, s.gw + lp.b_off);
    cur ^= 1;
    top = s.nl - 1;
    have_db = true;
  }
  for (size_t li = top; li-- > 0;) {
    const si_layer& ly = s.layers[li];
    const double* hprev = li > 0 ? s.hs[li - 1] : s.X;
    // (db of this layer rides along with its weight gradient unless the fused tail has produced it already)
    launch_backward_weight(st, s.delta[cur], hprev, s.bwpart, ly.out, ly.in, s.B, ctx->num_cu, s.gw + ly.w_off,
                           (have_db && li + 1 == top) ? nullptr : s.gw + ly.b_off);
    if (li > 0) {
      launch_backward_data(st, s.w + ly.w_off, s.delta[cur], hprev, s.delta[cur ^ 1], ly.out, ly.in, s.B, s.layers[li - 1].act);
      cur ^= 1;
    }
  }
  return SI_OK;
}

}  // namespace si

using namespace si;


static void free_push_staging(si_ctx* ctx);

extern "C" {

int32_t si_version(void) { return 500; }

const char* si_last_error(si_ctx* ctx) { return ctx ? ctx->err.c_str() : g_create_err.c_str(); }

int32_t si_create(si_ctx** out, int32_t device_id) {
  if (!out) return fail(nullptr, SI_ERR_INVALID, "si_create: out is NULL");
  *out = nullptr;
  int ndev = 0;
  hipError_t e = hipGetDeviceCount(&ndev);
  if (e != hipSuccess || ndev <= 0)
    return fail(nullptr, SI_ERR_NODEVICE,
                std::string("si_create: no HIP device available (") +
                    (e != hipSuccess ? hipGetErrorString(e) : "device count 0") +
                    "); this library has no CPU backend");
  if (device_id < 0 || device_id >= ndev)
    return fail(nullptr, SI_ERR_INVALID, "si_create: device_id out of range");
  si_ctx* c = new (std::nothrow) si_ctx();
  if (!c) return fail(nullptr, SI_ERR_NOMEM, "si_create: host allocation failed");
  c->device = device_id;
  hipDeviceProp_t prop;
  if ((e = hipSetDevice(device_id)) != hipSuccess || (e = hipGetDeviceProperties(&prop, device_id)) != hipSuccess ||
      (e = hipStreamCreateWithFlags(&c->own_stream, hipStreamNonBlocking)) != hipSuccess) {
    std::string m = std::string("si_create: ") + hipGetErrorString(e);
    delete c;
    return fail(nullptr, SI_ERR_HIP, m);
  }
  c->stream = c->own_stream;
  c->num_cu = prop.multiProcessorCount > 0 ? prop.multiProcessorCount : 256;
  snprintf(c->devname, sizeof(c->devname), "%s (%s)", prop.name, prop.gcnArchName);
  if (std::strncmp(prop.gcnArchName, "gfx950", 6) != 0) {
    std::string m = std::string("si_create: device is ") + prop.gcnArchName +
                    ", but libsubspace_hip.so carries gfx950 code objects only";
    (void)hipStreamDestroy(c->own_stream);
    delete c;
    return fail(nullptr, SI_ERR_NODEVICE, m);
  }
#ifdef SI_DEV_KNOBS
  if (const char* e = getenv("SI_OVERLAP_HALVES")) c->overlap_halves = e[0] == '1';
#endif
  *out = c;
  return SI_OK;
}

int32_t si_destroy(si_ctx* ctx) {
  CHECK_CTX(ctx);
  (void)hipSetDevice(ctx->device);
  (void)hipStreamSynchronize(ctx->stream);
  resolve_events(ctx);
  for (auto e : ctx->event_pool) (void)hipEventDestroy(e);
  comm_release(ctx);
  free_train(ctx);
  free_construct(ctx);
  free_infer(ctx);
  dev_free(ctx->d_wstage);
  free_push_staging(ctx);
  free_wstream(ctx);
  dev_free(ctx->d_nvals);
  if (ctx->h_pin) (void)hipHostFree(ctx->h_pin);
  if (ctx->h_outpin) (void)hipHostFree(ctx->h_outpin);
  for (int b = 0; b < 2; ++b) {
    if (ctx->h_stage[b]) (void)hipHostFree(ctx->h_stage[b]);
    dev_free(ctx->d_stage[b]);
    dev_free(ctx->d_zstage[b]);
  }
  if (ctx->stream2) (void)hipStreamDestroy(ctx->stream2);
  if (ctx->ev_fork) (void)hipEventDestroy(ctx->ev_fork);
  if (ctx->ev_join) (void)hipEventDestroy(ctx->ev_join);
  if (ctx->own_stream) (void)hipStreamDestroy(ctx->own_stream);
  delete ctx;
  return SI_OK;
}

int32_t si_set_stream(si_ctx* ctx, void* hip_stream) {
  CHECK_CTX(ctx);
  BIND(ctx);
  SI_HIP(ctx, hipStreamSynchronize(ctx->stream));
  ctx->stream = hip_stream ? static_cast<hipStream_t>(hip_stream) : ctx->own_stream;
  return SI_OK;
}

int32_t si_synchronize(si_ctx* ctx) {
  CHECK_CTX(ctx);
  BIND(ctx);
  SI_HIP(ctx, hipStreamSynchronize(ctx->stream));
  return SI_OK;
}

int32_t si_set_profiling_classes(si_ctx* ctx, uint32_t class_mask) {
  CHECK_CTX(ctx);
  ctx->prof_mask = class_mask;
  return SI_OK;
}

int32_t si_set_profiling(si_ctx* ctx, int32_t on) {
  CHECK_CTX(ctx);
  ctx->profiling = on != 0;
  return SI_OK;
}

int32_t si_get_stats(si_ctx* ctx, si_stats* out) {
  CHECK_CTX(ctx);
  if (!out) return fail(ctx, SI_ERR_INVALID, "si_get_stats: out is NULL");
  BIND(ctx);
  SI_HIP(ctx, hipStreamSynchronize(ctx->stream));
  resolve_events(ctx);
  *out = ctx->stats;
  return SI_OK;
}

int32_t si_reset_stats(si_ctx* ctx) {
  CHECK_CTX(ctx);
  BIND(ctx);
  SI_HIP(ctx, hipStreamSynchronize(ctx->stream));
  resolve_events(ctx);
  std::memset(&ctx->stats, 0, sizeof(ctx->stats));
  return SI_OK;
}

int32_t si_device_name(si_ctx* ctx, char* buf, int32_t buflen) {
  CHECK_CTX(ctx);
  if (!buf || buflen <= 0) return fail(ctx, SI_ERR_INVALID, "si_device_name: bad buffer");
  std::strncpy(buf, ctx->devname, (size_t)buflen - 1);
  buf[buflen - 1] = 0;
  return SI_OK;
}

// =================================================================================================
// construction
// =================================================================================================
// The deviation matrix: ldA x Kcap elements of a_dtype behind d_A.  Allocated by the FIRST use after si_construct_begin -- a
// push, or si_construct_set_storage -- so that it is sized for the storage type the construction really uses (a construction
// that fits the device only with fp32 columns must not fail at begin on the fp64 size); (re)allocated when too
// small.  The padding rows [N, ldA) of every column must be zero (the Gram kernels read whole slabs): zeroed whenever the
// buffer is new, was last used with another element size, or has more columns in use than were zeroed for this element size
// -- pushes only ever write rows < N, but a push of the OTHER element size writes over this size's padding rows.
static int32_t ensure_A(si_ctx* ctx) {
  ctx->a_pending = false;
  const size_t esz = ctx->a_dtype == SI_F32 ? 4 : 8;
  const size_t need = (size_t)ctx->ldA * (size_t)ctx->Kcap * esz;
  if (ctx->d_A == nullptr || ctx->a_bytes < need) {
    SI_HIP(ctx, hipStreamSynchronize(ctx->stream));
    dev_free(ctx->d_A);
    ctx->a_bytes = 0;
    if (dev_alloc(reinterpret_cast<char**>(&ctx->d_A), need) != hipSuccess) {
      ctx->d_A = nullptr;
      return fail(ctx, SI_ERR_NOMEM, "si_construct_begin: device allocation of the deviation matrix failed");
    }
    ctx->a_bytes = need;
    ctx->a_zero_dtype = -1;
    ctx->a_zero_cols = 0;
  }
  if (ctx->a_zero_dtype != ctx->a_dtype || ctx->a_zero_cols < ctx->Kcap) {
    // only the padding rows [N, ldA) of every column (a strided fill of < 64 elements per column, not the whole matrix)
    if (ctx->ldA > ctx->N)
      SI_HIP(ctx, hipMemset2DAsync(reinterpret_cast<char*>(ctx->d_A) + (size_t)ctx->N * esz, (size_t)ctx->ldA * esz, 0,
                                   (size_t)(ctx->ldA - ctx->N) * esz, (size_t)ctx->Kcap, ctx->stream));
    ctx->a_zero_dtype = ctx->a_dtype;
    ctx->a_zero_cols = ctx->Kcap;
  }
  return SI_OK;
}

int32_t si_construct_begin(si_ctx* ctx, int64_t N, int64_t K_capacity, int32_t max_cols) {
  CHECK_CTX(ctx);
  if (N <= 0 || K_capacity <= 0 || max_cols < 0)
    return fail(ctx, SI_ERR_INVALID, "si_construct_begin: N and K_capacity must be positive, max_cols >= 0");
  BIND(ctx);
  SI_HIP(ctx, hipStreamSynchronize(ctx->stream));
  const int64_t kcap = max_cols > 0 ? std::min<int64_t>(max_cols, K_capacity) : K_capacity;
  if (ctx->d_swa != nullptr && ctx->N == N) {
    // same problem size as the previous construction: keep the buffers (ensure_A re-uses A when it is large enough)
    if (ctx->i_ready && ctx->i_swa == ctx->d_swa) free_infer(ctx);  // an inference bound to the old W_swa / P
    ctx->c_finished = ctx->gram_valid = false;
    ctx->K = 0;
    ctx->npush = 0;
  } else {
    free_construct(ctx);
    ctx->N = N;
    ctx->ldA = pad_ld(N);
    if (dev_alloc(&ctx->d_swa, (size_t)ctx->ldA) != hipSuccess) {
      free_construct(ctx);
      return fail(ctx, SI_ERR_NOMEM, "si_construct_begin: device allocation of W_swa failed");
    }
  }
  ctx->max_cols = max_cols;
  ctx->Kcap = kcap;
  ctx->a_dtype = SI_F64;   // the reference's storage (A = Array{Float64}); si_construct_set_storage changes it before the first push
  ctx->a_pending = true;   // ensure_A runs at the first push / at si_construct_set_storage: sized for the storage type in use
  // W_swa = zeros(N)  (reference :31, quirk Q1: NOT the pretrained weights)
  SI_HIP(ctx, hipMemsetAsync(ctx->d_swa, 0, (size_t)ctx->ldA * sizeof(double), ctx->stream));
  ctx->c_active = true;
  return SI_OK;
}

// NON-DEFAULT option ("fp32 storage is an opt-in bandwidth optimisation that must still meet rtol 1e-4"):
// the deviation columns w - W_swa are formed in fp64 and stored rounded once to fp32; W_swa, the Gram matrix, the eigen-
// decomposition and P stay fp64.  Halves the memory of A (cfg5: 52 -> 26 GB) and the bytes K2 / K3 stream.
int32_t si_construct_set_storage(si_ctx* ctx, int32_t a_dtype) {
  CHECK_CTX(ctx);
  if (!ctx->c_active || ctx->npush != 0)
    return fail(ctx, SI_ERR_STATE, "si_construct_set_storage: call right after si_construct_begin, before the first push");
  if (a_dtype != SI_F32 && a_dtype != SI_F64) return fail(ctx, SI_ERR_INVALID, "si_construct_set_storage: SI_F64 (the reference) or SI_F32");
  BIND(ctx);
  ctx->a_dtype = a_dtype;
  return ensure_A(ctx);
}

int32_t si_construct_set_mean(si_ctx* ctx, const void* w_host, int32_t w_dtype) {
  CHECK_CTX(ctx);
  if (!ctx->c_active || ctx->npush != 0)
    return fail(ctx, SI_ERR_STATE, "si_construct_set_mean: call it right after si_construct_begin, before the first push");
  if (!w_host || (w_dtype != SI_F32 && w_dtype != SI_F64)) return fail(ctx, SI_ERR_INVALID, "si_construct_set_mean: bad pointer or dtype");
  BIND(ctx);
  if (w_dtype == SI_F64) {
    SI_HIP(ctx, hipMemcpyAsync(ctx->d_swa, w_host, (size_t)ctx->N * sizeof(double), hipMemcpyHostToDevice, ctx->stream));
  } else {
    const size_t bytes = (size_t)ctx->N * 4;
    if (ctx->wstage_bytes < bytes) {
      SI_HIP(ctx, hipStreamSynchronize(ctx->stream));
      dev_free(ctx->d_wstage);
      ctx->wstage_bytes = 0;
      if (hipMalloc(&ctx->d_wstage, bytes) != hipSuccess) return fail(ctx, SI_ERR_NOMEM, "si_construct_set_mean: staging allocation failed");
      ctx->wstage_bytes = bytes;
    }
    SI_HIP(ctx, hipMemcpyAsync(ctx->d_wstage, w_host, bytes, hipMemcpyHostToDevice, ctx->stream));
    launch_widen_f32(ctx->stream, static_cast<const float*>(ctx->d_wstage), ctx->d_swa, ctx->N, ctx->num_cu);
  }
  SI_HIP(ctx, hipGetLastError());
  SI_HIP(ctx, hipStreamSynchronize(ctx->stream));  // w_host is caller-owned
  return SI_OK;
}

static int32_t push_common(si_ctx* ctx, const void* w_dev, int32_t w_dtype, double n) {
  // column slot: keep-all appends; with max_cols the oldest column is overwritten (ring). The order of the
  // columns does not change A*A' and hence neither the singular values nor P up to sign.
  int64_t slot;
  if (ctx->max_cols > 0) {
    slot = ctx->npush % ctx->Kcap;
  } else {
    if (ctx->K >= ctx->Kcap) return fail(ctx, SI_ERR_STATE, "si_construct_push: more pushes than K_capacity");
    slot = ctx->K;
  }
  if (ctx->a_pending) {
    const int32_t arc = ensure_A(ctx);
    if (arc != SI_OK) return arc;
  }
  const size_t wsz = w_dtype == SI_F32 ? 4 : 8, asz = ctx->a_dtype == SI_F32 ? 4 : 8;
  {
    ProfScope ps(ctx, SI_K_PUSH, 4.0 * (double)ctx->N, (double)ctx->N * (double)(wsz + 16 + asz));
    launch_swa_dev_push(ctx->stream, w_dev, w_dtype, ctx->d_swa, reinterpret_cast<char*>(ctx->d_A) + (size_t)slot * ctx->ldA * asz,
                        ctx->N, n, ctx->num_cu, ctx->a_dtype);
  }
  SI_HIP(ctx, hipGetLastError());
  ctx->npush += 1;
  ctx->K = std::min(ctx->npush, ctx->Kcap);
  ctx->gram_valid = false;
  ctx->c_finished = false;
  return SI_OK;
}

int32_t si_construct_push_dev(si_ctx* ctx, const void* w_dev, int32_t w_dtype, double n) {
  CHECK_CTX(ctx);
  if (!ctx->c_active) return fail(ctx, SI_ERR_STATE, "si_construct_push_dev: call si_construct_begin first");
  if (!w_dev || (w_dtype != SI_F32 && w_dtype != SI_F64))
    return fail(ctx, SI_ERR_INVALID, "si_construct_push_dev: bad pointer or dtype");
  BIND(ctx);
  return push_common(ctx, w_dev, w_dtype, n);
}

int32_t si_construct_push_batch_dev(si_ctx* ctx, const void* w_dev, int32_t w_dtype, int64_t ld, int32_t count,
                                    const double* n_host) {
  CHECK_CTX(ctx);
  if (!ctx->c_active) return fail(ctx, SI_ERR_STATE, "si_construct_push_batch_dev: call si_construct_begin first");
  if (!w_dev || !n_host || count <= 0 || ld < ctx->N || (w_dtype != SI_F32 && w_dtype != SI_F64))
    return fail(ctx, SI_ERR_INVALID, "si_construct_push_batch_dev: bad pointer, count, ld or dtype");
  if (ctx->max_cols == 0 && ctx->K + count > ctx->Kcap)
    return fail(ctx, SI_ERR_STATE, "si_construct_push: more pushes than K_capacity");
  BIND(ctx);
  if (ctx->a_pending) {
    const int32_t arc = ensure_A(ctx);
    if (arc != SI_OK) return arc;
  }
  if (ctx->nvals_cap < count) {
    SI_HIP(ctx, hipStreamSynchronize(ctx->stream));
    dev_free(ctx->d_nvals);
    if (dev_alloc(&ctx->d_nvals, (size_t)count) != hipSuccess) {
      ctx->nvals_cap = 0;
      return fail(ctx, SI_ERR_NOMEM, "si_construct_push_batch_dev: allocation failed");
    }
    ctx->nvals_cap = count;
  }
  SI_HIP(ctx, hipMemcpyAsync(ctx->d_nvals, n_host, (size_t)count * sizeof(double), hipMemcpyHostToDevice, ctx->stream));
  SI_HIP(ctx, hipStreamSynchronize(ctx->stream));  // n_host is caller-owned and only valid during the call
  const size_t wsz = w_dtype == SI_F32 ? 4 : 8;
  {
    ProfScope ps(ctx, SI_K_PUSH, 4.0 * (double)ctx->N * count, (double)ctx->N * ((double)count * (wsz + (ctx->a_dtype == SI_F32 ? 4 : 8)) + 16.0));
    launch_swa_dev_push_batch(ctx->stream, w_dev, w_dtype, ld, ctx->d_swa, ctx->d_A, ctx->ldA, ctx->N, count,
                              ctx->d_nvals, ctx->max_cols > 0 ? ctx->npush % ctx->Kcap : ctx->K, ctx->Kcap, ctx->num_cu, ctx->a_dtype);
  }
  SI_HIP(ctx, hipGetLastError());
  ctx->npush += count;
  ctx->K = std::min(ctx->npush, ctx->Kcap);
  ctx->gram_valid = false;
  ctx->c_finished = false;
  return SI_OK;
}

// Host snapshots (the Julia wrapper's default path: `W = extract_params(ps)` lives in pageable host memory and is only
// valid during the call).  Pipelined over two pinned staging buffers and two device buffers: the call copies the snapshot
// into pinned memory with the host copy pool (host_copy.cpp), queues H2D + K1 on the stream and RETURNS -- the DMA and
// the kernel overlap the caller's next gradient / update!; the only wait is for the staging buffer of two pushes ago.
static void free_push_staging(si_ctx* ctx) {
  if (ctx->stream2) (void)hipStreamSynchronize(ctx->stream2);
  for (int b = 0; b < 2; ++b) {
    if (ctx->h_wpin[b]) (void)hipHostFree(ctx->h_wpin[b]);
    ctx->h_wpin[b] = nullptr;
    dev_free(ctx->d_wpush[b]);
    if (ctx->ev_wpin[b]) (void)hipEventDestroy(ctx->ev_wpin[b]);
    if (ctx->ev_wk1[b]) (void)hipEventDestroy(ctx->ev_wk1[b]);
    ctx->ev_wpin[b] = ctx->ev_wk1[b] = nullptr;
    ctx->wpin_busy[b] = false;
  }
  ctx->wpin_bytes = 0;
}

int32_t si_construct_push(si_ctx* ctx, const void* w_host, int32_t w_dtype, double n) {
  CHECK_CTX(ctx);
  if (!ctx->c_active) return fail(ctx, SI_ERR_STATE, "si_construct_push: call si_construct_begin first");
  if (!w_host || (w_dtype != SI_F32 && w_dtype != SI_F64))
    return fail(ctx, SI_ERR_INVALID, "si_construct_push: bad pointer or dtype");
  if (ctx->max_cols == 0 && ctx->K >= ctx->Kcap) return fail(ctx, SI_ERR_STATE, "si_construct_push: more pushes than K_capacity");
  BIND(ctx);
  const size_t bytes = (size_t)ctx->N * (w_dtype == SI_F32 ? 4 : 8);
  if (!ctx->stream2) SI_HIP(ctx, hipStreamCreateWithFlags(&ctx->stream2, hipStreamNonBlocking));
  if (ctx->wpin_bytes < bytes) {
    SI_HIP(ctx, hipStreamSynchronize(ctx->stream));
    free_push_staging(ctx);
    for (int b = 0; b < 2; ++b) {
      if (hipHostMalloc(&ctx->h_wpin[b], bytes, hipHostMallocDefault) != hipSuccess ||
          hipMalloc(&ctx->d_wpush[b], bytes) != hipSuccess ||
          hipEventCreateWithFlags(&ctx->ev_wpin[b], hipEventDisableTiming) != hipSuccess ||
          hipEventCreateWithFlags(&ctx->ev_wk1[b], hipEventDisableTiming) != hipSuccess) {
        free_push_staging(ctx);
        return fail(ctx, SI_ERR_NOMEM, "si_construct_push: staging allocation failed");
      }
    }
    ctx->wpin_bytes = bytes;
  }
  // Three stages in flight: the host copy pool fills pinned buffer b (push i), the copy stream moves pinned -> device
  // buffer b, the compute stream runs K1 on it.  Buffer b of two pushes ago must be done: its H2D (host side: the pinned
  // buffer is rewritten) and its K1 (copy stream: the device buffer is rewritten).  The H2Ds of consecutive pushes run back
  // to back on their own stream -- on ONE stream the 9 us K1 sat between them and the link idled 10 % of the time.
  const int b = (int)(ctx->wpin_next & 1);
  ctx->wpin_next += 1;
  if (ctx->wpin_busy[b]) SI_HIP(ctx, hipEventSynchronize(ctx->ev_wpin[b]));
  host_copy(ctx->h_wpin[b], w_host, bytes);
  if (ctx->wpin_busy[b]) SI_HIP(ctx, hipStreamWaitEvent(ctx->stream2, ctx->ev_wk1[b], 0));
  SI_HIP(ctx, hipMemcpyAsync(ctx->d_wpush[b], ctx->h_wpin[b], bytes, hipMemcpyHostToDevice, ctx->stream2));
  SI_HIP(ctx, hipEventRecord(ctx->ev_wpin[b], ctx->stream2));
  SI_HIP(ctx, hipStreamWaitEvent(ctx->stream, ctx->ev_wpin[b], 0));
  ctx->wpin_busy[b] = true;
  const int32_t rc = push_common(ctx, ctx->d_wpush[b], w_dtype, n);
  if (rc == SI_OK) SI_HIP(ctx, hipEventRecord(ctx->ev_wk1[b], ctx->stream));
  return rc;
}

int32_t si_construct_gram(si_ctx* ctx) {
  CHECK_CTX(ctx);
  if (!ctx->c_active || ctx->K <= 0) return fail(ctx, SI_ERR_STATE, "si_construct_gram: nothing pushed");
  BIND(ctx);
  const int64_t K = ctx->K;
  const size_t need = launch_gram(ctx->stream, ctx->d_A, ctx->ldA, ctx->N, K, nullptr, nullptr, ctx->num_cu, nullptr, ctx->a_dtype);
  if (ctx->gpart_bytes < need) {
    SI_HIP(ctx, hipStreamSynchronize(ctx->stream));
    dev_free(ctx->d_Gpart);
    if (hipMalloc(reinterpret_cast<void**>(&ctx->d_Gpart), need) != hipSuccess) {
      ctx->gpart_bytes = 0;
      return fail(ctx, SI_ERR_NOMEM, "si_construct_gram: partial-slab allocation failed");
    }
    ctx->gpart_bytes = need;
  }
  if (ctx->g_cap < K * K) {
    SI_HIP(ctx, hipStreamSynchronize(ctx->stream));
    dev_free(ctx->d_G);
    ctx->g_cap = 0;
    if (dev_alloc(&ctx->d_G, (size_t)K * K) != hipSuccess)
      return fail(ctx, SI_ERR_NOMEM, "si_construct_gram: G allocation failed");
    ctx->g_cap = K * K;
  }
  launch_gram(ctx->stream, ctx->d_A, ctx->ldA, ctx->N, K, ctx->d_Gpart, ctx->d_G, ctx->num_cu, ctx, ctx->a_dtype);
  SI_HIP(ctx, hipGetLastError());
  ctx->gram_valid = true;
  if (ctx->refine_stage != 0 || ctx->d_B) {  // a fresh first-stage Gram: drop the second stage of an earlier finish
    SI_HIP(ctx, hipStreamSynchronize(ctx->stream));
    dev_free(ctx->d_B);
    ctx->refine_stage = 0;
  }
  return SI_OK;
}

int32_t si_construct_gram_get(si_ctx* ctx, double* G_host, int64_t* K_out) {
  CHECK_CTX(ctx);
  if (!ctx->gram_valid) return fail(ctx, SI_ERR_STATE, "si_construct_gram_get: call si_construct_gram first");
  BIND(ctx);
  if (K_out) *K_out = ctx->K;
  if (G_host) {
    SI_HIP(ctx, hipMemcpyAsync(G_host, ctx->d_G, (size_t)ctx->K * ctx->K * sizeof(double), hipMemcpyDeviceToHost,
                               ctx->stream));
    SI_HIP(ctx, hipStreamSynchronize(ctx->stream));
  }
  return SI_OK;
}

int32_t si_construct_gram_ptr(si_ctx* ctx, double** G_dev_out, int64_t* K_out) {
  CHECK_CTX(ctx);
  if (!ctx->gram_valid) return fail(ctx, SI_ERR_STATE, "si_construct_gram_ptr: call si_construct_gram first");
  if (G_dev_out) *G_dev_out = ctx->d_G;
  if (K_out) *K_out = ctx->K;
  return SI_OK;
}

int32_t si_construct_gram_set(si_ctx* ctx, const double* G_host) {
  CHECK_CTX(ctx);
  if (!ctx->gram_valid || !G_host)
    return fail(ctx, SI_ERR_STATE, "si_construct_gram_set: call si_construct_gram first / NULL input");
  BIND(ctx);
  SI_HIP(ctx, hipMemcpyAsync(ctx->d_G, G_host, (size_t)ctx->K * ctx->K * sizeof(double), hipMemcpyHostToDevice,
                             ctx->stream));
  SI_HIP(ctx, hipStreamSynchronize(ctx->stream));
  return SI_OK;
}

// ---- H1 + K3 ----------------------------------------------------------------------------------------------------------
// Two routes to (s, P = A V_M):
//   Gram route        G = A'A -> top-M eigenpairs -> P = A V_M.  Squares the condition number: used while
//                     lambda_M > SI_GRAM_ROUTE_MIN * lambda_1 (s_M > ~3e-5 s_1), where it delivers s to ~1e-5 and better.
//   two-stage route   (ill-conditioned A; what psvd's rtol = 5 eps still resolves)  full eigenbasis V of G ->
//                     B = A V on the device (columns graded: resolved directions sorted out, the unresolved ones mixed
//                     among themselves at their own small scale) -> G2 = B'B on the device -> scaled-criterion Jacobi
//                     on the host (relative accuracy per eigenvalue) -> W -> P = B W_M.  Each stage resolves ~8 decades
//                     of singular-value spread; what is left is the backward error of ANY fp64 SVD, eps*s_1/s_j.
// BoundsError (the reference's U[:,1:M] on a psvd that returned fewer columns) only when s_M <= SI_RANK_RTOL * s_1.
static constexpr double SI_GRAM_ROUTE_MIN = 1e-9;
static constexpr double SI_EPS = 2.220446049250313e-16;

static int32_t ensure_pin(si_ctx* ctx, size_t elems) {
  if (ctx->h_pin_cap >= elems) return SI_OK;
  SI_HIP(ctx, hipStreamSynchronize(ctx->stream));
  if (ctx->h_pin) (void)hipHostFree(ctx->h_pin);
  ctx->h_pin = nullptr;
  ctx->h_pin_cap = 0;
  if (hipHostMalloc(reinterpret_cast<void**>(&ctx->h_pin), elems * sizeof(double), hipHostMallocDefault) != hipSuccess)
    return fail(ctx, SI_ERR_NOMEM, "si_construct_finish: pinned staging allocation failed");
  ctx->h_pin_cap = elems;
  return SI_OK;
}

static int32_t ensure_V(si_ctx* ctx, size_t v_elems) {
  if (ctx->v_cap >= (int64_t)v_elems) return SI_OK;
  SI_HIP(ctx, hipStreamSynchronize(ctx->stream));
  dev_free(ctx->d_V);
  ctx->v_cap = 0;
  if (dev_alloc(&ctx->d_V, v_elems) != hipSuccess) return fail(ctx, SI_ERR_NOMEM, "si_construct_finish: allocation of V failed");
  ctx->v_cap = (int64_t)v_elems;
  return SI_OK;
}

// top-M eigenpairs of the K x K matrix at h_pin[0 .. K*K) (left intact by the fast route, destroyed by the fallback)
static int32_t top_eigen(si_ctx* ctx, int64_t K, int32_t M, std::vector<double>& wtop, std::vector<double>& Vtop) {
  double* const G = ctx->h_pin;
  wtop.assign((size_t)M, 0.0);
  Vtop.assign((size_t)K * M, 0.0);
  const auto t0 = std::chrono::steady_clock::now();
  int erc = 0;
  if (sym_eig_top((int)K, G, (int)M, wtop.data(), Vtop.data()) != 0) {
    std::vector<double> lam((size_t)K);
    erc = sym_eig((int)K, G, lam.data());
    for (int m = 0; m < M && erc == 0; ++m) {
      wtop[(size_t)m] = lam[(size_t)(K - 1 - m)];
      std::copy(G + (size_t)(K - 1 - m) * K, G + (size_t)(K - m) * K, Vtop.data() + (size_t)m * K);
    }
  }
  ctx->stats.ms[SI_K_EIG_HOST] += std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
  ctx->stats.launches[SI_K_EIG_HOST] += 1;
  if (erc != 0) return fail(ctx, SI_ERR_INVALID, "si_construct_finish: eigensolver did not converge");
  return SI_OK;
}

static int32_t fetch_G(si_ctx* ctx, int64_t K) {
  const int32_t rc = ensure_pin(ctx, (size_t)K * K * 2 + (size_t)K * project_mpad((int)K));
  if (rc != SI_OK) return rc;
  SI_HIP(ctx, hipMemcpyAsync(ctx->h_pin, ctx->d_G, (size_t)K * K * sizeof(double), hipMemcpyDeviceToHost, ctx->stream));
  SI_HIP(ctx, hipStreamSynchronize(ctx->stream));
  return SI_OK;
}

int32_t si_construct_needs_refine(si_ctx* ctx, int32_t M, int32_t* out) {
  CHECK_CTX(ctx);
  if (!ctx->gram_valid || !out) return fail(ctx, SI_ERR_STATE, "si_construct_needs_refine: call si_construct_gram first / NULL output");
  if (M <= 0 || M > ctx->K) return fail(ctx, SI_ERR_BOUNDS, "BoundsError: M exceeds the number of deviation columns K");
  BIND(ctx);
  if (ctx->refine_stage == 1) {
    *out = 0;  // already refined: d_G holds the second-stage Gram matrix
    return SI_OK;
  }
  int32_t rc = fetch_G(ctx, ctx->K);
  if (rc != SI_OK) return rc;
  std::vector<double> wtop, Vtop;
  if ((rc = top_eigen(ctx, ctx->K, M, wtop, Vtop)) != SI_OK) return rc;
  *out = !(wtop[0] > 0.0) || wtop[(size_t)M - 1] <= SI_GRAM_ROUTE_MIN * wtop[0];
  return SI_OK;
}

// second stage: B = A * V_full, G2 = B'B left in d_G (so that si_construct_gram_ptr / _get / _set all-reduce IT for a
// row-sharded construction: every rank holds the same all-reduced G, hence the same V_full)
int32_t si_construct_refine(si_ctx* ctx) {
  CHECK_CTX(ctx);
  if (!ctx->gram_valid) return fail(ctx, SI_ERR_STATE, "si_construct_refine: call si_construct_gram first");
  if (ctx->refine_stage == 1) return SI_OK;
  BIND(ctx);
  const int64_t K = ctx->K, N = ctx->N;
  int32_t rc = fetch_G(ctx, K);
  if (rc != SI_OK) return rc;
  double* const G = ctx->h_pin;
  std::vector<double> lam((size_t)K);
  {
    const auto t0 = std::chrono::steady_clock::now();
    const int erc = sym_eig((int)K, G, lam.data());  // ascending; G <- eigenvectors
    ctx->stats.ms[SI_K_EIG_HOST] += std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
    ctx->stats.launches[SI_K_EIG_HOST] += 1;
    if (erc != 0) return fail(ctx, SI_ERR_INVALID, "si_construct_refine: eigensolver did not converge");
  }
  ctx->vfull.assign((size_t)K * K, 0.0);
  for (int64_t j = 0; j < K; ++j) std::copy(G + (size_t)(K - 1 - j) * K, G + (size_t)(K - j) * K, ctx->vfull.data() + (size_t)j * K);
  const int Kpad = project_mpad((int)K);
  double* const V = ctx->h_pin + (size_t)K * K * 2;
  std::fill(V, V + (size_t)K * Kpad, 0.0);
  for (int64_t j = 0; j < K; ++j)
    for (int64_t k = 0; k < K; ++k) V[(size_t)k * Kpad + j] = ctx->vfull[(size_t)j * K + k];
  if ((rc = ensure_V(ctx, (size_t)K * Kpad)) != SI_OK) return rc;
  if (ctx->d_B == nullptr) {
    SI_HIP(ctx, hipStreamSynchronize(ctx->stream));
    if (dev_alloc(&ctx->d_B, (size_t)ctx->ldA * K) != hipSuccess)
      return fail(ctx, SI_ERR_NOMEM, "si_construct_refine: allocation of the second-stage matrix (N x K) failed");
    SI_HIP(ctx, hipMemsetAsync(ctx->d_B, 0, (size_t)ctx->ldA * K * sizeof(double), ctx->stream));  // padding rows stay zero
  }
  SI_HIP(ctx, hipMemcpyAsync(ctx->d_V, V, (size_t)K * Kpad * sizeof(double), hipMemcpyHostToDevice, ctx->stream));
  {
    ProfScope ps(ctx, SI_K_PROJECT, 2.0 * (double)N * (double)K * (double)K, (double)N * (double)(2 * K) * 8.0);
    launch_project(ctx->stream, ctx->d_A, ctx->ldA, N, K, ctx->d_V, (int32_t)K, Kpad, ctx->d_B, ctx->ldA, ctx->num_cu, ctx->a_dtype);
  }
  launch_gram(ctx->stream, ctx->d_B, ctx->ldA, N, K, ctx->d_Gpart, ctx->d_G, ctx->num_cu, ctx);
  SI_HIP(ctx, hipGetLastError());
  SI_HIP(ctx, hipStreamSynchronize(ctx->stream));  // the pinned V may be rewritten
  ctx->refine_stage = 1;
  return SI_OK;
}

static int32_t alloc_P(si_ctx* ctx, int32_t M) {
  if (ctx->d_P != nullptr && ctx->M_built == M) return SI_OK;
  SI_HIP(ctx, hipStreamSynchronize(ctx->stream));
  // P is re-allocated: an inference bound to the old P of this construction must not outlive it
  if (ctx->i_ready && ctx->i_P == ctx->d_P && ctx->d_P != nullptr) free_infer(ctx);
  dev_free(ctx->d_P);
  ctx->M_built = 0;
  if (dev_alloc(&ctx->d_P, (size_t)ctx->ldA * M) != hipSuccess)
    return fail(ctx, SI_ERR_NOMEM, "si_construct_finish: allocation of P failed");
  // zeroed once, for the padding rows [N, ldA): every row < N of every column is written on each finish
  SI_HIP(ctx, hipMemsetAsync(ctx->d_P, 0, (size_t)ctx->ldA * M * sizeof(double), ctx->stream));
  return SI_OK;
}

// K > N: A' on the device (transpose), G = A A' with the Gram kernel, top-M eigenpairs of the N x N matrix on the host,
// the right singular vectors A'U only to read the deterministic column signs from, P = U * Diagonal(s) uploaded.
// *done = false: lambda_M is too close to the rounding floor of a squared-condition Gram matrix (the caller takes the K x K route)
static int32_t finish_wide(si_ctx* ctx, int32_t M, bool* done) {
  *done = false;
  const int64_t K = ctx->K, N = ctx->N, ldt = pad_ld(K);
  int32_t rc;
  if (ctx->at_cap < ldt * N) {
    SI_HIP(ctx, hipStreamSynchronize(ctx->stream));
    dev_free(ctx->d_At);
    ctx->at_cap = 0;
    if (dev_alloc(&ctx->d_At, (size_t)ldt * N) != hipSuccess) return fail(ctx, SI_ERR_NOMEM, "si_construct_finish: allocation of A' failed");
    ctx->at_cap = ldt * N;
  }
  // the Gram kernel reads whole 64-row slabs: rows [K, ldt) of every column must be zero (K differs between constructions)
  SI_HIP(ctx, hipMemsetAsync(ctx->d_At, 0, (size_t)ldt * N * sizeof(double), ctx->stream));
  const size_t need = launch_gram(ctx->stream, ctx->d_At, ldt, K, N, nullptr, nullptr, ctx->num_cu, nullptr);
  if (ctx->gpart_bytes < need) {
    SI_HIP(ctx, hipStreamSynchronize(ctx->stream));
    dev_free(ctx->d_Gpart);
    ctx->gpart_bytes = 0;
    if (hipMalloc(reinterpret_cast<void**>(&ctx->d_Gpart), need) != hipSuccess)
      return fail(ctx, SI_ERR_NOMEM, "si_construct_finish: partial-slab allocation failed");
    ctx->gpart_bytes = need;
  }
  if (ctx->g_cap < N * N) {
    SI_HIP(ctx, hipStreamSynchronize(ctx->stream));
    dev_free(ctx->d_G);
    ctx->g_cap = 0;
    if (dev_alloc(&ctx->d_G, (size_t)N * N) != hipSuccess) return fail(ctx, SI_ERR_NOMEM, "si_construct_finish: G allocation failed");
    ctx->g_cap = N * N;
  }
  {
    ProfScope ps(ctx, SI_K_PUSH, 0.0, 16.0 * (double)N * (double)K);
    launch_transpose(ctx->stream, ctx->d_A, ctx->a_dtype, ctx->ldA, N, K, ctx->d_At, ldt);
  }
  launch_gram(ctx->stream, ctx->d_At, ldt, K, N, ctx->d_Gpart, ctx->d_G, ctx->num_cu, ctx);   // N x N: A A'
  SI_HIP(ctx, hipGetLastError());
  if ((rc = fetch_G(ctx, N)) != SI_OK) return rc;
  std::vector<double> wtop, U;
  if ((rc = top_eigen(ctx, N, M, wtop, U)) != SI_OK) return rc;
  if (!(wtop[0] > 0.0)) return fail(ctx, SI_ERR_BOUNDS, "BoundsError: the deviation matrix is zero (rank 0 < M)");
  if (!(wtop[(size_t)M - 1] > SI_GRAM_ROUTE_MIN * wtop[0])) return SI_OK;   // ill-conditioned: the K x K route with its second stage
  ctx->svals.assign((size_t)M, 0.0);
  for (int m = 0; m < M; ++m) ctx->svals[(size_t)m] = std::sqrt(wtop[(size_t)m]);
  // signs: the right singular vector s_m v_m = A' u_m, largest-magnitude entry positive (the convention of the K x K route)
  const int Mpad = project_mpad(M);
  if ((rc = ensure_pin(ctx, (size_t)N * N * 2 + (size_t)N * Mpad + (size_t)ldt * M + (size_t)N * M)) != SI_OK) return rc;
  double* const Uh = ctx->h_pin + (size_t)N * N * 2;          // N x Mpad, row n contiguous
  double* const Rh = Uh + (size_t)N * Mpad;                    // ldt x M
  double* const Ph = Rh + (size_t)ldt * M;                     // N x M
  std::fill(Uh, Uh + (size_t)N * Mpad, 0.0);
  for (int m = 0; m < M; ++m)
    for (int64_t n = 0; n < N; ++n) Uh[(size_t)n * Mpad + m] = U[(size_t)m * N + n];
  if ((rc = ensure_V(ctx, (size_t)N * Mpad + (size_t)ldt * M)) != SI_OK) return rc;
  double* const dU = ctx->d_V;
  double* const dR = ctx->d_V + (size_t)N * Mpad;
  SI_HIP(ctx, hipMemcpyAsync(dU, Uh, (size_t)N * Mpad * sizeof(double), hipMemcpyHostToDevice, ctx->stream));
  {
    ProfScope ps(ctx, SI_K_PROJECT, 2.0 * (double)N * (double)K * (double)M, (double)K * (double)(N + M) * 8.0);
    launch_project(ctx->stream, ctx->d_At, ldt, K, N, dU, M, Mpad, dR, ldt, ctx->num_cu);   // R = A' U_M  (K x M)
  }
  SI_HIP(ctx, hipGetLastError());
  SI_HIP(ctx, hipMemcpyAsync(Rh, dR, (size_t)ldt * M * sizeof(double), hipMemcpyDeviceToHost, ctx->stream));
  SI_HIP(ctx, hipStreamSynchronize(ctx->stream));
  for (int m = 0; m < M; ++m) {
    const double* v = Rh + (size_t)m * ldt;
    int64_t imax = 0;
    for (int64_t k = 1; k < K; ++k)
      if (std::fabs(v[k]) > std::fabs(v[imax])) imax = k;
    const double f = (v[imax] < 0.0 ? -1.0 : 1.0) * ctx->svals[(size_t)m];
    for (int64_t n = 0; n < N; ++n) Ph[(size_t)m * N + n] = f * U[(size_t)m * N + n];
  }
  if ((rc = alloc_P(ctx, M)) != SI_OK) return rc;
  SI_HIP(ctx, hipMemcpy2DAsync(ctx->d_P, (size_t)ctx->ldA * sizeof(double), Ph, (size_t)N * sizeof(double), (size_t)N * sizeof(double),
                               (size_t)M, hipMemcpyHostToDevice, ctx->stream));
  SI_HIP(ctx, hipStreamSynchronize(ctx->stream));   // the pinned buffers may be rewritten
  *done = true;
  return SI_OK;
}

int32_t si_construct_finish(si_ctx* ctx, int32_t M, double* W_swa_out, double* P_out, double* s_out,
                            int64_t* K_out) {
  CHECK_CTX(ctx);
  if (!ctx->c_active || ctx->K <= 0) return fail(ctx, SI_ERR_STATE, "si_construct_finish: nothing pushed");
  if (M <= 0) return fail(ctx, SI_ERR_INVALID, "si_construct_finish: M must be positive");
  BIND(ctx);
  const int64_t K = ctx->K, N = ctx->N;
  if (K_out) *K_out = K;
  // U[:,1:M] throws BoundsError in the reference when psvd returns fewer than M columns (:65)
  if (M > std::min<int64_t>(N, K))
    return fail(ctx, SI_ERR_BOUNDS, "BoundsError: M exceeds min(N, K), the largest possible rank of the deviation matrix");
  int32_t rc;
  // ---- K > N (the README toy: batchsize 1 x 100 observations x 10 epochs = 1000 deviation columns of 682 weights): the Gram
  // matrix on the SMALLER side, A A' (N x N) instead of A'A (K x K).  Its top eigenpairs are (s^2, U) directly:
  // P = U[:, 1:M] * Diagonal(s[1:M]) (src/subspace_construction.jl:65).  Only when the caller has not asked for the K x K
  // Gram matrix itself (si_construct_gram + all-reduce: a row-sharded construction sums A'A over the ranks, A A' does not add).
  if (!ctx->gram_valid && ctx->refine_stage == 0 && K > N) {
    bool done = false;
    if ((rc = finish_wide(ctx, M, &done)) != SI_OK) return rc;
    if (done) {
      ctx->M_built = M;
      ctx->c_finished = true;
      if (s_out) std::copy(ctx->svals.begin(), ctx->svals.end(), s_out);
      if (W_swa_out)
        SI_HIP(ctx, hipMemcpyAsync(W_swa_out, ctx->d_swa, (size_t)N * sizeof(double), hipMemcpyDeviceToHost, ctx->stream));
      if (P_out)
        SI_HIP(ctx, hipMemcpy2DAsync(P_out, (size_t)N * sizeof(double), ctx->d_P, (size_t)ctx->ldA * sizeof(double),
                                     (size_t)N * sizeof(double), (size_t)M, hipMemcpyDeviceToHost, ctx->stream));
      SI_HIP(ctx, hipStreamSynchronize(ctx->stream));
      return SI_OK;
    }   // (an ill-conditioned A A' falls through to the K x K route, which has the two-stage refinement)
  }
  if (!ctx->gram_valid && (rc = si_construct_gram(ctx)) != SI_OK) return rc;
  const int Mpad = project_mpad(M);
  const size_t v_elems = (size_t)K * Mpad;
  const double* src = ctx->d_A;  // matrix the projection reads: A (Gram route) or B (two-stage route)
  std::vector<double> vcols((size_t)K * M);  // right factor in the basis of `src`, columns m < M
  ctx->svals.assign((size_t)M, 0.0);
  if (ctx->refine_stage == 0) {
    // H1: eigen-decomposition of G on the host (K x K); G comes down into, and V goes up from, pinned memory
    if ((rc = fetch_G(ctx, K)) != SI_OK) return rc;
    std::vector<double> wtop, Vtop;
    if ((rc = top_eigen(ctx, K, M, wtop, Vtop)) != SI_OK) return rc;
    if (!(wtop[0] > 0.0))
      return fail(ctx, SI_ERR_BOUNDS, "BoundsError: the deviation matrix is zero (rank 0 < M)");
    if (wtop[(size_t)M - 1] > SI_GRAM_ROUTE_MIN * wtop[0]) {
      for (int m = 0; m < M; ++m) ctx->svals[(size_t)m] = std::sqrt(wtop[(size_t)m]);
      vcols = Vtop;
    } else if ((rc = si_construct_refine(ctx)) != SI_OK) {
      return rc;
    }
  }
  std::vector<double> vsign;  // vectors the deterministic sign is read from (in the basis of A's columns)
  if (ctx->refine_stage == 1) {
    // second-stage Gram matrix (all-reduced by the caller when rows are sharded) -> scaled Jacobi
    if ((rc = fetch_G(ctx, K)) != SI_OK) return rc;
    std::vector<double> lam2((size_t)K), W((size_t)K * K);
    {
      const auto t0 = std::chrono::steady_clock::now();
      const int jrc = jacobi_eig_psd((int)K, ctx->h_pin, lam2.data(), W.data());
      ctx->stats.ms[SI_K_EIG_HOST] += std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
      ctx->stats.launches[SI_K_EIG_HOST] += 1;
      if (jrc != 0) return fail(ctx, SI_ERR_INVALID, "si_construct_finish: eigensolver did not converge (second-stage Jacobi)");
    }
    // numerical rank like psvd's rtol = 5 eps, widened by the rounding floor of the two products (~sqrt(K) eps)
    const double s1 = lam2[0] > 0.0 ? std::sqrt(lam2[0]) : 0.0;
    const double sM = lam2[(size_t)M - 1] > 0.0 ? std::sqrt(lam2[(size_t)M - 1]) : 0.0;
    if (!(s1 > 0.0) || sM <= 8.0 * std::sqrt((double)K) * SI_EPS * s1)
      return fail(ctx, SI_ERR_BOUNDS, "BoundsError: M exceeds the numerical rank of the deviation matrix (s_M <= ~5 eps s_1)");
    for (int m = 0; m < M; ++m) ctx->svals[(size_t)m] = std::sqrt(lam2[(size_t)m]);
    std::copy(W.begin(), W.begin() + (size_t)K * M, vcols.begin());
    src = ctx->d_B;
    // V_final = V_full * W_M, only for the sign convention
    vsign.assign((size_t)K * M, 0.0);
    for (int m = 0; m < M; ++m)
      for (int64_t j = 0; j < K; ++j) {
        const double wjm = W[(size_t)m * K + j];
        const double* vj = ctx->vfull.data() + (size_t)j * K;
        double* dst = vsign.data() + (size_t)m * K;
        for (int64_t k = 0; k < K; ++k) dst[k] += vj[k] * wjm;
      }
  }
  // V_M with a deterministic sign (largest-magnitude entry of the right singular vector positive)
  double* const V = ctx->h_pin + (size_t)K * K * 2;
  std::fill(V, V + v_elems, 0.0);
  for (int m = 0; m < M; ++m) {
    const double* vs = (vsign.empty() ? vcols.data() : vsign.data()) + (size_t)m * K;
    int64_t imax = 0;
    for (int64_t k = 1; k < K; ++k)
      if (std::fabs(vs[k]) > std::fabs(vs[imax])) imax = k;
    const double sgn = vs[imax] < 0.0 ? -1.0 : 1.0;
    const double* v = vcols.data() + (size_t)m * K;
    for (int64_t k = 0; k < K; ++k) V[(size_t)k * Mpad + m] = sgn * v[k];
  }
  if ((rc = alloc_P(ctx, M)) != SI_OK) return rc;
  if ((rc = ensure_V(ctx, std::max(v_elems, (size_t)K * project_mpad((int)K) * (ctx->refine_stage ? 1 : 0)))) != SI_OK) return rc;
  SI_HIP(ctx, hipMemcpyAsync(ctx->d_V, V, v_elems * sizeof(double), hipMemcpyHostToDevice, ctx->stream));
  {
    ProfScope ps(ctx, SI_K_PROJECT, 2.0 * (double)N * (double)K * (double)M,
                 (double)N * ((double)K * ((src == ctx->d_A && ctx->a_dtype == SI_F32) ? 4.0 : 8.0) + (double)M * 8.0));
    launch_project(ctx->stream, src, ctx->ldA, N, K, ctx->d_V, M, Mpad, ctx->d_P, ctx->ldA, ctx->num_cu,
                   src == ctx->d_A ? ctx->a_dtype : SI_F64);
  }
  SI_HIP(ctx, hipGetLastError());
  SI_HIP(ctx, hipStreamSynchronize(ctx->stream));  // the pinned V may be rewritten by the next finish
  ctx->M_built = M;
  ctx->c_finished = true;
  if (s_out) std::copy(ctx->svals.begin(), ctx->svals.end(), s_out);
  if (W_swa_out)
    SI_HIP(ctx, hipMemcpyAsync(W_swa_out, ctx->d_swa, (size_t)N * sizeof(double), hipMemcpyDeviceToHost, ctx->stream));
  if (P_out)
    SI_HIP(ctx, hipMemcpy2DAsync(P_out, (size_t)N * sizeof(double), ctx->d_P, (size_t)ctx->ldA * sizeof(double),
                                 (size_t)N * sizeof(double), (size_t)M, hipMemcpyDeviceToHost, ctx->stream));
  SI_HIP(ctx, hipStreamSynchronize(ctx->stream));
  return SI_OK;
}

int32_t si_construct_get_result(si_ctx* ctx, double* W_swa_out, double* P_out, double* s_out, int64_t* N_out, int32_t* M_out) {
  CHECK_CTX(ctx);
  if (!ctx->c_finished) return fail(ctx, SI_ERR_STATE, "si_construct_get_result: no finished construction");
  BIND(ctx);
  const int64_t N = ctx->N;
  const int32_t M = ctx->M_built;
  if (N_out) *N_out = N;
  if (M_out) *M_out = M;
  if (s_out) std::copy(ctx->svals.begin(), ctx->svals.end(), s_out);
  if (W_swa_out) SI_HIP(ctx, hipMemcpyAsync(W_swa_out, ctx->d_swa, (size_t)N * sizeof(double), hipMemcpyDeviceToHost, ctx->stream));
  if (P_out)
    SI_HIP(ctx, hipMemcpy2DAsync(P_out, (size_t)N * sizeof(double), ctx->d_P, (size_t)ctx->ldA * sizeof(double),
                                 (size_t)N * sizeof(double), (size_t)M, hipMemcpyDeviceToHost, ctx->stream));
  SI_HIP(ctx, hipStreamSynchronize(ctx->stream));
  return SI_OK;
}

int32_t si_construct_get_A(si_ctx* ctx, int64_t k0, int64_t nk, double* A_out) {
  CHECK_CTX(ctx);
  if (!ctx->c_active) return fail(ctx, SI_ERR_STATE, "si_construct_get_A: no construction in progress");
  if (k0 < 0 || nk < 0 || k0 + nk > ctx->K || !A_out) return fail(ctx, SI_ERR_INVALID, "si_construct_get_A: bad range");
  BIND(ctx);
  if (nk == 0) return SI_OK;
  if (ctx->a_dtype == SI_F32) {   // fp32 storage: read the floats back and widen them (exact)
    std::vector<float> tmp((size_t)ctx->N * (size_t)nk);
    SI_HIP(ctx, hipMemcpy2DAsync(tmp.data(), (size_t)ctx->N * sizeof(float), reinterpret_cast<const float*>(ctx->d_A) + k0 * ctx->ldA,
                                 (size_t)ctx->ldA * sizeof(float), (size_t)ctx->N * sizeof(float), (size_t)nk, hipMemcpyDeviceToHost,
                                 ctx->stream));
    SI_HIP(ctx, hipStreamSynchronize(ctx->stream));
    for (size_t i = 0; i < tmp.size(); ++i) A_out[i] = (double)tmp[i];
    return SI_OK;
  }
  SI_HIP(ctx, hipMemcpy2DAsync(A_out, (size_t)ctx->N * sizeof(double), ctx->d_A + k0 * ctx->ldA,
                               (size_t)ctx->ldA * sizeof(double), (size_t)ctx->N * sizeof(double), (size_t)nk,
                               hipMemcpyDeviceToHost, ctx->stream));
  SI_HIP(ctx, hipStreamSynchronize(ctx->stream));
  return SI_OK;
}

}  // extern "C"
