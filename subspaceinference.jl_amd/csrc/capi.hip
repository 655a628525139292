// C ABI of libsubspace_hip.so (see include/subspace_hip.h for the contract and the reference lines each
// entry point replaces).  Host-side orchestration only: every arithmetic step of the hot path runs in the
// HIP kernels of kernels_*.hip, except the K x K symmetric eigensolve (eig.cpp, host, K ~ 100).
// There is no CPU fallback anywhere in this file.
#include <algorithm>
#include <cmath>
#include <chrono>
#include <cstdlib>
#include <cstring>
#include <new>
#include <vector>

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

// ---- memory helpers ----------------------------------------------------------------------------
template <typename T>
static hipError_t dev_alloc(T** p, size_t count) {
  *p = nullptr;
  if (count == 0) count = 1;
  return hipMalloc(reinterpret_cast<void**>(p), count * sizeof(T));
}
template <typename T>
static void dev_free(T*& p) {
  if (p) (void)hipFree((void*)p);
  p = nullptr;
}

static void free_infer(Ctx* c);

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

static void free_infer(Ctx* c) {
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

#define CHECK_CTX(ctx) \
  if (!(ctx)) return SI_ERR_INVALID
#define BIND(ctx) SI_HIP(ctx, hipSetDevice((ctx)->device))

static void free_push_staging(si_ctx* ctx);
static void free_wstream(si_ctx* ctx);

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
static void fused_fill_program(const si_ctx* ctx, ChainFusedPlan& fp) {
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
  if (compute_dtype == SI_F32 && plan.has_conv)
    return fail(ctx, SI_ERR_INVALID, "si_infer_setup: compute_dtype = SI_F32 is implemented for Dense chains; Conv / MaxPool / flatten chains compute in SI_F64");
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
      (ctx->f32 && dev_alloc(&ctx->d_X32, (size_t)pad_ld((int64_t)in_dim * B)) != hipSuccess) ||
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
  if (ctx->f32) launch_narrow_f32(ctx->stream, ctx->d_X, ctx->d_X32, (int64_t)in_dim * B);   // X rounded to fp32 once
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

static int32_t ensure_chains(si_ctx* ctx, int32_t C) {
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
static int32_t eval_density(si_ctx* ctx, int c0, int nc, const double** yhat_out) {
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
        launch_chain_fused(ctx->stream, fp, nb, wave_tiles, lds, ctx->d_w, ldw, ctx->d_X, ctx->d_yhat, d, nc);
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
static int32_t eval_density_all(si_ctx* ctx, int C) {
  const int w = std::max(1, ctx->fw_slots);
  for (int c0 = 0; c0 < C; c0 += w) {
    const int32_t rc = eval_density(ctx, c0, std::min(w, C - c0), nullptr);
    if (rc != SI_OK) return rc;
  }
  return SI_OK;
}

static double mvnormal_c0(double d, double sigma) {
  // Distributions.mvnormal_c0: -(d*log(2pi) + logdet(Sigma))/2 with logdet = d*log(sigma^2)
  return -(d * std::log(2.0 * 3.14159265358979323846) + d * std::log(sigma * sigma)) / 2.0;
}

// log N(w; 0, sigma_p^2 I) = c0p - ||w||^2 / (2 sigma_p^2): the term the reference writes AFTER its `return` (Q4)
static double prior_c0(const si_ctx* ctx) { return ctx->sigma_p > 0.0 ? mvnormal_c0((double)ctx->iN, ctx->sigma_p) : 0.0; }

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

// ---- streamed output map (a13, src/space_inference.jl:125: `map(z -> W_swa + P*z.params, chm)`) ------------------------
// K4 has already produced W_swa + P z' for every proposal (d_w); the weight vector of sample t is that vector when the
// proposal was accepted and the previous sample's otherwise.  A select kernel keeps the CURRENT weights of every chain in
// a small device ring (8 bytes read + 8 written per weight, ~4 us at cfg2 -- instead of a second K4 pass over P), a DMA
// on the second stream moves ring slot t into pinned memory while transition t+1 computes, and the host copy pool
// moves it into the caller's array R-1 transitions later.  The chain never waits for PCIe.
static constexpr int SI_WRING = 4;

static void free_wstream(si_ctx* ctx) {
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
    if (nb != 0) {
      if (ctx->gridsync_chains < C) {
        dev_free(ctx->d_gridsync);
        ctx->gridsync_chains = 0;
        if (dev_alloc(&ctx->d_gridsync, (size_t)32 * ((size_t)C + 1)) != hipSuccess) return fail(ctx, SI_ERR_NOMEM, std::string(who) + ": allocation failed");
        ctx->gridsync_chains = C;
      }
      a.swa = ctx->i_swa; a.P = ctx->i_P; a.X = ctx->d_X; a.Y = ctx->d_Y;
      a.wbuf = ctx->d_w; a.w_stride = ldw;
      a.ybuf = ctx->d_yhat; a.y_stride = (int64_t)ctx->out_dim * ctx->B;
      a.cnt = ctx->d_gridsync; a.status = ctx->d_gridsync + (size_t)32 * (size_t)C;
      a.Z_out = dZ; a.lp_out = dlp; a.nacc_out = ctx->d_nacc;
      a.ldP = ctx->ldP; a.itr = itr; a.seed = seed; a.sigma_z = sigma_z; a.c0 = c0; a.sigma2 = s2;
      a.N = (int)N; a.chain_id0 = chain_id0;
      hipError_t e = hipMemsetAsync(ctx->d_gridsync, 0, (size_t)32 * ((size_t)C + 1) * sizeof(unsigned), ctx->stream);
      if (e == hipSuccess) {
        const double fl = 2.0 * (double)N * (double)ctx->B * (double)itr * C;
        ProfScope ps(ctx, SI_K_RWMH, fl, 0.0);
        e = launch_chain_grid(ctx->stream, a, nb, C, lds);
      }
      std::vector<int64_t> nacc((size_t)C);
      unsigned status = 0;
      if (e == hipSuccess) e = hipMemcpyAsync(&status, a.status, sizeof(unsigned), hipMemcpyDeviceToHost, ctx->stream);
      if (e == hipSuccess && Z_out) e = hipMemcpyAsync(Z_out, dZ, (size_t)M * itr * C * sizeof(double), hipMemcpyDeviceToHost, ctx->stream);
      if (e == hipSuccess && lp_out) e = hipMemcpyAsync(lp_out, dlp, (size_t)itr * C * sizeof(double), hipMemcpyDeviceToHost, ctx->stream);
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
  if (on < 0 || on > 2) return fail(ctx, SI_ERR_INVALID, "si_set_chain_loop: 0 (one launch per layer and step), 1 (automatic) or 2 (fused launches, no device-resident loop)");
  ctx->chain_mode = on;
  ctx->chain_loop_enabled = on != 0;
  return SI_OK;
}

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
