// R1 -- the RCCL communicator behind the C ABI.  One process per GPU, one si_ctx per process, one
// RCCL communicator per ctx; every collective below runs IN PLACE on a buffer the library owns, stream-ordered on the
// ctx's stream (no host staging, no synchronisation of its own), so a Julia host reaches more than one GPU through
// `ccall` alone.  The reference is single-process (zero collectives); the call sites these collectives
// parallelise are src/subspace_construction.jl:45-52,63 (row-sharded SWA / deviation / Gram) and
// src/space_inference.jl:94 (data-sharded log-likelihood).
//
// RCCL is bound at RUN TIME (dlopen), not at link time: a PyTorch host process already carries its own librccl.so.1
// (bundled in the wheel, built against the HIP runtime the wheel bundles) and a second copy of the library in one
// process is asking for trouble; dlopen by soname returns the copy that is already loaded, and loads /opt/rocm's in a
// host (Julia) that has none.  Without any librccl the si_comm_* entry points fail with a message; nothing else in the
// library depends on it.
#include <dlfcn.h>
#include <rccl/rccl.h>  // types and prototypes only

#include <algorithm>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>

#include "si_internal.h"

namespace si {

namespace {

struct Rccl {
  void* handle = nullptr;
  std::string err, path;
  decltype(&ncclGetUniqueId) GetUniqueId = nullptr;
  decltype(&ncclCommInitRank) CommInitRank = nullptr;
  decltype(&ncclCommDestroy) CommDestroy = nullptr;
  decltype(&ncclAllReduce) AllReduce = nullptr;
  decltype(&ncclBroadcast) Broadcast = nullptr;
  decltype(&ncclAllGather) AllGather = nullptr;
  decltype(&ncclGroupStart) GroupStart = nullptr;
  decltype(&ncclGroupEnd) GroupEnd = nullptr;
  decltype(&ncclGetErrorString) GetErrorString = nullptr;
  decltype(&ncclGetVersion) GetVersion = nullptr;
};

template <typename F>
bool bind(Rccl& r, F& fn, const char* name) {
  fn = reinterpret_cast<F>(dlsym(r.handle, name));
  if (!fn) r.err = std::string("librccl (") + r.path + ") has no symbol " + name;
  return fn != nullptr;
}

Rccl load_rccl() {
  Rccl r;
  std::vector<std::string> names;
  if (const char* e = getenv("SI_RCCL_LIB")) names.push_back(e);  // explicit path of the RCCL build to use
  names.insert(names.end(), {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"});
  for (const auto& n : names) {
    r.handle = dlopen(n.c_str(), RTLD_NOW | RTLD_LOCAL);
    if (r.handle) {
      r.path = n;
      break;
    }
  }
  if (!r.handle) {
    const char* de = dlerror();
    r.err = std::string("RCCL is not available: dlopen(librccl.so.1) failed (") + (de ? de : "?") +
            "); multi-GPU entry points need ROCm's librccl on the library path (or SI_RCCL_LIB=/path/to/librccl.so)";
    return r;
  }
  const bool ok = bind(r, r.GetUniqueId, "ncclGetUniqueId") && bind(r, r.CommInitRank, "ncclCommInitRank") &&
                  bind(r, r.CommDestroy, "ncclCommDestroy") && bind(r, r.AllReduce, "ncclAllReduce") &&
                  bind(r, r.Broadcast, "ncclBroadcast") && bind(r, r.AllGather, "ncclAllGather") &&
                  bind(r, r.GroupStart, "ncclGroupStart") && bind(r, r.GroupEnd, "ncclGroupEnd") &&
                  bind(r, r.GetErrorString, "ncclGetErrorString") && bind(r, r.GetVersion, "ncclGetVersion");
  if (!ok) {
    dlclose(r.handle);
    r.handle = nullptr;
  }
  return r;
}

Rccl& rccl() {
  static Rccl r = load_rccl();
  return r;
}

inline ncclComm_t comm_of(Ctx* c) { return static_cast<ncclComm_t>(c->comm); }

}  // namespace

#define SI_NCCL(c, expr)                                                                                   \
  do {                                                                                                     \
    ncclResult_t r_ = (expr);                                                                              \
    if (r_ != ncclSuccess) return si::fail((c), SI_ERR_COMM, std::string(#expr) + ": " + rccl().GetErrorString(r_)); \
  } while (0)

void comm_release(Ctx* c) {
  if (c->comm) {
    (void)hipStreamSynchronize(c->stream);
    (void)rccl().CommDestroy(comm_of(c));
  }
  c->comm = nullptr;
  c->comm_world = 0;
  c->comm_rank = 0;
  if (c->d_commtmp) (void)hipFree(c->d_commtmp);
  c->d_commtmp = nullptr;
  if (c->d_gathertmp) (void)hipFree(c->d_gathertmp);
  c->d_gathertmp = nullptr;
  c->gathertmp_cap = 0;
}

static int32_t need_comm(Ctx* c, const char* who) {
  if (!c->comm) return fail(c, SI_ERR_STATE, std::string(who) + ": no communicator (call si_comm_init_rank first)");
  return SI_OK;
}

constexpr int64_t COMM_TMP_ELEMS = 4096;  // device scratch of the host-value collectives

}  // namespace si

using namespace si;

extern "C" {

int32_t si_comm_unique_id(uint8_t* id_out) {
  if (!id_out) return fail(nullptr, SI_ERR_INVALID, "si_comm_unique_id: id_out is NULL");
  static_assert(SI_COMM_ID_BYTES == NCCL_UNIQUE_ID_BYTES, "SI_COMM_ID_BYTES must equal NCCL_UNIQUE_ID_BYTES");
  Rccl& r = rccl();
  if (!r.handle) return fail(nullptr, SI_ERR_COMM, "si_comm_unique_id: " + r.err);
  ncclUniqueId id;
  const ncclResult_t rc = r.GetUniqueId(&id);
  if (rc != ncclSuccess) return fail(nullptr, SI_ERR_COMM, std::string("si_comm_unique_id: ncclGetUniqueId: ") + r.GetErrorString(rc));
  std::memcpy(id_out, id.internal, SI_COMM_ID_BYTES);
  return SI_OK;
}

int32_t si_comm_init_rank(si_ctx* ctx, int32_t world, int32_t rank, const uint8_t* id) {
  if (!ctx) return SI_ERR_INVALID;
  if (!id || world <= 0 || rank < 0 || rank >= world) return fail(ctx, SI_ERR_INVALID, "si_comm_init_rank: need 0 <= rank < world and the 128-byte id");
  if (ctx->comm) return fail(ctx, SI_ERR_STATE, "si_comm_init_rank: this ctx already has a communicator (si_comm_destroy first)");
  Rccl& r = rccl();
  if (!r.handle) return fail(ctx, SI_ERR_COMM, "si_comm_init_rank: " + r.err);
  SI_HIP(ctx, hipSetDevice(ctx->device));  // the communicator is bound to the ctx's GPU
  ncclUniqueId uid;
  std::memcpy(uid.internal, id, SI_COMM_ID_BYTES);
  ncclComm_t comm = nullptr;
  SI_NCCL(ctx, r.CommInitRank(&comm, world, uid, rank));
  ctx->comm = comm;
  ctx->comm_world = world;
  ctx->comm_rank = rank;
  // `world` processes of the library now share this host's CPUs (one node, one process per GPU): the host copy pool of
  // this process keeps to its share of the CPU quota (8 ranks on a 16-CPU quota: one copy thread each, not eight)
  host_copy_set_share(world);
  if (hipMalloc(reinterpret_cast<void**>(&ctx->d_commtmp), (COMM_TMP_ELEMS + 8) * sizeof(double)) != hipSuccess) {   // (+ the status word of comm_agree)
    comm_release(ctx);
    return fail(ctx, SI_ERR_NOMEM, "si_comm_init_rank: scratch allocation failed");
  }
  return SI_OK;
}

int32_t si_comm_destroy(si_ctx* ctx) {
  if (!ctx) return SI_ERR_INVALID;
  SI_HIP(ctx, hipSetDevice(ctx->device));
  comm_release(ctx);
  return SI_OK;
}

int32_t si_comm_info(si_ctx* ctx, int32_t* world_out, int32_t* rank_out, int32_t* rccl_version_out) {
  if (!ctx) return SI_ERR_INVALID;
  if (world_out) *world_out = ctx->comm ? ctx->comm_world : 0;
  if (rank_out) *rank_out = ctx->comm ? ctx->comm_rank : 0;
  if (rccl_version_out) {
    int v = 0;
    if (rccl().handle) (void)rccl().GetVersion(&v);
    *rccl_version_out = v;
  }
  return SI_OK;
}

int32_t si_row_shard(int64_t n_total, int32_t rank, int32_t world, int64_t* r0_out, int64_t* r1_out) {
  if (n_total < 0 || world <= 0 || rank < 0 || rank >= world || !r0_out || !r1_out) return SI_ERR_INVALID;
  const int64_t align = 32;  // 256 B: every shard keeps the kernels' 16-byte access alignment
  const int64_t blocks = (n_total + align - 1) / align;
  const int64_t base = blocks / world, rem = blocks % world;
  const int64_t b0 = rank * base + std::min<int64_t>(rank, rem);
  const int64_t b1 = b0 + base + (rank < rem ? 1 : 0);
  *r0_out = std::min(b0 * align, n_total);
  *r1_out = std::min(b1 * align, n_total);
  return SI_OK;
}

// ---- host-value collectives (timings, losses, rank counts): H2D -> RCCL -> D2H, synchronous -------------------------
int32_t si_comm_allreduce_host(si_ctx* ctx, double* inout, int64_t n, int32_t op) {
  if (!ctx) return SI_ERR_INVALID;
  int32_t rc = need_comm(ctx, "si_comm_allreduce_host");
  if (rc != SI_OK) return rc;
  if (!inout || n <= 0 || n > COMM_TMP_ELEMS || (op != SI_COMM_SUM && op != SI_COMM_MAX))
    return fail(ctx, SI_ERR_INVALID, "si_comm_allreduce_host: need 1 <= n <= 4096 values and op = SI_COMM_SUM / SI_COMM_MAX");
  SI_HIP(ctx, hipSetDevice(ctx->device));
  SI_HIP(ctx, hipMemcpyAsync(ctx->d_commtmp, inout, (size_t)n * sizeof(double), hipMemcpyHostToDevice, ctx->stream));
  SI_NCCL(ctx, rccl().AllReduce(ctx->d_commtmp, ctx->d_commtmp, (size_t)n, ncclFloat64, op == SI_COMM_SUM ? ncclSum : ncclMax,
                                comm_of(ctx), ctx->stream));
  SI_HIP(ctx, hipMemcpyAsync(inout, ctx->d_commtmp, (size_t)n * sizeof(double), hipMemcpyDeviceToHost, ctx->stream));
  SI_HIP(ctx, hipStreamSynchronize(ctx->stream));
  return SI_OK;
}

// Collective error agreement: an entry point that validates PER RANK and returns before its data collective
// would leave every other rank blocked inside RCCL for good.  Every rank therefore first contributes its local status to
// ONE small all-reduce (max) and only then issues the data collective -- or none of them does: the failing rank returns its
// own error, the others SI_ERR_COMM.  Synchronous (one 8-byte all-reduce + a stream sync): used by the one-off collectives
// and by si_train_step_dp, never inside the per-transition loop of a sharded chain.
static int32_t comm_agree(si_ctx* ctx, int32_t local_rc, const char* who) {
  double flag = local_rc == SI_OK ? 0.0 : 1.0;
  double* slot = ctx->d_commtmp + COMM_TMP_ELEMS;
  const std::string own = ctx->err;
  // A local HIP failure in front of the all-reduce must NOT make this rank skip it (its peers would wait inside theirs for
  // good -- ADVICE r4): the all-reduce is entered regardless, on whatever the slot holds; the failing rank reports its HIP
  // error below, and if its flag never reached the device its peers learn of the failure from the next collective's own
  // error handling rather than from a hang here.
  hipError_t e = hipSetDevice(ctx->device);
  if (e == hipSuccess) e = hipMemcpyAsync(slot, &flag, sizeof(double), hipMemcpyHostToDevice, ctx->stream);
  const ncclResult_t nr = rccl().AllReduce(slot, slot, 1, ncclFloat64, ncclMax, comm_of(ctx), ctx->stream);
  if (e == hipSuccess && nr == ncclSuccess) e = hipMemcpyAsync(&flag, slot, sizeof(double), hipMemcpyDeviceToHost, ctx->stream);
  const hipError_t e2 = hipStreamSynchronize(ctx->stream);
  if (local_rc != SI_OK) {   // this rank's own failure is what it reports
    ctx->err = own;
    return local_rc;
  }
  if (nr != ncclSuccess) return fail(ctx, SI_ERR_COMM, std::string(who) + ": status all-reduce: " + rccl().GetErrorString(nr));
  if (e != hipSuccess || e2 != hipSuccess) return fail(ctx, SI_ERR_HIP, std::string(who) + ": " + hipGetErrorString(e != hipSuccess ? e : e2));
  if (flag != 0.0) return fail(ctx, SI_ERR_COMM, std::string(who) + ": another rank failed before the collective (nothing was exchanged)");
  return SI_OK;
}

int32_t si_comm_barrier(si_ctx* ctx) {
  double one = 1.0;
  return si_comm_allreduce_host(ctx, &one, 1, SI_COMM_SUM);  // completes only when every rank's stream has reached it
}

int32_t si_comm_allgather_host(si_ctx* ctx, const double* send, int64_t n, double* recv) {
  if (!ctx) return SI_ERR_INVALID;
  int32_t rc = need_comm(ctx, "si_comm_allgather_host");
  if (rc != SI_OK) return rc;
  if (!send || !recv || n <= 0) return fail(ctx, SI_ERR_INVALID, "si_comm_allgather_host: bad argument");
  SI_HIP(ctx, hipSetDevice(ctx->device));
  const size_t W = (size_t)ctx->comm_world;
  const size_t need = (W + 1) * (size_t)n;
  if (ctx->gathertmp_cap < need) {   // scratch kept between calls (the chains' (Z, lp) gathers repeat with the same size)
    SI_HIP(ctx, hipStreamSynchronize(ctx->stream));
    if (ctx->d_gathertmp) (void)hipFree(ctx->d_gathertmp);
    ctx->d_gathertmp = nullptr;
    ctx->gathertmp_cap = 0;
    if (hipMalloc(reinterpret_cast<void**>(&ctx->d_gathertmp), need * sizeof(double)) != hipSuccess)
      return fail(ctx, SI_ERR_NOMEM, "si_comm_allgather_host: allocation failed");
    ctx->gathertmp_cap = need;
  }
  double* tmp = ctx->d_gathertmp;
  hipError_t e = hipMemcpyAsync(tmp + W * (size_t)n, send, (size_t)n * sizeof(double), hipMemcpyHostToDevice, ctx->stream);
  ncclResult_t nr = ncclSuccess;
  if (e == hipSuccess) nr = rccl().AllGather(tmp + W * (size_t)n, tmp, (size_t)n, ncclFloat64, comm_of(ctx), ctx->stream);
  if (e == hipSuccess && nr == ncclSuccess)
    e = hipMemcpyAsync(recv, tmp, W * (size_t)n * sizeof(double), hipMemcpyDeviceToHost, ctx->stream);
  const hipError_t e2 = hipStreamSynchronize(ctx->stream);
  if (nr != ncclSuccess) return fail(ctx, SI_ERR_COMM, std::string("si_comm_allgather_host: ncclAllGather: ") + rccl().GetErrorString(nr));
  if (e != hipSuccess || e2 != hipSuccess)
    return fail(ctx, SI_ERR_HIP, std::string("si_comm_allgather_host: ") + hipGetErrorString(e != hipSuccess ? e : e2));
  return SI_OK;
}

// ---- in-place collectives on the library's own buffers ---------------------------------------------------------------
// row-sharded construction (src/subspace_construction.jl:63 on row blocks): G <- sum over ranks of the local A'A
// (after si_construct_refine: the second-stage Gram matrix), K x K fp64 -- 80 KB at K = 100, latency-bound
int32_t si_construct_allreduce_gram(si_ctx* ctx) {
  if (!ctx) return SI_ERR_INVALID;
  int32_t rc = need_comm(ctx, "si_construct_allreduce_gram");
  if (rc != SI_OK) return rc;
  if (!ctx->gram_valid) return fail(ctx, SI_ERR_STATE, "si_construct_allreduce_gram: call si_construct_gram first");
  SI_HIP(ctx, hipSetDevice(ctx->device));
  SI_NCCL(ctx, rccl().AllReduce(ctx->d_G, ctx->d_G, (size_t)ctx->K * (size_t)ctx->K, ncclFloat64, ncclSum, comm_of(ctx), ctx->stream));
  return SI_OK;
}

// data-sharded density (src/space_inference.jl:94 on column blocks of X, Y): the nchains partial sums of squared errors
// between si_rwmh_step_eval(NULL) and si_rwmh_step_accept(NULL)
int32_t si_rwmh_allreduce_sse(si_ctx* ctx) {
  if (!ctx) return SI_ERR_INVALID;
  int32_t rc = need_comm(ctx, "si_rwmh_allreduce_sse");
  if (rc != SI_OK) return rc;
  if (!ctx->sw_Z || !ctx->sw_evaluated) return fail(ctx, SI_ERR_STATE, "si_rwmh_allreduce_sse: no evaluated step pending (si_rwmh_step_eval first)");
  SI_HIP(ctx, hipSetDevice(ctx->device));
  SI_NCCL(ctx, rccl().AllReduce(ctx->d_sse, ctx->d_sse, (size_t)ctx->sw_C, ncclFloat64, ncclSum, comm_of(ctx), ctx->stream));
  return SI_OK;
}

// data-parallel training step (src/subspace_construction.jl:39-43 on shares of the batch): the N-double gradient and the
// local sum of squared errors, one grouped launch
int32_t si_train_allreduce_grad(si_ctx* ctx, double* sse_total_out) {
  if (!ctx) return SI_ERR_INVALID;
  int32_t rc = need_comm(ctx, "si_train_allreduce_grad");
  if (rc != SI_OK) return rc;
  TrainState* t = ctx->train;
  if (!t || !t->grad_ready) return fail(ctx, SI_ERR_STATE, "si_train_allreduce_grad: no gradient pending (call si_train_grad)");
  SI_HIP(ctx, hipSetDevice(ctx->device));
  SI_NCCL(ctx, rccl().GroupStart());
  const ncclResult_t r1 = rccl().AllReduce(t->gw, t->gw, (size_t)t->N, ncclFloat64, ncclSum, comm_of(ctx), ctx->stream);
  const ncclResult_t r2 = rccl().AllReduce(t->sse, t->sse, 1, ncclFloat64, ncclSum, comm_of(ctx), ctx->stream);
  SI_NCCL(ctx, rccl().GroupEnd());
  SI_NCCL(ctx, r1);
  SI_NCCL(ctx, r2);
  if (sse_total_out) {
    SI_HIP(ctx, hipMemcpyAsync(sse_total_out, t->sse, sizeof(double), hipMemcpyDeviceToHost, ctx->stream));
    SI_HIP(ctx, hipStreamSynchronize(ctx->stream));
  }
  return SI_OK;
}

int32_t si_train_step_dp(si_ctx* ctx, const int64_t* idx, int64_t nb, int64_t nb_total, double* loss_out) {
  if (!ctx) return SI_ERR_INVALID;
  int32_t rc = need_comm(ctx, "si_train_step_dp");
  if (rc != SI_OK) return rc;
  // (a rank whose forward / reverse sweep failed must not leave the others inside the gradient all-reduce)
  if ((rc = comm_agree(ctx, si_train_grad(ctx, idx, nb, nb_total, nullptr), "si_train_step_dp")) != SI_OK) return rc;
  double sse = 0.0;
  if ((rc = si_train_allreduce_grad(ctx, loss_out ? &sse : nullptr)) != SI_OK) return rc;
  if ((rc = si_train_apply(ctx)) != SI_OK) return rc;
  if (loss_out) *loss_out = sse / ((double)ctx->train->out_dim * (double)nb_total);
  return SI_OK;
}

// independent chains (cfg3): (W_swa, P, s) of the construction finished on `root` -> every rank, device to device.
// A receiving ctx ends up holding a finished construction of its own (si_infer_setup with W_swa = P = NULL uses it in
// place); 168 MB at cfg2, once.
int32_t si_bcast_subspace(si_ctx* ctx, int32_t root, int64_t N, int32_t M) {
  if (!ctx) return SI_ERR_INVALID;
  int32_t rc = need_comm(ctx, "si_bcast_subspace");
  if (rc != SI_OK) return rc;
  if (root < 0 || root >= ctx->comm_world || N <= 0 || M <= 0 || M > COMM_TMP_ELEMS) return fail(ctx, SI_ERR_INVALID, "si_bcast_subspace: bad root / N / M");
  SI_HIP(ctx, hipSetDevice(ctx->device));
  const int64_t ld = pad_ld(N);
  // local preparation (root: a finished construction of this shape; receivers: buffers to receive into), then agreement
  auto prepare = [&]() -> int32_t {
    if (ctx->comm_rank == root) {
      if (!ctx->c_finished || ctx->N != N || ctx->M_built != M)
        return fail(ctx, SI_ERR_STATE, "si_bcast_subspace: the root has no finished construction of this N / M");
      SI_HIP(ctx, hipMemcpyAsync(ctx->d_commtmp, ctx->svals.data(), (size_t)M * sizeof(double), hipMemcpyHostToDevice, ctx->stream));
    } else if (!(ctx->c_finished && !ctx->c_active && ctx->N == N && ctx->M_built == M && ctx->d_swa && ctx->d_P)) {
      return construct_adopt(ctx, N, M);
    }
    return SI_OK;
  };
  if ((rc = comm_agree(ctx, prepare(), "si_bcast_subspace")) != SI_OK) return rc;
  SI_NCCL(ctx, rccl().GroupStart());
  const ncclResult_t r1 = rccl().Broadcast(ctx->d_swa, ctx->d_swa, (size_t)ld, ncclFloat64, root, comm_of(ctx), ctx->stream);
  const ncclResult_t r2 = rccl().Broadcast(ctx->d_P, ctx->d_P, (size_t)ld * (size_t)M, ncclFloat64, root, comm_of(ctx), ctx->stream);
  const ncclResult_t r3 = rccl().Broadcast(ctx->d_commtmp, ctx->d_commtmp, (size_t)M, ncclFloat64, root, comm_of(ctx), ctx->stream);
  SI_NCCL(ctx, rccl().GroupEnd());
  SI_NCCL(ctx, r1);
  SI_NCCL(ctx, r2);
  SI_NCCL(ctx, r3);
  if (ctx->comm_rank != root) {
    ctx->svals.assign((size_t)M, 0.0);
    SI_HIP(ctx, hipMemcpyAsync(ctx->svals.data(), ctx->d_commtmp, (size_t)M * sizeof(double), hipMemcpyDeviceToHost, ctx->stream));
    SI_HIP(ctx, hipStreamSynchronize(ctx->stream));
  }
  return SI_OK;
}

// row-sharded construction (cfg4 / cfg5): every rank holds rows si_row_shard(n_total, rank, world) of W_swa and P after
// si_construct_finish; assemble the FULL (W_swa, P) on every rank, device to device (column chunks bound the temporary
// to ~1 GiB: P is 26 GB at cfg5).  The ctx then holds a finished construction of n_total rows, like si_bcast_subspace.
int32_t si_construct_allgather(si_ctx* ctx, int64_t n_total) {
  if (!ctx) return SI_ERR_INVALID;
  int32_t rc = need_comm(ctx, "si_construct_allgather");
  if (rc != SI_OK) return rc;
  if (!ctx->c_finished) return fail(ctx, SI_ERR_STATE, "si_construct_allgather: no finished construction");
  const int W = ctx->comm_world, me = ctx->comm_rank;
  std::vector<int64_t> r0((size_t)W), r1((size_t)W);
  int64_t mx = 0;
  for (int r = 0; r < W; ++r) {
    (void)si_row_shard(n_total, r, W, &r0[(size_t)r], &r1[(size_t)r]);
    mx = std::max(mx, r1[(size_t)r] - r0[(size_t)r]);
  }
  const int64_t n_loc = r1[(size_t)me] - r0[(size_t)me];
  SI_HIP(ctx, hipSetDevice(ctx->device));
  const int32_t M = ctx->M_built;
  const int64_t ld_loc = ctx->ldA, ld = pad_ld(n_total);
  mx = (mx + 1) & ~(int64_t)1;
  const int64_t ncols = (int64_t)M + 1;  // column M = W_swa
  const int64_t chunk = std::max<int64_t>(1, std::min<int64_t>(ncols, ((int64_t)1 << 27) / std::max<int64_t>(1, (int64_t)W * mx)));
  double *send = nullptr, *recv = nullptr, *w_full = nullptr, *p_full = nullptr;
  auto cleanup = [&] {
    if (send) (void)hipFree(send);
    if (recv) (void)hipFree(recv);
  };
  // validation and allocation are LOCAL and may fail on one rank only: agree before the first ncclAllGather
  int32_t lrc = SI_OK;
  if (ctx->N != n_loc)
    lrc = fail(ctx, SI_ERR_INVALID, "si_construct_allgather: this rank's construction does not hold rows si_row_shard(n_total, rank, world)");
  else if (hipMalloc(reinterpret_cast<void**>(&send), (size_t)chunk * mx * sizeof(double)) != hipSuccess ||
           hipMalloc(reinterpret_cast<void**>(&recv), (size_t)W * chunk * mx * sizeof(double)) != hipSuccess ||
           hipMalloc(reinterpret_cast<void**>(&w_full), (size_t)ld * sizeof(double)) != hipSuccess ||
           hipMalloc(reinterpret_cast<void**>(&p_full), (size_t)ld * M * sizeof(double)) != hipSuccess)
    lrc = fail(ctx, SI_ERR_NOMEM, "si_construct_allgather: device allocation failed");
  if ((rc = comm_agree(ctx, lrc, "si_construct_allgather")) != SI_OK) {
    cleanup();
    if (w_full) (void)hipFree(w_full);
    if (p_full) (void)hipFree(p_full);
    return rc;
  }
  hipStream_t st = ctx->stream;
  hipError_t e = hipMemsetAsync(w_full, 0, (size_t)ld * sizeof(double), st);
  if (e == hipSuccess) e = hipMemsetAsync(p_full, 0, (size_t)ld * M * sizeof(double), st);
  if (e == hipSuccess) e = hipMemsetAsync(send, 0, (size_t)chunk * mx * sizeof(double), st);
  ncclResult_t nr = ncclSuccess;
  for (int64_t c0 = 0; c0 < ncols && e == hipSuccess && nr == ncclSuccess; c0 += chunk) {
    const int64_t nc = std::min(chunk, ncols - c0);
    for (int64_t j = 0; j < nc && e == hipSuccess; ++j) {  // pack this rank's rows of columns [c0, c0 + nc)
      const int64_t col = c0 + j;
      const double* src = col < M ? ctx->d_P + col * ld_loc : ctx->d_swa;
      if (n_loc > 0) e = hipMemcpyAsync(send + j * mx, src, (size_t)n_loc * sizeof(double), hipMemcpyDeviceToDevice, st);
    }
    if (e != hipSuccess) break;
    nr = rccl().AllGather(send, recv, (size_t)(chunk * mx), ncclFloat64, comm_of(ctx), st);
    for (int r = 0; r < W && e == hipSuccess && nr == ncclSuccess; ++r) {
      const int64_t nr_rows = r1[(size_t)r] - r0[(size_t)r];
      if (nr_rows <= 0) continue;
      const double* blk = recv + (size_t)r * chunk * mx;
      const int64_t np = std::min(nc, std::max<int64_t>(0, (int64_t)M - c0));  // columns of P in this chunk
      if (np > 0)
        e = hipMemcpy2DAsync(p_full + c0 * ld + r0[(size_t)r], (size_t)ld * sizeof(double), blk, (size_t)mx * sizeof(double),
                             (size_t)nr_rows * sizeof(double), (size_t)np, hipMemcpyDeviceToDevice, st);
      if (e == hipSuccess && c0 + nc == ncols)  // the last column of the last chunk is W_swa
        e = hipMemcpyAsync(w_full + r0[(size_t)r], blk + (nc - 1) * mx, (size_t)nr_rows * sizeof(double), hipMemcpyDeviceToDevice, st);
    }
  }
  const hipError_t e2 = hipStreamSynchronize(st);
  cleanup();
  if (nr != ncclSuccess || e != hipSuccess || e2 != hipSuccess) {
    (void)hipFree(w_full);
    (void)hipFree(p_full);
    if (nr != ncclSuccess) return fail(ctx, SI_ERR_COMM, std::string("si_construct_allgather: ncclAllGather: ") + rccl().GetErrorString(nr));
    return fail(ctx, SI_ERR_HIP, std::string("si_construct_allgather: ") + hipGetErrorString(e != hipSuccess ? e : e2));
  }
  const std::vector<double> sv = ctx->svals;
  construct_install(ctx, n_total, M, w_full, p_full);  // frees the row-local construction; takes ownership of both
  ctx->svals = sv;
  return SI_OK;
}

// the whole data-sharded chain in one call: per transition  eval (this rank's observations) -> all-reduce of the
// nchains partial sums -> accept, all on the ctx's stream, no host round trip per step
int32_t si_sample_rwmh_sharded(si_ctx* ctx, int64_t itr, double sigma_z, uint64_t seed, int32_t chain_id0, int32_t nchains,
                               int64_t d_total, double* Z_out, double* lp_out, double* accept_rate_out) {
  if (!ctx) return SI_ERR_INVALID;
  int32_t rc = need_comm(ctx, "si_sample_rwmh_sharded");
  if (rc != SI_OK) return rc;
  // (set-up state, arguments and the output allocation are local: agree before the first per-transition all-reduce)
  if ((rc = comm_agree(ctx, si_rwmh_begin(ctx, itr, sigma_z, seed, chain_id0, nchains, d_total), "si_sample_rwmh_sharded")) != SI_OK) {
    if (ctx->sw_Z) {
      const std::string msg = ctx->err;
      (void)si_rwmh_abort(ctx);
      ctx->err = msg;
    }
    return rc;
  }
  for (int64_t t = 0; t < itr; ++t) {
    if ((rc = si_rwmh_step_eval(ctx, nullptr)) != SI_OK || (rc = si_rwmh_allreduce_sse(ctx)) != SI_OK ||
        (rc = si_rwmh_step_accept(ctx, nullptr)) != SI_OK) {
      const std::string msg = ctx->err;
      (void)si_rwmh_abort(ctx);
      ctx->err = msg;
      return rc;
    }
  }
  return si_rwmh_end(ctx, Z_out, lp_out, accept_rate_out);
}

}  // extern "C"
