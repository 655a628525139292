// Internal declarations shared by the C-ABI translation unit and the kernel launchers.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include <string>
#include <vector>

#include "../../include/subspace_hip.h"

namespace si {

// leading dimensions of device matrices are padded to a multiple of this many elements (512 B) so that every
// column starts 512-B aligned, 16-B vector accesses are legal whatever N is (N is odd at cfg2), and the 64-row slabs
// of the Gram kernel never run past a column (the padding rows of A are kept at zero).
constexpr int64_t LD_ALIGN = 64;
inline int64_t pad_ld(int64_t n) { return (n + LD_ALIGN - 1) / LD_ALIGN * LD_ALIGN; }

// one 16-feature tile of a layer as a wave runs it (64 bytes: one scalar load); chain_fused_program builds the lists
enum { SI_CG_SYNC = 1, SI_CG_GLOBAL = 2, SI_CG_MARKER = 4 };
struct CgTileD {
  int64_t woff, boff;      // elements into the weight vector: the layer's W and bias
  int32_t out, row0;       // rows of W (its pitch); first feature of the tile
  int32_t in, pad2;        // columns of W
  int32_t flags;           // SI_CG_SYNC: the wave's last tile of a layer; _GLOBAL: outputs go to yhat; _MARKER: no tile, barrier only
  int32_t img_in, ldi;     // input image (0 X tile, 1 / 2 the two activation images) and its pitch
  int32_t img_out, ldo;    // output image and its pitch
  int32_t act, pad0, pad1;  // act: SI_ACT_* | layer index << 8
};
struct EventPair {
  hipEvent_t a, b;
  int cls;
};

// geometry of one convolution-shaped gather (kernels_conv.hip): the tensor T holds Cp channels (channel-fastest) on a
// Wi x Hi grid per image; GEMM position (wo, ho) and tap (a, c) address ((wo, ho)*snum - pad + (a, c)*dil) / sden
struct ConvGeom {
  int Cp, Wi, Hi, Wo, Ho, KW, KH;
  int snum_w, snum_h, sden_w, sden_h, pad_w, pad_h, dil_w, dil_h;
  int Kvalid;          // Cp*KW*KH: taps at or past it read as zero (k is padded to whole 16-deep tiles)
  int64_t img_stride;  // Cp*Wi*Hi
};

// one layer of a Chain as the device executes it (capi_net.hip builds it from the caller's si_layer table)
struct LayerPlan {
  int kind = 0, act = 0;
  int64_t w_off = 0, b_off = 0;
  int in_feat = 0, out_feat = 0;         // the reference's feature counts (si_layer in / out)
  int64_t in_elems = 0, out_elems = 0;   // per observation in the layout the device uses (channel pitch inside conv stacks)
  int C = 0, Cp = 0, Co = 0, Cop = 0;    // input / output channels and their even pitches
  int Wi = 0, Hi = 0, Wo = 0, Ho = 0, KW = 0, KH = 0, sw = 1, sh = 1;
  int Kp = 0, KpT = 0;                   // padded tap counts of the forward / data-gradient GEMMs
  size_t wp_off = 0, bp_off = 0;         // packed weights / bias inside the pack workspace (doubles)
  ConvGeom g{}, gT{};
};
struct NetPlan {
  std::vector<LayerPlan> L;
  bool has_conv = false;        // any non-Dense layer: the generic path of capi_net.hip runs the chain
  bool input_spatial = false;   // the chain starts on (W, H, C) data: X is re-laid channel-fastest once
  int in_W = 0, in_H = 0, in_C = 0, in_Cp = 0;
  int64_t in_elems = 0;         // per observation, device layout
  int64_t max_elems = 1;        // largest per-observation activation (device layout) over input and all layers
  size_t wpack_elems = 1;       // forward packs of all conv layers
  size_t wt_elems = 1;          // largest transposed pack (data gradient), one layer at a time
  int max_rows = 1;             // largest row count handed to the row-sum kernels
};
struct NetScratch {             // reverse-sweep workspaces (sized by net_scratch_sizes)
  double *bwpart = nullptr, *rspart = nullptr, *wt = nullptr, *dbtmp = nullptr;
  // gradient mode: pidx[l] = byte index tensor of conv layer l whose MaxPool ran fused with it (net_grad_fused), else nullptr
  uint8_t* const* pidx = nullptr;
};

// on-device training (capi_train.hip)
struct TrainState {
  std::vector<si_layer> layers;
  int64_t N = 0, Btot = 0, Bmax = 0;
  int32_t in_dim = 0, out_dim = 0;
  double *X = nullptr, *Y = nullptr, *Xb = nullptr, *Yb = nullptr;
  int64_t* idx = nullptr;
  // batch indices travel through two pinned buffers (copy in, async H2D, event): a step never synchronises the stream, so the
  // caller's work between two steps (its next batch, the K1 push) overlaps the GPU's
  int64_t* idx_pin[2] = {nullptr, nullptr};
  hipEvent_t idx_ev[2] = {nullptr, nullptr};
  bool idx_busy[2] = {false, false};
  int idx_slot = 0;
  float *w32 = nullptr, *m32 = nullptr, *v32 = nullptr;
  double *w64 = nullptr, *gw = nullptr;
  std::vector<double*> hs;
  double* delta[2] = {nullptr, nullptr};
  double *bwpart = nullptr, *rspart = nullptr, *ssepart = nullptr, *sse = nullptr;
  int sse_blocks = 0;
  int opt = 0;
  double eta = 0.0, p1 = 0.0, p2 = 0.0, bp1 = 0.0, bp2 = 0.0;
  bool grad_ready = false;  // gw holds a gradient that has not been applied yet
  bool fuse_tail = false;   // narrow head: folded into the epilogue of the layer in front of it (forward and reverse)
  int fuse_slots = 0;
  double* part = nullptr;   // [fuse_slots][out_last][Bmax] partial head products
  // compute_dtype = SI_F32 (Dense chains; reference src/subspace_construction.jl:39-43 with a Float32 model AND Float32 data):
  // fp32 data, activations, deltas and gradient; fp64 only for the loss, the head's partial sums and the sums over the batch
  bool f32 = false;
  float *X32 = nullptr, *Xb32 = nullptr;
  struct SweepF32Ws* ws32 = nullptr;   // kept outputs, deltas, gradient, scratch of the fp32 sweep
  // chains with Conv / MaxPool / flatten layers (generic path, capi_net.hip)
  NetPlan plan;
  double *Xc = nullptr, *wpack = nullptr;
  NetScratch scratch;
  std::vector<uint8_t*> pidx;   // per layer: window-index bytes of a Conv layer fused with its MaxPool (net_grad_fused)
};

struct Ctx {
  int device = -1;
  TrainState* train = nullptr;
  hipStream_t own_stream = nullptr;
  hipStream_t stream = nullptr;
  std::string err;
  char devname[256] = {0};
  int num_cu = 256;
  // R1: RCCL communicator of this ctx (comm.hip; an ncclComm_t, kept opaque here), one rank per ctx
  void* comm = nullptr;
  int comm_world = 0, comm_rank = 0;
  double* d_commtmp = nullptr;  // device scratch of the host-value collectives
  double* d_gathertmp = nullptr;  // device scratch of si_comm_allgather_host, kept between calls
  size_t gathertmp_cap = 0;

  // development build only (-DSI_DEV_KNOBS, SI_OVERLAP_HALVES=1): the two halves of the batch of ONE chain on two streams
  bool overlap_halves = false;
  hipStream_t stream2 = nullptr;
  hipEvent_t ev_fork = nullptr, ev_join = nullptr;
  // profiling
  // pinned host staging for the small construct-time transfers (G down, V up): pageable copies cost tens of us each
  double* h_pin = nullptr;
  size_t h_pin_cap = 0;
  char* h_outpin = nullptr;      // pinned staging of the samples / lp of a device-resident loop on their way to the caller's arrays
  size_t h_outpin_cap = 0;
  double* h_stage[2] = {nullptr, nullptr};   // pinned staging of si_reconstruct's output pipeline (kept between calls)
  size_t h_stage_cap = 0;
  double *d_stage[2] = {nullptr, nullptr}, *d_zstage[2] = {nullptr, nullptr};   // its device-side double buffer
  size_t d_stage_cap = 0, d_zstage_cap = 0;
  // streamed output map (si_sample_rwmh_weights): device ring of the chains' CURRENT weights, pinned twin, events
  double* d_wring = nullptr;
  int32_t* d_accflag = nullptr;
  double* h_wring[4] = {nullptr, nullptr, nullptr, nullptr};
  hipEvent_t ev_wcomp[4] = {nullptr, nullptr, nullptr, nullptr}, ev_wcopy[4] = {nullptr, nullptr, nullptr, nullptr};
  int64_t wring_N = 0;
  int32_t wring_C = 0;
  bool profiling = false;
  uint32_t prof_mask = 0xffffffffu;  // classes that get event pairs while profiling is on
  std::vector<EventPair> pending;
  std::vector<hipEvent_t> event_pool;
  si_stats stats{};

  // ---- construction state (reference src/subspace_construction.jl:31-33,45-52,61-65)
  bool c_active = false, c_finished = false, gram_valid = false;
  int64_t N = 0, ldA = 0, Kcap = 0, K = 0;  // K = columns held
  int32_t max_cols = 0;
  int64_t npush = 0;
  double* d_swa = nullptr;    // N (padded) fp64
  double* d_A = nullptr;      // ldA x Kcap, column-major (fp64; FLOATS behind the same pointer when a_dtype == SI_F32); with max_cols>0 a ring of max_cols columns
  int32_t a_dtype = SI_F64;   // storage of the deviation matrix (si_construct_set_storage)
  size_t a_bytes = 0;         // bytes allocated behind d_A
  int32_t a_zero_dtype = -1;  // element type for which the padding rows of d_A are known to be zero (-1: unknown)
  int64_t a_zero_cols = 0;    // ... in the first a_zero_cols columns
  bool a_pending = false;     // si_construct_begin ran, the matrix is allocated / checked by the first push or si_construct_set_storage
  void* d_wstage = nullptr;   // staging of si_construct_set_mean
  // pipelined host pushes (si_construct_push): two pinned host buffers -> two device buffers, events mark the H2D of each
  void* h_wpin[2] = {nullptr, nullptr};
  void* d_wpush[2] = {nullptr, nullptr};
  hipEvent_t ev_wpin[2] = {nullptr, nullptr};   // H2D of buffer b done (copy stream)
  hipEvent_t ev_wk1[2] = {nullptr, nullptr};    // K1 on buffer b done (compute stream)
  bool wpin_busy[2] = {false, false};
  size_t wpin_bytes = 0;
  uint64_t wpin_next = 0;
  double* d_nvals = nullptr;  // per-push n values of a batched push
  int64_t nvals_cap = 0;
  size_t wstage_bytes = 0;
  double* d_G = nullptr;      // K x K
  int64_t g_cap = 0, v_cap = 0, a_cols_alloc = 0;
  double* d_Gpart = nullptr;  // partial slabs
  size_t gpart_bytes = 0;
  double* d_V = nullptr;      // K x Mpad (row k contiguous)
  double* d_P = nullptr;      // ldA x M
  int32_t M_built = 0;
  std::vector<double> svals;
  // ill-conditioned route (second-stage Gram): B = A * V_full, d_G then holds G2 = B'B
  int refine_stage = 0;           // 0: d_G = A'A;  1: d_B valid and d_G = B'B (possibly all-reduced by the caller)
  double* d_At = nullptr;         // K > N route: A' (pad_ld(K) x N), kept between finishes of one shape
  int64_t at_cap = 0;
  double* d_B = nullptr;          // ldA x K
  std::vector<double> vfull;      // K x K eigenvectors of A'A, columns DESCENDING by eigenvalue (host)

  // ---- inference state (reference src/space_inference.jl:88-95,111-116,125)
  bool i_ready = false;
  bool i_owns_swaP = false;
  std::vector<si_layer> layers;
  int64_t iN = 0, ldP = 0;
  int32_t iM = 0;
  const double* i_swa = nullptr;  // device
  const double* i_P = nullptr;    // device, ldP x M
  double* d_iswa = nullptr;       // owned copies when set from host
  double* d_iP = nullptr;
  double* d_X = nullptr;
  double* d_Y = nullptr;
  int32_t in_dim = 0, out_dim = 0;
  int64_t B = 0;
  double sigma_m = 1.0;
  double sigma_p = 0.0;          // > 0: the prior term the reference leaves dead is added (si_infer_set_prior; non-default)
  double* d_wsq = nullptr;       // ||W_swa + P z_c||^2 per chain (chains_cap), only with the prior on
  double* d_wsqpart = nullptr;   // its block partials (fw_slots x wsq_blocks)
  int wsq_blocks = 0;
  // forward workspace: `fw_slots` chain slots (1 after si_infer_setup; grown by ensure_batch for multi-chain calls)
  int fw_slots = 0;
  double* d_w = nullptr;       // reconstructed weights, fw_slots x pad_ld(N)
  double* d_act[2] = {nullptr, nullptr};  // ping-pong activations, fw_slots x act_elems (maxwidth x B padded)
  int64_t act_elems = 0, max_stored = 0;  // act_elems = pad_ld(max_stored * B): slot stride of d_act
  double* d_ssepart = nullptr;  // per-block SSE partials, fw_slots x sse_blocks
  bool fuse_tail = false;       // last layer folded into the epilogue of the layer in front of it
  int fuse_slots = 0;
  double* d_part = nullptr;     // fw_slots x [slots][out_last][B] partial last-layer products
  double* d_yhat = nullptr;     // fw_slots x out_dim x B, only filled on request (si_forward)
  // compute_dtype = SI_F32 (Dense chains): X rounded once, weights rounded once per evaluation (K4 writes both), fp32 activations
  bool f32 = false;
  int fuse_slots32 = 0;                      // feature slots of the fp32 fused head (fuse_slots stays the fp64 path's count)
  float* d_X32 = nullptr;
  float* d_w32 = nullptr;                    // fw_slots x pad_ld(N)
  float* d_act32[2] = {nullptr, nullptr};    // fw_slots x act_elems
  int sse_blocks = 0;
  int main_layer = 0;
  // gradient workspace (allocated on the first si_logdensity_grad)
  bool g_ready = false;
  std::vector<double*> d_hs;   // post-activation output of every layer, out_l x B
  double* d_delta[2] = {nullptr, nullptr};
  double* d_gw = nullptr;      // gradient w.r.t. the flat weight vector
  double* d_bwpart = nullptr;  // split-K partials of dW
  double* d_rspart = nullptr;  // row-sum partials
  double* d_ptgpart = nullptr;
  double* d_gz = nullptr;
  struct SweepF32Ws* g_ws32 = nullptr;   // compute_dtype = SI_F32: the fp32 sweep's workspace (si_logdensity_grad)
  // chains with Conv / MaxPool / flatten layers (generic path, capi_net.hip)
  NetPlan plan;
  double* d_Xc = nullptr;      // X re-laid channel-fastest (input_spatial)
  double* d_wpack = nullptr;   // packed conv weights of the current evaluation
  float* d_wpack32 = nullptr;  // the same in fp32 (compute_dtype = SI_F32 on a Conv chain)
  NetScratch g_scratch;
  std::vector<uint8_t*> d_pidx;  // per layer: window-index bytes of a Conv layer fused with its MaxPool (gradient workspace)
  // sampler state (device)
  double* d_zcur = nullptr;   // M x C
  double* d_zprop = nullptr;  // M x C
  double* d_lpcur = nullptr;  // C
  double* d_sse = nullptr;    // C
  int64_t* d_nacc = nullptr;  // C
  uint64_t* d_steps = nullptr;  // C: transition counter of every chain (device-resident)
  int32_t chains_cap = 0;
  // step-wise sampler session (si_rwmh_begin .. si_rwmh_end)
  double* sw_Z = nullptr;
  double* sw_lp = nullptr;
  int64_t sw_itr = 0, sw_next = 0;
  double sw_sigma_z = 0.0, sw_d = 0.0;
  uint64_t sw_seed = 0;
  int32_t sw_chain0 = 0, sw_C = 0;
  bool sw_evaluated = false;
  double *d_outZ = nullptr, *d_outlp = nullptr;   // device-side sample / lp arrays of si_sample_rwmh*, kept between calls
  size_t outZ_cap = 0, outlp_cap = 0;             // (a hipMalloc / hipFree pair per call was 0.4 ms of a 20-transition call)
  bool defer_sse_final = false;   // eval_density leaves the SSE block partials in d_ssepart (sample_rwmh_impl's fused tail sums them)
  bool chain_loop_enabled = true;   // si_set_chain_loop: 0 forces the launch-per-step loop (the parity tests compare the two)
  // narrow Dense chains (kernels_chain_grid.hip): every layer of the density in ONE launch, and K6 as a persistent grid loop
  int chain_mode = 1;               // si_set_chain_loop: 0 per-layer launches, 1 automatic, 2 the fused forward without the persistent loops
  bool fused_ok = false;            // the chain set up is of that class (infer_setup_common)
  unsigned* d_gridsync = nullptr;   // grid loop: one 128-byte counter line per chain + the status line
  int gridsync_chains = 0;
  CgTileD* d_cgprog = nullptr;      // the chain's tile program (chain_fused_program)
  int cg_start[5] = {0}, cg_count[5] = {0}, cg_chunks[5] = {0};
  // the same two kernels specialised to the chain's shapes at run time (chain_spec_rtc.cpp); buffers of the specialised loop
  bool chain_spec = true;           // si_set_chain_loop 3 / 4: generic kernels only
  double* d_specw = nullptr;        // 4 weight vectors in fragment order per chain ([parity][accepted, rejected])
  double* d_specy = nullptr;        // 2 output vectors per chain ([parity])
  int* d_specperm = nullptr;        // natural weight index -> fragment order
  const void* specperm_for = nullptr;   // (the SpecKernels the permutation was generated by)
  int spec_chains = 0, spec_fo = 0;
  int last_density_spec = 0, last_loop_spec = 0;   // si_chain_kernel_info: did the last density / sampling call run specialised kernels
  std::string spec_message;         // why not, when they did not (hiprtc log, class limits)
};

void free_train(Ctx* c);
void comm_release(Ctx* c);  // comm.hip: destroy the communicator (si_destroy / si_comm_destroy)
// capi.hip: make `c` hold a FINISHED construction of N rows and M columns without a deviation matrix (what a rank that
// receives (W_swa, P) from another rank ends up with).  adopt allocates zeroed W_swa / P; install takes ownership of the
// two device buffers (ld = pad_ld(N), padding rows zero).  Both drop whatever construction / bound inference was there.
int32_t construct_adopt(Ctx* c, int64_t N, int32_t M);
void construct_install(Ctx* c, int64_t N, int32_t M, double* w_swa, double* P);

// error helpers -------------------------------------------------------------------------------
int32_t fail(Ctx* c, int32_t code, const std::string& msg);
#define SI_HIP(c, expr)                                                                    \
  do {                                                                                     \
    hipError_t e_ = (expr);                                                                \
    if (e_ != hipSuccess)                                                                  \
      return si::fail((c), SI_ERR_HIP, std::string(#expr) + ": " + hipGetErrorString(e_)); \
  } while (0)

// Dynamic-LDS opt-in of a kernel (hipFuncAttributeMaxDynamicSharedMemorySize) is a per-DEVICE property of the loaded code
// object: a process may hold contexts on several GPUs, so the "already set" note is kept per device.
struct LdsOptIn {
  size_t done[64] = {0};
  void ensure(const void* kern, size_t lds) {
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) dev = -1;
    if (dev >= 0 && done[dev] >= lds) return;
    (void)hipFuncSetAttribute(kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    if (dev >= 0) done[dev] = lds;
  }
};

// profiling scope: records an event pair around kernel launches of one class
struct ProfScope {
  Ctx* c;
  int cls;
  hipEvent_t a = nullptr, b = nullptr;
  ProfScope(Ctx* c, int cls, double flops, double bytes);
  ~ProfScope();
};

// ---- kernel launchers (kernels_*.hip) -------------------------------------------------------
// K1: s <- (n*s + w)/(n+1); acol <- w - s   (three rounded ops, no FMA; reference :46-47,51)
// (a_dtype: element type of the deviation matrix -- SI_F64, the reference's, or the opt-in SI_F32 storage; ldA in ELEMENTS)
void launch_swa_dev_push(hipStream_t st, const void* w, int32_t w_dtype, double* s, void* acol,
                         int64_t N, double n, int num_cu, int32_t a_dtype = SI_F64);
void launch_swa_dev_push_batch(hipStream_t st, const void* w, int32_t w_dtype, int64_t ld, double* s, void* A,
                               int64_t ldA, int64_t N, int count, const double* nvals_dev, int64_t slot0, int64_t kcap,
                               int num_cu, int32_t a_dtype = SI_F64);
// K2: G = A'A over columns [0,K) of A (ldA), rows [0,N); result K x K col-major symmetric in G
// returns bytes of partial workspace required (if Gpart == nullptr nothing is launched)
size_t launch_gram(hipStream_t st, const void* A, int64_t ldA, int64_t N, int64_t K, double* Gpart,
                   double* G, int num_cu, Ctx* prof, int32_t a_dtype = SI_F64);
// K3: P[:, m] = sum_k A[:, k] * V[k*Mpad + m], m < M;  Mpad = project_mpad(M) (V rows zero-padded)
int project_mpad(int M);
void launch_project(hipStream_t st, const void* A, int64_t ldA, int64_t N, int64_t K, const double* V,
                    int32_t M, int32_t Mpad, double* P, int64_t ldP, int num_cu, int32_t a_dtype = SI_F64);
// the same product on the matrix cores (kernels_bwd.hip gemm_f64_kernel); launch_project uses it for M > 32
void launch_project_mfma(hipStream_t st, const double* A, int64_t ldA, int64_t N, int64_t K, const double* V, int32_t M,
                         int32_t Mpad, double* P, int64_t ldP);
// K3 for wide subspaces as a slab stream (kernels_project.hip; K <= 128, else returns false)
bool launch_project_stream(hipStream_t st, const double* A, int64_t ldA, int64_t N, int64_t K, const double* V, int32_t M,
                           int32_t Mpad, double* P, int64_t ldP, int num_cu);
bool launch_project_stream_f32(hipStream_t st, const float* A, int64_t ldA, int64_t N, int64_t K, const double* V, int32_t M,
                               int32_t Mpad, double* P, int64_t ldP, int num_cu);   // the same for an fp32-stored A
// K4: w[c*ldw + r] = swa[r] + sum_m P[r + m*ldP] * Z[m + c*M]
void launch_reconstruct(hipStream_t st, const double* swa, const double* P, int64_t ldP, int64_t N,
                        int32_t M, const double* Z, int32_t C, double* w, int64_t ldw, int num_cu,
                        float* w32 = nullptr /* SI_F32: the same sums rounded once to fp32 */, int64_t ldw32 = 0);
// Chain batching of the forward pass: `n` chain slots run in ONE launch (grid.y = slot).  Strides, in elements, from
// one slot to the next: `w` of the reconstructed weight vectors (W, bias, Wlast all live in it), `hin` / `hout` of
// the layer's input / output activations (hin = 0 for the first layer: X is shared), `part` of the fused-tail partials.
struct ChainBatch {
  int n = 1;
  int64_t w = 0, hin = 0, hout = 0, part = 0;
  int64_t part_ld = 0;  // column pitch of the fused-tail partials when a launch covers only a column range of B (0 = B)
};
// K5: Hout[i + out*b] = act(sum_k W[i + out*k] * Hin[k + in*b] + bias[i])
void launch_dense_f64(hipStream_t st, const double* W, const double* bias, const double* Hin,
                      double* Hout, int32_t out, int32_t in, int64_t B, int32_t act, const ChainBatch& cb = ChainBatch());
// K5 fused tail: the layer in front of a narrow (out_last <= SI_FUSE_MAX_OUT) last layer does not store its output;
// it writes per-slot partial products with the last layer's weights, summed by launch_tail_sse (block partials of
// (y - yhat)^2 to `blockpart`, to be finished by sse_final via launch_sse_final)
constexpr int SI_FUSE_MAX_OUT = 4;
int dense_fused_slots(int32_t out);
void launch_dense_f64_fused(hipStream_t st, const double* W, const double* bias, const double* Hin, int32_t out,
                            int32_t in, int64_t B, int32_t act, const double* Wlast, int32_t out_last, double* part,
                            const ChainBatch& cb = ChainBatch(), double* Hkeep = nullptr /* also store the layer's output */);
// K5 for a wide layer with a short reduction (kernels_gemm_panel.hip: in <= 128, persistent workgroups that keep their X operands in
// registers, stream W through an LDS ring and let every store drain under the MFMAs of the next tiles); same bits as the big-tile
// kernel.  launch_dense_f64 routes single-chain layers to it by shape (dense_panel_applies).
bool dense_panel_applies(const double* W, int32_t out, int32_t in, int64_t B, int32_t act);
bool launch_dense_f64_panel(hipStream_t st, const double* W, const double* bias, const double* Hin, double* Hout, int32_t out,
                            int32_t in, int64_t B, int32_t act, int grid = 0);
// K5 for small layers (kernels_gemm_small.hip): one wave per (feature slot, 16 observations) tile, operands straight from global
// memory; same bits as the big-tile kernel.  launch_dense_f64 / _fused route to it by shape (dense_small_applies).
bool dense_small_applies(int32_t out, int32_t in, int64_t B, int nchains, int32_t bm);
void launch_dense_small_f64(hipStream_t st, const double* W, const double* bias, const double* Hin, double* Hout, int32_t out,
                            int32_t in, int64_t B, int32_t act, int32_t slot_feats, const ChainBatch& cb);
void launch_dense_small_f64_fused(hipStream_t st, const double* W, const double* bias, const double* Hin, int32_t out, int32_t in,
                                  int64_t B, int32_t act, int32_t slot_feats, int32_t nslots, const double* Wlast, int32_t out_last,
                                  double* part, const ChainBatch& cb, double* Hkeep);
// K5 in fp32 (kernels_gemm_f32.hip; compute_dtype = SI_F32): same operation and chain batching, fp32 operands / outputs on
// v_mfma_f32_32x32x2_f32; the fused head writes fp64 partials that launch_tail_sse sums as in the fp64 path
void launch_dense_f32(hipStream_t st, const float* W, const float* bias, const float* Hin, float* Hout, int32_t out, int32_t in,
                      int64_t B, int32_t act, const ChainBatch& cb = ChainBatch());
int dense_f32_fused_slots(int32_t out, int32_t in, bool aligned);
bool launch_dense_f32_dx(hipStream_t st, const float* Wt, const float* zero_bias, const float* Delta, float* Dout, int32_t out,
                         int32_t in, int64_t B, const float* Hprev, int32_t act_prev);
void launch_dense_f32_fused(hipStream_t st, const float* W, const float* bias, const float* Hin, int32_t out, int32_t in, int64_t B,
                            int32_t act, const float* Wlast, int32_t out_last, double* part, const ChainBatch& cb = ChainBatch(),
                            float* Hkeep = nullptr);
void launch_narrow_f32(hipStream_t st, const double* src, float* dst, int64_t n);
void launch_sse_f32(hipStream_t st, const float* yhat, const double* y, int64_t d, double* part, int nblocks, double* sse_out,
                    int nch = 1, int64_t yhat_stride = 0, double* yhat64 = nullptr, int64_t yhat64_stride = 0, bool with_final = true);
void launch_tail_sse(hipStream_t st, const double* part, int slots, int out_last, int64_t B, const double* bias_last,
                     int act_last, const double* Y, double* yhat, double* blockpart, int nblocks,
                     const ChainBatch& cb = ChainBatch());
// sse_out[ch] = sum of blockpart[ch*nblocks .. +nblocks) for ch < nch
void launch_sse_final(hipStream_t st, const double* blockpart, int nblocks, double* sse_out, int nch = 1);
// K5: sse = sum (y - yhat)^2 over d elements; deterministic two-stage
int sse_num_blocks(int64_t d, int num_cu);
// (nch chain slots: yhat advances by yhat_stride per slot, y is shared, part holds nch*nblocks partials)
// (with_final = false: the block partials stay in `part`; the fused sampler loop sums them inside its accept kernel)
void launch_sse(hipStream_t st, const double* yhat, const double* y, int64_t d, double* part,
                int nblocks, double* sse_out, int nch = 1, int64_t yhat_stride = 0, bool with_final = true);
// backward pass (kernels_bwd.hip): gradient of the log-density w.r.t. the flat weights and its pull-back P' g
void launch_backward_data(hipStream_t st, const double* W, const double* Delta, const double* Hprev, double* DeltaPrev,
                          int32_t out, int32_t in, int64_t B, int32_t act_prev);
// Reverse sweep through a Dense chain shared by si_logdensity_grad (capi_infer.hip) and the training step (capi_train.hip).
// On entry delta[0] holds Delta_L (launch_delta_out) and gw is zeroed, both on `st`; on return every dW / db of the
// chain is in gw.
struct DenseSweep {
  const si_layer* layers;
  size_t nl;
  bool fuse_tail;
  const double* w;          // flat fp64 weights
  const double* X;          // input of layer 0, in_0 x B
  double* const* hs;        // kept outputs of every layer
  double* delta[2];
  double* gw;
  double* rspart;
  double* bwpart;
  int64_t B;
};
int32_t dense_reverse_sweep(Ctx* ctx, hipStream_t st, const DenseSweep& s);
// The same in fp32 together with the forward pass (capi_train.hip; kernels_gemm_f32.hip + kernels_bwd_f32.hip): forward with every
// layer's output kept, the SSE (fp64), Delta_L = scale * (y - yhat) * act_L'(yhat), reverse sweep; on return gw32 holds
// d (scale/2 * -SSE) / dw rounded to fp32 and *sse the sum of squared errors.  Shared by the training step and si_logdensity_grad
// with compute_dtype = SI_F32.
struct SweepF32Ws {   // workspace of one sweep at up to Bmax observations (sweep_f32_alloc / _free)
  std::vector<float*> hs32;
  float* delta32[2] = {nullptr, nullptr};
  float *gw32 = nullptr, *wt32 = nullptr, *zero32 = nullptr, *part32 = nullptr;
  double *rspart64 = nullptr, *tailpart64 = nullptr, *yhat64 = nullptr;
};
bool sweep_f32_alloc(Ctx* ctx, SweepF32Ws& ws, const si_layer* layers, int L, bool fuse_tail, int64_t N, int32_t in_dim, int32_t out_dim,
                     int64_t Bmax);
void sweep_f32_free(SweepF32Ws& ws);
struct DenseSweepF32 {
  const si_layer* layers;
  size_t nl;
  bool fuse_tail;
  const float* w32;     // flat fp32 weights
  const double* w64;    // the same in fp64 (the head's bias is added in fp64)
  const float* X32;     // input of layer 0, in_0 x B
  const double* Y;
  SweepF32Ws* ws;
  double* part;         // head partials [slots][out_last][B]
  double *ssepart, *sse;
  int sse_blocks;
  int64_t B, N;
  double scale;         // of Delta_L: -2 / d for the mse loss, 1 / sigma^2 for the log-density
};
int32_t dense_value_and_grad_f32(Ctx* ctx, hipStream_t st, const DenseSweepF32& s);
void launch_widen_f32_to_f64(hipStream_t st, const float* src, int64_t n, double* dst, int num_cu);
// dW[out x in] = Delta * Hprev' into dW, through `part` (backward_weight_part_elems doubles): split-K GEMM + fixed-order reduction;
// db != nullptr: db[out] = rowsum(Delta) as well (inside the GEMM where the LDS-DMA kernel runs, else by launch_rowsum)
size_t backward_weight_part_elems(int32_t out, int32_t in, int64_t B, int num_cu);
void launch_backward_weight(hipStream_t st, const double* Delta, const double* Hprev, double* part, int32_t out,
                            int32_t in, int64_t B, int num_cu, double* dW, double* db = nullptr);
void launch_split_reduce(hipStream_t st, const double* part, int nsplit, int64_t elems, double* dst);
void launch_delta_out(hipStream_t st, const double* Y, const double* Yhat, int64_t d, double scale, int act, double* delta);
void launch_rowsum(hipStream_t st, const double* D, int32_t out, int64_t B, double* part, double* db);
// reverse sweep through a narrow last layer (out_last <= SI_FUSE_MAX_OUT) in one pass over its input H (F x B):
// DeltaPrev = (W' Delta) .* act_prev'(H), dW = Delta H', dbprev = rowsum(DeltaPrev); part: tail_bwd_part_elems doubles
size_t tail_bwd_part_elems(int32_t out_last, int32_t F);
void launch_tail_bwd(hipStream_t st, const double* W, const double* Delta, const double* H, int32_t out_last, int32_t F,
                     int64_t B, int32_t act_prev, double* DeltaPrev, double* part, double* dW, double* dbprev);
int rowsum_chunks();
// the reverse sweep in fp32 (kernels_bwd_f32.hip): compute_dtype = SI_F32 of the on-device training step
size_t backward_weight_f32_part_elems(int32_t out, int32_t in, int64_t B, int num_cu);
void launch_backward_weight_f32(hipStream_t st, const float* Delta, const float* Hprev, float* part, int32_t out, int32_t in, int64_t B,
                                int num_cu, float* dW);
void launch_transpose_f32(hipStream_t st, const float* W, int32_t out, int32_t in, float* Wt);
size_t rowsum_f32_part_elems(int max_rows);
void launch_mul_dact_rowsum_f32(hipStream_t st, const float* G, const float* H, int rows, int64_t B, int act, float* D, double* part,
                                float* db);   // D = G .* act'(H) (H == nullptr: G itself; D == nullptr: nothing stored), db = rowsum(D)
void launch_delta_out_f32(hipStream_t st, const double* Y, const double* Yhat64, const float* Yhat32, int64_t d, double scale, int act,
                          float* delta);
size_t tail_bwd_f32_part_elems(int32_t out_last, int32_t F);
void launch_tail_bwd_f32(hipStream_t st, const float* W, const float* Delta, const float* H, int32_t out_last, int32_t F, int64_t B,
                         int32_t act_prev, float* DeltaPrev, double* part, float* dW, float* dbprev);
void launch_ptg(hipStream_t st, const double* P, int64_t ldP, int64_t N, int M, const double* g, double* part, double* gz);
int ptg_blocks();
// ---- Conv / MaxPool / flatten (kernels_conv.hip; host side capi_net.hip) ---------------------------------------------
void launch_conv_pack(hipStream_t st, const double* w, const double* b, double* Wp, double* bp, int KW, int KH, int CIN, int COUT,
                      int CINp, int COUTp, int Kp);
void launch_conv_pack_t(hipStream_t st, const double* w, double* Wt, int KW, int KH, int CIN, int COUT, int CINp, int COUTp, int KpT);
void launch_conv_forward(hipStream_t st, const double* Wp, const double* bp, const double* In, double* Out, const ConvGeom& g,
                         int COUTp, int Kp, int64_t npos, int act);
void launch_conv_forward_pool2(hipStream_t st, const double* Wp, const double* bp, const double* In, double* Out, const ConvGeom& g,
                         int COUTp, int Kp, int64_t npos, int act);
void launch_conv_forward_pool2_idx(hipStream_t st, const double* Wp, const double* bp, const double* In, double* Out, uint8_t* Idx,
                                   const ConvGeom& g, int COUTp, int Kp, int64_t npos, int act);
void launch_pool2_bwd_idx(hipStream_t st, const double* G, const double* Hp, const uint8_t* Idx, double* D, int Cp, int W2, int H2,
                          int64_t B, int act, double* part, int nout, double* db);
void launch_conv_backward_data(hipStream_t st, const double* Wt, const double* Delta, double* dX, const ConvGeom& gT, int CINp,
                               int KpT, int64_t npos_in);
int conv_dw_max_splits(int COUTp, int Kp, int64_t npos, int num_cu);   // bound over every position count up to npos (scratch sizing)
int conv_dw_splits(int COUTp, int Kp, int64_t npos, int num_cu, int64_t* ksplit_out);
void launch_conv_backward_weight(hipStream_t st, const double* Delta, const double* In, double* part, const ConvGeom& g, int COUTp,
                                 int Kp, int64_t npos, int nsplit, int64_t ksplit);
void launch_conv_unpack_dw(hipStream_t st, const double* part, int nsplit, double* gw, int KW, int KH, int CIN, int COUT, int CINp,
                           int COUTp, int Kp);
void launch_whcn_to_cwhn(hipStream_t st, const double* X, double* Xc, int W, int H, int C, int Cp, int64_t B);
void launch_cwhn_to_whcn(hipStream_t st, const double* Xc, double* X, int W, int H, int C, int Cp, int64_t B);
void launch_maxpool(hipStream_t st, const double* In, double* Out, int Cp, int Wi, int Hi, int Wo, int Ho, int PW, int PH, int sw,
                    int sh, int64_t B);
void launch_maxpool_bwd(hipStream_t st, const double* In, const double* Out, const double* Gout, double* Gin, int Cp, int Wi, int Hi,
                        int Wo, int Ho, int PW, int PH, int sw, int sh, int64_t B);
void launch_mul_dact(hipStream_t st, const double* G, const double* H, int64_t n, int act, double* D);
size_t dact_rowsum_ws_elems(int max_rows);
void launch_act_inplace(hipStream_t st, double* H, int64_t n, int act);
// the same forward kernels on fp32 operands (kernels_conv.hip compiled a second time with -DSI_CONV_F32: v_mfma_f32_16x16x4_f32,
// the same stagers, LDS images and index maps) -- compute_dtype = SI_F32 on Conv chains
void launch_conv_pack(hipStream_t st, const float* w, const float* b, float* Wp, float* bp, int KW, int KH, int CIN, int COUT, int CINp,
                      int COUTp, int Kp);
void launch_conv_forward(hipStream_t st, const float* Wp, const float* bp, const float* In, float* Out, const ConvGeom& g, int COUTp,
                         int Kp, int64_t npos, int act);
void launch_conv_forward_pool2(hipStream_t st, const float* Wp, const float* bp, const float* In, float* Out, const ConvGeom& g,
                               int COUTp, int Kp, int64_t npos, int act);
void launch_whcn_to_cwhn(hipStream_t st, const float* X, float* Xc, int W, int H, int C, int Cp, int64_t B);
void launch_cwhn_to_whcn(hipStream_t st, const float* Xc, float* X, int W, int H, int C, int Cp, int64_t B);
void launch_maxpool(hipStream_t st, const float* In, float* Out, int Cp, int Wi, int Hi, int Wo, int Ho, int PW, int PH, int sw, int sh,
                    int64_t B);
void launch_act_inplace(hipStream_t st, float* H, int64_t n, int act);
void launch_dense_narrow(hipStream_t st, const float* W, const float* bias, const float* Hin, float* Hout, int out, int in, int64_t B,
                         int act);
bool dense_narrow_applies(int out, int in, int64_t B, int num_cu);
void launch_dense_narrow(hipStream_t st, const double* W, const double* bias, const double* Hin, double* Hout, int out, int in, int64_t B,
                         int act);
void launch_maxpool_bwd_dact_rowsum(hipStream_t st, const double* In, const double* Out, const double* Gout, double* D, int Cp, int Wi,
                                    int Hi, int Wo, int Ho, int PW, int PH, int sw, int sh, int64_t B, int act, double* part, int nout,
                                    double* db);
void launch_mul_dact_rowsum(hipStream_t st, const double* G, const double* H, int rows, int64_t ncols, int act, double* D, double* part,
                            int nout, double* db);
// capi_net.hip: validation + geometry of a layer table (Dense chains included), and the generic forward / reverse sweep
int32_t net_plan(Ctx* c, const char* who, const si_layer* layers, int L, int64_t N, int in_dim, int out_dim, NetPlan& out);
// xin: input in device layout (net_input re-lays a (features x B) matrix when the chain starts on images); outs[l] = where
// layer l writes (out_elems * B doubles each); wpack: plan.wpack_elems doubles
void net_input(Ctx* c, const NetPlan& p, const double* X, double* Xc, int64_t B);
// `outs[l]` receives layer l's output.  pingpong = true (density path: nothing but the last output is needed): `outs` holds TWO
// buffers used alternately per EXECUTED layer, a Conv directly followed by MaxPool((2, 2)) on even sizes runs as one fused
// kernel, and *final_out is the buffer that holds the last layer's output.
// pidx != nullptr (gradient mode, outs[l] per layer): a Conv layer with net_grad_fused(p, l) runs fused with its MaxPool, writes
// outs[l + 1] and pidx[l] and leaves outs[l] untouched (it may be nullptr)
int32_t net_forward_f32(Ctx* c, const NetPlan& p, const float* w, const float* xin, int64_t B, float* const* outs, float* wpack,
                        float** final_out);
int32_t net_forward(Ctx* c, const NetPlan& p, const double* w, const double* xin, int64_t B, double* const* outs, double* wpack,
                    bool pingpong = false, double** final_out = nullptr, uint8_t* const* pidx = nullptr);
// Conv layer l directly followed by MaxPool((2, 2), stride 2) on even sizes, activation carried by the GEMM epilogue: in
// gradient mode the pair runs as one kernel that keeps a byte index instead of the un-pooled activation
bool net_grad_fused(const NetPlan& p, size_t l);
size_t net_pidx_bytes(const NetPlan& p, size_t l, int64_t B);
void net_scratch_sizes(const NetPlan& p, int64_t B, int num_cu, size_t* bwpart, size_t* rspart, size_t* wt, size_t* dbtmp);
// g0 holds d / d(output of the last layer) (out_feat x B) on entry; g0 / g1: max_elems * B doubles each; hs[l]: kept outputs
int32_t net_backward(Ctx* c, const NetPlan& p, const double* w, const double* xin, int64_t B, double* const* hs, double* g0,
                     double* g1, double* gw, const NetScratch& s);
// K6, device-resident loop (kernels_chain.hip): all `itr` transitions of `nchains` small Dense chains in ONE launch, one
// workgroup per chain, weights / data / activations in LDS; bit-identical to the launch-per-step loop of si_sample_rwmh
constexpr int SI_CHAIN_MAX_LAYERS = 8;
struct ChainLoopArgs {
  si_layer lay[SI_CHAIN_MAX_LAYERS];
  const double *swa, *P, *X, *Y;
  double *Z_out, *lp_out;
  int64_t* nacc_out;
  int64_t ldP, itr;
  uint64_t seed;
  double sigma_z, c0, sigma2;
  int N, M, B, L, chain_id0;
  int slot_feats, fuse_slots, nblocks;   // head slots of layer L-2 (BM / WM of dense_f64_kernel's choice), SSE virtual blocks
  int p_in_lds;
  int o_X, o_Y, o_act0, o_act1, o_part, o_blk, o_z, o_red, o_P, o_swa, o_map;
  int wp[SI_CHAIN_MAX_LAYERS], bp[SI_CHAIN_MAX_LAYERS];   // padded W / bias image of every layer (chain_loop_plan)
  long long* dbg_stamps;   // tools/chain_bench.hip (-DSI_CHAIN_STAMPS) only; nullptr in the library   // LDS offsets (doubles), set by chain_loop_plan
};
size_t chain_loop_plan(ChainLoopArgs& a, size_t lds_limit);   // LDS bytes, 0 = the model does not fit / does not apply
void launch_chain_loop(hipStream_t st, const ChainLoopArgs& a, int nchains, size_t lds);
// K5 fused over all layers + K6 as a persistent grid loop for narrow Dense chains (kernels_chain_grid.hip)
struct ChainFusedPlan {
  si_layer lay[SI_CHAIN_MAX_LAYERS];
  int L, B;
  int fuse_tail, slot_feats, fuse_slots;   // narrow head on the image of layer L-2 (slots of dense_f64_kernel<FUSE>)
  int ld[SI_CHAIN_MAX_LAYERS + 1];         // pitch of the input image of layer l; ld[l + 1] = of its output image
  int o_x, o_buf[2], o_part;               // LDS offsets (doubles)
  int lds_doubles;
  const CgTileD* prog;                     // device: the tile lists (0: a wave owns the batch tile; 1 .. 4: wave 0 .. 3 of a workgroup that shares it)
  int prog_start[5], prog_count[5], prog_chunks[5];
};
void chain_fused_program(const si_layer* layers, int L, bool fuse_tail, std::vector<CgTileD>& prog, int start[5], int count[5],
                         int chunks[5]);
size_t chain_fused_plan(ChainFusedPlan& p, const si_layer* layers, int L, int64_t B, int NB, bool fuse_tail, int slot_feats,
                        int fuse_slots);   // LDS bytes for batch tiles of 16 NB observations; 0 = not a chain of this class
// yhat[chain][o + out_last * b] for `nchains` weight vectors w + chain * w_stride (grid.y), every layer in one launch
// (wave_tiles: one WAVE per batch tile, four regions of `lds` bytes per workgroup, no workgroup barrier; else one workgroup per tile)
void launch_chain_fused(hipStream_t st, const ChainFusedPlan& p, int NB, bool wave_tiles, size_t lds, const double* w, int64_t w_stride,
                        const double* X, double* yhat, int64_t y_stride, int nchains);
struct ChainGridArgs {
  ChainFusedPlan p;
  const double *swa, *P, *X, *Y;
  double *wbuf, *ybuf;        // per chain: W_swa + P z' (w_stride apart) and the model outputs (y_stride apart); handed between workgroups
  int64_t w_stride, y_stride;
  unsigned* cnt;              // 32 words (one 128-byte line) per chain, zeroed before the launch
  unsigned* status;           // raised by a workgroup whose barrier wait timed out
  double *Z_out, *lp_out;
  int64_t* nacc_out;
  int64_t ldP, itr;
  uint64_t seed;
  double sigma_z, c0, sigma2;
  int N, M, G, chain_id0, nblocks;
  int y_in_lds, o_y, o_blk, o_z, o_red, o_flag;   // LDS offsets behind the images (chain_grid_plan)
};
size_t chain_grid_plan(ChainGridArgs& a, size_t lds_fused);
hipError_t launch_chain_grid(hipStream_t st, const ChainGridArgs& a, int NB, int nchains, size_t lds);
int dense_fused_slot_feats(int32_t out);   // features per head slot of the fused fp64 layer (kernels_gemm.hip: BM / WM)
// K6
void launch_rwmh_init(hipStream_t st, double* zcur, double* lpcur, int64_t* nacc, uint64_t* steps, int32_t M, int32_t C);
void launch_rwmh_propose(hipStream_t st, const double* zcur, double* zprop, int32_t M, int32_t C,
                         double sigma_z, uint64_t seed, int32_t chain_id0, const uint64_t* steps);
// last stage of the SSE reduction + accept + the next transition's proposal in one launch (kernels_stream.hip rwmh_tail_kernel)
void launch_rwmh_tail(hipStream_t st, const double* ssepart, int nparts, double* sse, double* zcur, double* zprop, double* lpcur,
                      int64_t* nacc, int32_t M, int32_t C, double c0, double sigma2, double sigma_z, uint64_t seed, int32_t chain_id0,
                      uint64_t* steps, double* Z_out, double* lp_out, int64_t itr, int32_t* accflag, bool propose_next);
void launch_rwmh_accept(hipStream_t st, double* zcur, const double* zprop, double* lpcur,
                        const double* sse, int64_t* nacc, int32_t M, int32_t C, double c0,
                        double sigma2, uint64_t seed, int32_t chain_id0, uint64_t* steps,
                        double* Z_out, double* lp_out, int64_t itr, const double* wsq = nullptr, double c0p = 0.0,
                        double sigma_p2 = 1.0, int32_t* accflag = nullptr /* per chain: 1 = this step's proposal was kept */);
// current weights of every chain after a transition: dst[c] = flag[c] ? wprop[c] : prev[c]   (prev == nullptr: all kept)
void launch_weights_select(hipStream_t st, const int32_t* flag, const double* wprop, int64_t ldw, const double* prev, double* dst,
                           int64_t ldd, int64_t N, int32_t C, int num_cu);
void launch_prior_grad(hipStream_t st, double* g, const double* w, int64_t n, double inv_s2, int num_cu);
void launch_widen_f32(hipStream_t st, const float* src, double* dst, int64_t n, int num_cu);
void launch_transpose(hipStream_t st, const void* A, int32_t a_dtype, int64_t lda, int64_t N, int64_t K, double* At, int64_t ldt);

// host copy pool (host_copy.cpp): parallel memcpy between pageable caller arrays and pinned staging
void host_copy(void* dst, const void* src, size_t bytes);
int host_copy_threads();
void host_copy_set_share(int nproc);   // `nproc` processes of the library share this host (one per GPU): shrink the pool's share
int host_cpu_budget();                 // affinity mask capped by the cgroup CPU quota
double parse_cpu_max(const char* text);
int host_copy_plan(int budget, int nproc, const char* env);

// host symmetric eigensolver (eig.cpp): a is n x n symmetric col-major, overwritten by eigenvectors
// (columns), w gets eigenvalues ascending.  Returns 0 on success.
int sym_eig(int n, double* a, double* w);
// M largest eigenpairs without the full eigenvector matrix (eig.cpp); g is left intact; w_top descending, V n x m.
// Verified (residual, orthogonality); returns non-zero when the caller should use sym_eig instead.
int sym_eig_top(int n, const double* g, int m, double* w_top, double* V);
// scaled-criterion two-sided Jacobi for a PSD matrix (eig_dispatch.cpp): a destroyed; w DESCENDING; v eigenvectors
int jacobi_eig_psd(int n, double* a, double* w, double* v);

}  // namespace si

struct si_ctx : public si::Ctx {};
