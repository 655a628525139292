// K5  Dense-chain forward in fp64 on the gfx950 matrix cores.
// Replaces reference src/space_inference.jl:92-94: `model_re(in_model, new_W)` + `new_model(in_data)`, i.e. per
// Flux Dense layer  H' = act.(W*H .+ b)  with W (out x in) taken IN PLACE from the flat weight vector (the static
// layer-offset table replaces the per-call Flux.destructure/re copies of src/libs.jl:55-57).
//
// All matrices are column-major (Julia):  W[i + out*k],  Hin[k + in*b],  Hout[i + out*b].
//
// Kernel: LDS-tiled GEMM on v_mfma_f64_16x16x4_f64.  One 16x16 MFMA tile has its ROWS on the batch index b and
// its COLUMNS on the feature index i, so that a wave stores 128-B contiguous runs of Hout.
//   A operand (16x4): lane l holds Hin[k = 4s + (l>>4)][b = l&15]     <- sH[b][k]   (row stride 18 doubles)
//   B operand (4x16): lane l holds   W[i = l&15][k = 4s + (l>>4)]     <- sW[k][i]   (row stride BM+16 doubles)
//   C/D: lane l, reg r holds D[b = (l>>4) + 4r][i = l&15]              (f64 map; NOT the f32 one)
// LDS row strides are chosen so that every ds_read_b64 of a 32-lane half hits 32 distinct bank pairs:
//   sW: (BM+16)*2 dwords == 32 (mod 64)  -> the two k rows of a half land in different halves of the bank row
//   sH: 18*2 = 36 dwords per b; 36*b mod 64 runs over all multiples of 4 for b = 0..15, +2 for the second k
// Staging is global -> registers -> LDS, double-buffered, one barrier per 16-deep k tile; two workgroups per CU
// overlap each other's barriers.  Roofline: MFMA f64 (78.6 TFLOP/s); algorithmic flops = 2*out*in*B per layer.
#include "si_internal.h"

namespace si {

typedef double d4 __attribute__((ext_vector_type(4)));

__device__ __forceinline__ double apply_act(double v, int act) {
  switch (act) {
    case SI_ACT_RELU: return v > 0.0 ? v : 0.0;   // Flux relu(x) = max(0, x)
    case SI_ACT_TANH: return tanh(v);
    case SI_ACT_SIGMOID: return 1.0 / (1.0 + exp(-v));
    default: return v;
  }
}

template <int BM, int BN, int WM, int WN>
__global__ __launch_bounds__(256, 2) void dense_f64_kernel(const double* __restrict__ W,
                                                           const double* __restrict__ bias,
                                                           const double* __restrict__ Hin,
                                                           double* __restrict__ Hout, int out, int in,
                                                           int64_t B, int act, int nMt, int64_t nNt) {
  constexpr int BK = 16;
  constexpr int BMP = BM + 16;
  constexpr int BKP = 18;
  constexpr int TM = BM / WM / 16;  // feature tiles per wave
  constexpr int TN = BN / WN / 16;  // batch tiles per wave
  constexpr int WREGS = BK * BM / 256;
  constexpr int HREGS = BK * BN / 256;
  static_assert(BM % 32 == 0 && WM * WN == 4, "tile shape");
  extern __shared__ double smem[];
  double* sW = smem;                  // [2][BK][BMP]
  double* sH = smem + 2 * BK * BMP;   // [2][BN][BKP]

  // XCD-aware block -> tile map: blocks b and b+8 share an XCD (its L2); give the nMt feature tiles of one
  // batch panel to consecutive blocks of ONE XCD so the Hin panel is fetched into that L2 once.
  const int64_t bid = blockIdx.x;
  const int xcd = (int)(bid & 7);
  const int64_t j = bid >> 3;
  const int mt = (int)(j % nMt);
  const int64_t nt = (j / nMt) * 8 + xcd;
  if (nt >= nNt) return;  // uniform per block: whole workgroup exits before any barrier

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wm = wave % WM, wn = wave / WM;
  const int i0 = mt * BM;
  const int64_t b0 = nt * BN;
  const int q = lane >> 4, c = lane & 15;

  d4 acc[TM][TN];
#pragma unroll
  for (int a = 0; a < TM; ++a)
#pragma unroll
    for (int b = 0; b < TN; ++b) acc[a][b] = (d4){0.0, 0.0, 0.0, 0.0};

  double wreg[WREGS], hreg[HREGS];
  const int nk = (in + BK - 1) / BK;

  auto load_tiles = [&](int kt) {
    const int k0 = kt * BK;
#pragma unroll
    for (int r = 0; r < WREGS; ++r) {
      const int idx = tid + 256 * r;
      const int ii = idx % BM, k = idx / BM;
      const int gi = i0 + ii, gk = k0 + k;
      wreg[r] = (gi < out && gk < in) ? W[gi + (int64_t)out * gk] : 0.0;
    }
#pragma unroll
    for (int r = 0; r < HREGS; ++r) {
      const int idx = tid + 256 * r;
      const int kk = idx & 15, b = idx >> 4;
      const int gk = k0 + kk;
      const int64_t gb = b0 + b;
      hreg[r] = (gb < B && gk < in) ? Hin[gk + (int64_t)in * gb] : 0.0;
    }
  };
  auto store_tiles = [&](int buf) {
    double* w = sW + buf * BK * BMP;
    double* h = sH + buf * BN * BKP;
#pragma unroll
    for (int r = 0; r < WREGS; ++r) {
      const int idx = tid + 256 * r;
      w[(idx / BM) * BMP + (idx % BM)] = wreg[r];
    }
#pragma unroll
    for (int r = 0; r < HREGS; ++r) {
      const int idx = tid + 256 * r;
      h[(idx >> 4) * BKP + (idx & 15)] = hreg[r];
    }
  };

  load_tiles(0);
  store_tiles(0);
  __syncthreads();

  for (int kt = 0; kt < nk; ++kt) {
    const int buf = kt & 1;
    if (kt + 1 < nk) load_tiles(kt + 1);  // global loads in flight under the MFMAs below
    const double* w = sW + buf * BK * BMP + wm * (BM / WM) + c;
    const double* h = sH + buf * BN * BKP + (wn * (BN / WN) + c) * BKP;
#pragma unroll
    for (int s = 0; s < BK / 4; ++s) {
      double wf[TM], hf[TN];
#pragma unroll
      for (int a = 0; a < TM; ++a) wf[a] = w[(4 * s + q) * BMP + a * 16];
#pragma unroll
      for (int b = 0; b < TN; ++b) hf[b] = h[b * 16 * BKP + 4 * s + q];
#pragma unroll
      for (int a = 0; a < TM; ++a)
#pragma unroll
        for (int b = 0; b < TN; ++b)
          acc[a][b] = __builtin_amdgcn_mfma_f64_16x16x4f64(hf[b], wf[a], acc[a][b], 0, 0, 0);
    }
    if (kt + 1 < nk) store_tiles(buf ^ 1);
    __syncthreads();
  }

  // epilogue: bias + activation, 128-B contiguous runs along i
#pragma unroll
  for (int a = 0; a < TM; ++a) {
    const int gi = i0 + wm * (BM / WM) + a * 16 + c;
    const double bv = (gi < out) ? bias[gi] : 0.0;
#pragma unroll
    for (int b = 0; b < TN; ++b) {
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int64_t gb = b0 + wn * (BN / WN) + b * 16 + q + 4 * r;
        if (gi < out && gb < B) Hout[gi + (int64_t)out * gb] = apply_act(acc[a][b][r] + bv, act);
      }
    }
  }
}

template <int BM, int BN, int WM, int WN>
static void launch_dense_cfg(hipStream_t st, const double* W, const double* bias, const double* Hin,
                             double* Hout, int32_t out, int32_t in, int64_t B, int32_t act) {
  constexpr size_t lds = 2 * (16 * (BM + 16) + BN * 18) * sizeof(double);
  static bool attr_set = false;
  auto kern = dense_f64_kernel<BM, BN, WM, WN>;
  if (!attr_set) {
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(kern),
                              hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    attr_set = true;
  }
  const int nMt = (out + BM - 1) / BM;
  const int64_t nNt = (B + BN - 1) / BN;
  const int64_t groups = (nNt + 7) / 8;  // batch panels per XCD lane
  const int64_t grid = groups * nMt * 8;
  hipLaunchKernelGGL(kern, dim3((unsigned)grid), dim3(256), lds, st, W, bias, Hin, Hout, (int)out, (int)in,
                     B, (int)act, nMt, nNt);
}

void launch_dense_f64(hipStream_t st, const double* W, const double* bias, const double* Hin,
                      double* Hout, int32_t out, int32_t in, int64_t B, int32_t act) {
  if (out <= 32)
    launch_dense_cfg<32, 128, 1, 4>(st, W, bias, Hin, Hout, out, in, B, act);
  else if (out <= 64)
    launch_dense_cfg<64, 128, 2, 2>(st, W, bias, Hin, Hout, out, in, B, act);
  else
    launch_dense_cfg<128, 128, 2, 2>(st, W, bias, Hin, Hout, out, in, B, act);
}

}  // namespace si
