// Host side of chains that contain Conv / MaxPool / flatten layers: validation of the caller's layer
// table, the geometry every kernel launch needs, and the generic forward pass / reverse sweep over such a chain.
// The reference's `model_re` restructures ANY Flux Chain (src/libs.jl:55-57) and `density` evaluates it on the full data
// (src/space_inference.jl:94); pure Dense chains keep their tuned path in capi_infer.hip / capi_train.hip (fused narrow tail,
// chain batching) and only pass through net_plan() for validation.  No arithmetic happens on the host.
#include <algorithm>

#include "kernels_gemm.h"
#include "si_internal.h"

namespace si {

static int even(int c) { return (c + 1) & ~1; }
static int conv_out(int wi, int k, int s, int p, int d) { return (wi + 2 * p - d * (k - 1) - 1) / s + 1; }

int32_t net_plan(Ctx* c, const char* who, const si_layer* layers, int L, int64_t N, int in_dim, int out_dim, NetPlan& out) {
  auto bad = [&](const std::string& m) { return fail(c, SI_ERR_INVALID, std::string(who) + ": " + m); };
  NetPlan p;
  p.L.resize((size_t)L);
  bool spatial = false;  // inside a convolutional stack: activations are (W, H, C) images
  int W = 0, H = 0, C = 0;
  int feat = in_dim;
  size_t pack = 0;
  for (int l = 0; l < L; ++l) {
    const si_layer& ly = layers[l];
    LayerPlan& q = p.L[(size_t)l];
    q.kind = ly.kind;
    q.act = ly.act;
    q.w_off = ly.w_off;
    q.b_off = ly.b_off;
    q.in_feat = ly.in;
    q.out_feat = ly.out;
    if (ly.in != feat || ly.out <= 0) return bad("layer dimensions do not chain");
    if (ly.kind == SI_LAYER_DENSE) {
      // a Dense layer on image-shaped activations without a flatten in between fails in Flux too (DimensionMismatch)
      if (spatial) return bad("DimensionMismatch: Dense layer applied to (W, H, C, N) activations (flatten missing)");
      if (ly.act < 0 || ly.act >= SI_ACT_COUNT) return bad("unknown activation");
      if (ly.w_off < 0 || ly.b_off < 0 || ly.w_off + (int64_t)ly.in * ly.out > N || ly.b_off + ly.out > N)
        return bad("layer offsets outside the flat weight vector");
      q.in_elems = ly.in;
      q.out_elems = ly.out;
      p.max_rows = std::max(p.max_rows, ly.out);
    } else if (ly.kind == SI_LAYER_CONV || ly.kind == SI_LAYER_MAXPOOL || ly.kind == SI_LAYER_FLATTEN) {
      p.has_conv = true;
      if (ly.wi <= 0 || ly.hi <= 0 || ly.cin <= 0) return bad("Conv / MaxPool / flatten layer without its input geometry");
      if (!spatial) {
        if (l != 0) return bad("a Conv / MaxPool / flatten layer cannot follow a Dense layer (no reshape layer is supported)");
        p.input_spatial = true;
        p.in_W = W = ly.wi;
        p.in_H = H = ly.hi;
        p.in_C = C = ly.cin;
        p.in_Cp = even(C);
        spatial = true;
      }
      if (ly.wi != W || ly.hi != H || ly.cin != C || (int64_t)W * H * C != ly.in) return bad("layer geometry does not chain");
      q.C = C;
      q.Cp = even(C);
      q.Wi = W;
      q.Hi = H;
      q.in_elems = (int64_t)q.Cp * W * H;
      if (ly.kind == SI_LAYER_FLATTEN) {
        if (ly.out != ly.in) return bad("flatten: in != out");
        q.Co = C;
        q.Cop = q.Cp;
        q.Wo = W;
        q.Ho = H;
        q.out_elems = ly.out;
        spatial = false;
      } else {
        if (ly.kw <= 0 || ly.kh <= 0 || ly.sw <= 0 || ly.sh <= 0) return bad("kernel / window and stride must be positive");
        q.KW = ly.kw;
        q.KH = ly.kh;
        q.sw = ly.sw;
        q.sh = ly.sh;
        if (ly.kind == SI_LAYER_MAXPOOL) {
          if (ly.cout != C) return bad("MaxPool: cin != cout");
          if (ly.pw != 0 || ly.ph != 0) return bad("MaxPool: only pad = 0 (Flux's default) is implemented");
          if (ly.kw > W || ly.kh > H) return bad("MaxPool window larger than its input");
          q.Co = C;
          q.Cop = q.Cp;
          q.Wo = (W - ly.kw) / ly.sw + 1;
          q.Ho = (H - ly.kh) / ly.sh + 1;
        } else {
          if (ly.act < 0 || ly.act >= SI_ACT_COUNT) return bad("unknown activation");
          if (ly.cout <= 0 || ly.pw < 0 || ly.ph < 0 || ly.dw <= 0 || ly.dh <= 0) return bad("Conv: bad cout / pad / dilation");
          const int64_t nw = (int64_t)ly.kw * ly.kh * C * ly.cout;
          if (ly.w_off < 0 || ly.b_off < 0 || ly.w_off + nw > N || ly.b_off + ly.cout > N)
            return bad("layer offsets outside the flat weight vector");
          q.Co = ly.cout;
          q.Cop = even(ly.cout);
          q.Wo = conv_out(W, ly.kw, ly.sw, ly.pw, ly.dw);
          q.Ho = conv_out(H, ly.kh, ly.sh, ly.ph, ly.dh);
          if (q.Wo <= 0 || q.Ho <= 0) return bad("Conv: kernel larger than the padded input");
          q.Kp = (q.Cp * ly.kw * ly.kh + 15) / 16 * 16;
          q.KpT = (q.Cop * ly.kw * ly.kh + 15) / 16 * 16;
          q.wp_off = pack;
          pack += (size_t)q.Cop * q.Kp;
          q.bp_off = pack;
          pack += (size_t)q.Cop;
          pack = (pack + 1) & ~(size_t)1;  // keep every pack 16-byte aligned
          p.wt_elems = std::max(p.wt_elems, (size_t)q.Cp * q.KpT);
          // forward gather: patches of the INPUT tensor, positions = output pixels
          q.g = ConvGeom{q.Cp, W, H, q.Wo, q.Ho, ly.kw, ly.kh, ly.sw, ly.sh, 1, 1, ly.pw, ly.ph, ly.dw, ly.dh,
                         q.Cp * ly.kw * ly.kh, (int64_t)q.Cp * W * H};
          // data gradient: patches of the DELTA tensor (Cop channels on the Wo x Ho grid), positions = input pixels,
          // wo = (wi - (dil*(K-1) - pad) + a'*dil) / stride
          q.gT = ConvGeom{q.Cop, q.Wo, q.Ho, W, H, ly.kw, ly.kh, 1, 1, ly.sw, ly.sh, ly.dw * (ly.kw - 1) - ly.pw,
                          ly.dh * (ly.kh - 1) - ly.ph, ly.dw, ly.dh, q.Cop * ly.kw * ly.kh, (int64_t)q.Cop * q.Wo * q.Ho};
          p.max_rows = std::max(p.max_rows, std::max(q.Cop, q.Cp));
        }
        if ((int64_t)q.Wo * q.Ho * q.Co != ly.out) return bad("layer output size does not match its geometry");
        q.out_elems = (int64_t)q.Cop * q.Wo * q.Ho;
        W = q.Wo;
        H = q.Ho;
        C = q.Co;
      }
    } else {
      return fail(c, SI_ERR_INVALID, "Error: model_re function is not available for this model (unknown layer kind)");
    }
    feat = ly.out;
    p.max_elems = std::max(p.max_elems, std::max(q.in_elems, q.out_elems));
  }
  if (spatial) return bad("the chain must end on (features x B) activations (flatten / Dense after the last Conv)");
  if (feat != out_dim) return bad("last layer width != out_dim");
  p.in_elems = p.input_spatial ? (int64_t)p.in_Cp * p.in_W * p.in_H : in_dim;
  p.max_elems = std::max(p.max_elems, p.in_elems);
  p.wpack_elems = std::max<size_t>(pack, 2);
  out = std::move(p);
  return SI_OK;
}

static bool net_pool_fusable(const NetPlan& p, size_t l) {
  if (l + 1 >= p.L.size() || p.L[l].kind != SI_LAYER_CONV) return false;
  const LayerPlan &q = p.L[l], &m = p.L[l + 1];
  return m.kind == SI_LAYER_MAXPOOL && m.KW == 2 && m.KH == 2 && m.sw == 2 && m.sh == 2 && q.Wo % 2 == 0 && q.Ho % 2 == 0;
}
// (the four later activations are applied to the POOLED tensor by the fused kernel: their window index would be taken on
// pre-activations, where NNlib's approximate-equality test can differ in the last bits -- they keep the un-fused route)
bool net_grad_fused(const NetPlan& p, size_t l) { return net_pool_fusable(p, l) && !act_is_extra(p.L[l].act); }
size_t net_pidx_bytes(const NetPlan& p, size_t l, int64_t B) {
  const LayerPlan& q = p.L[l];
  return (size_t)q.Cop * (size_t)(q.Wo / 2) * (size_t)(q.Ho / 2) * (size_t)B;
}

void net_input(Ctx* c, const NetPlan& p, const double* X, double* Xc, int64_t B) {
  launch_whcn_to_cwhn(c->stream, X, Xc, p.in_W, p.in_H, p.in_C, p.in_Cp, B);
}

int32_t net_forward(Ctx* c, const NetPlan& p, const double* w, const double* xin, int64_t B, double* const* outs, double* wpack,
                    bool pingpong, double** final_out, uint8_t* const* pidx) {
  hipStream_t st = c->stream;
  const double* h = xin;
  size_t executed = 0;
  for (size_t l = 0; l < p.L.size(); ++l) {
    const LayerPlan& q = p.L[l];
    double* o = pingpong ? outs[executed & 1] : outs[l];
    ++executed;
    if (final_out) *final_out = o;
    if ((double)std::max(q.in_elems, q.out_elems) * (double)B >= 2147483648.0)
      return fail(c, SI_ERR_INVALID, "activation tensors of 2^31 elements or more are not supported by the conv kernels");
    switch (q.kind) {
      case SI_LAYER_DENSE: {
        ProfScope ps(c, SI_K_DENSE, 2.0 * (double)q.in_feat * q.out_feat * (double)B,
                     ((double)q.in_feat * q.out_feat + q.out_feat + (double)(q.in_feat + q.out_feat) * (double)B) * 8.0);
        if (dense_narrow_applies(q.out_feat, q.in_feat, B, c->num_cu))
          launch_dense_narrow(st, w + q.w_off, w + q.b_off, h, o, q.out_feat, q.in_feat, B, q.act);
        else
          launch_dense_f64(st, w + q.w_off, w + q.b_off, h, o, q.out_feat, q.in_feat, B, q.act);
        break;
      }
      case SI_LAYER_CONV: {
        const int64_t npos = (int64_t)q.Wo * q.Ho * B;
        {
          ProfScope ps(c, SI_K_CONV_AUX, 0.0, ((double)q.KW * q.KH * q.C * q.Co + (double)q.Cop * q.Kp) * 8.0);
          launch_conv_pack(st, w + q.w_off, w + q.b_off, wpack + q.wp_off, wpack + q.bp_off, q.KW, q.KH, q.C, q.Co, q.Cp, q.Cop, q.Kp);
        }
        if (!pingpong && pidx && pidx[l] && net_grad_fused(p, l)) {
          // gradient mode: conv + pool in one kernel; the pooled output and a byte index are all the reverse sweep needs
          o = outs[l + 1];
          if (final_out) *final_out = o;
          ProfScope ps(c, SI_K_CONV, 2.0 * (double)q.KW * q.KH * q.C * q.Co * (double)npos,
                       ((double)q.in_elems + (double)p.L[l + 1].out_elems * 1.125) * (double)B * 8.0 + (double)q.Cop * q.Kp * 8.0);
          launch_conv_forward_pool2_idx(st, wpack + q.wp_off, wpack + q.bp_off, h, o, pidx[l], q.g, q.Cop, q.Kp, npos, q.act);
          ++l;
          ++executed;
          break;
        }
        const bool fuse_pool = pingpong && net_pool_fusable(p, l);
        if (fuse_pool) {   // the pooled tensor is all the next layer reads: skip the MaxPool layer
          ProfScope ps(c, SI_K_CONV, 2.0 * (double)q.KW * q.KH * q.C * q.Co * (double)npos,
                       ((double)q.in_elems + (double)p.L[l + 1].out_elems) * (double)B * 8.0 + (double)q.Cop * q.Kp * 8.0);
          launch_conv_forward_pool2(st, wpack + q.wp_off, wpack + q.bp_off, h, o, q.g, q.Cop, q.Kp, npos, q.act);
          ++l;
          break;
        }
        ProfScope ps(c, SI_K_CONV, 2.0 * (double)q.KW * q.KH * q.C * q.Co * (double)npos,
                     ((double)q.in_elems + (double)q.out_elems) * (double)B * 8.0 + (double)q.Cop * q.Kp * 8.0);
        launch_conv_forward(st, wpack + q.wp_off, wpack + q.bp_off, h, o, q.g, q.Cop, q.Kp, npos, q.act);
        break;
      }
      case SI_LAYER_MAXPOOL: {
        ProfScope ps(c, SI_K_CONV_AUX, 0.0, ((double)q.in_elems + (double)q.out_elems) * (double)B * 8.0);
        launch_maxpool(st, h, o, q.Cp, q.Wi, q.Hi, q.Wo, q.Ho, q.KW, q.KH, q.sw, q.sh, B);
        break;
      }
      default: {  // flatten: channel-fastest -> the reference's (W, H, C) feature order
        ProfScope ps(c, SI_K_CONV_AUX, 0.0, ((double)q.in_elems + (double)q.out_elems) * (double)B * 8.0);
        launch_cwhn_to_whcn(st, h, o, q.Wi, q.Hi, q.C, q.Cp, B);
        break;
      }
    }
    h = o;
  }
  hipError_t e = hipGetLastError();
  if (e != hipSuccess) return fail(c, SI_ERR_HIP, std::string("net_forward: ") + hipGetErrorString(e));
  return SI_OK;
}

// compute_dtype = SI_F32 on a Conv chain: the forward pass of net_forward (ping-pong activations, Conv + 2x2 MaxPool fused where it
// applies) on fp32 operands -- the conv kernels compiled for float (kernels_conv.hip -DSI_CONV_F32: v_mfma_f32_16x16x4_f32), the
// Dense layers behind `flatten` on kernels_gemm_f32.hip.  w32: the evaluation's weights rounded once from the fp64 sum (K4).
int32_t net_forward_f32(Ctx* c, const NetPlan& p, const float* w, const float* xin, int64_t B, float* const* outs, float* wpack,
                        float** final_out) {
  hipStream_t st = c->stream;
  const float* h = xin;
  size_t executed = 0;
  for (size_t l = 0; l < p.L.size(); ++l) {
    const LayerPlan& q = p.L[l];
    float* o = outs[executed & 1];
    ++executed;
    if (final_out) *final_out = o;
    if ((double)std::max(q.in_elems, q.out_elems) * (double)B >= 2147483648.0)
      return fail(c, SI_ERR_INVALID, "activation tensors of 2^31 elements or more are not supported by the conv kernels");
    switch (q.kind) {
      case SI_LAYER_DENSE: {
        ProfScope ps(c, SI_K_DENSE, 2.0 * (double)q.in_feat * q.out_feat * (double)B,
                     ((double)q.in_feat * q.out_feat + q.out_feat + (double)(q.in_feat + q.out_feat) * (double)B) * 4.0);
        if (dense_narrow_applies(q.out_feat, q.in_feat, B, c->num_cu))
          launch_dense_narrow(st, w + q.w_off, w + q.b_off, h, o, q.out_feat, q.in_feat, B, q.act);
        else
          launch_dense_f32(st, w + q.w_off, w + q.b_off, h, o, q.out_feat, q.in_feat, B, q.act);
        break;
      }
      case SI_LAYER_CONV: {
        const int64_t npos = (int64_t)q.Wo * q.Ho * B;
        {
          ProfScope ps(c, SI_K_CONV_AUX, 0.0, ((double)q.KW * q.KH * q.C * q.Co + (double)q.Cop * q.Kp) * 4.0);
          launch_conv_pack(st, w + q.w_off, w + q.b_off, wpack + q.wp_off, wpack + q.bp_off, q.KW, q.KH, q.C, q.Co, q.Cp, q.Cop, q.Kp);
        }
        if (net_pool_fusable(p, l)) {   // the pooled tensor is all the next layer reads: skip the MaxPool layer
          ProfScope ps(c, SI_K_CONV, 2.0 * (double)q.KW * q.KH * q.C * q.Co * (double)npos,
                       ((double)q.in_elems + (double)p.L[l + 1].out_elems) * (double)B * 4.0 + (double)q.Cop * q.Kp * 4.0);
          launch_conv_forward_pool2(st, wpack + q.wp_off, wpack + q.bp_off, h, o, q.g, q.Cop, q.Kp, npos, q.act);
          ++l;
          break;
        }
        ProfScope ps(c, SI_K_CONV, 2.0 * (double)q.KW * q.KH * q.C * q.Co * (double)npos,
                     ((double)q.in_elems + (double)q.out_elems) * (double)B * 4.0 + (double)q.Cop * q.Kp * 4.0);
        launch_conv_forward(st, wpack + q.wp_off, wpack + q.bp_off, h, o, q.g, q.Cop, q.Kp, npos, q.act);
        break;
      }
      case SI_LAYER_MAXPOOL: {
        ProfScope ps(c, SI_K_CONV_AUX, 0.0, ((double)q.in_elems + (double)q.out_elems) * (double)B * 4.0);
        launch_maxpool(st, h, o, q.Cp, q.Wi, q.Hi, q.Wo, q.Ho, q.KW, q.KH, q.sw, q.sh, B);
        break;
      }
      default: {  // flatten: channel-fastest -> the reference's (W, H, C) feature order
        ProfScope ps(c, SI_K_CONV_AUX, 0.0, ((double)q.in_elems + (double)q.out_elems) * (double)B * 4.0);
        launch_cwhn_to_whcn(st, h, o, q.Wi, q.Hi, q.C, q.Cp, B);
        break;
      }
    }
    h = o;
  }
  hipError_t e = hipGetLastError();
  if (e != hipSuccess) return fail(c, SI_ERR_HIP, std::string("net_forward_f32: ") + hipGetErrorString(e));
  return SI_OK;
}

void net_scratch_sizes(const NetPlan& p, int64_t B, int num_cu, size_t* bwpart, size_t* rspart, size_t* wt, size_t* dbtmp) {
  size_t part = 1;
  for (const LayerPlan& q : p.L) {
    if (q.kind == SI_LAYER_DENSE) {
      part = std::max(part, backward_weight_part_elems(q.out_feat, q.in_feat, B, num_cu));
    } else if (q.kind == SI_LAYER_CONV) {
      part = std::max(part, (size_t)conv_dw_max_splits(q.Cop, q.Kp, (int64_t)q.Wo * q.Ho * B, num_cu) * q.Cop * q.Kp);
    }
  }
  *bwpart = part;
  *rspart = std::max((size_t)rowsum_chunks() * (size_t)p.max_rows, dact_rowsum_ws_elems(p.max_rows));
  *wt = p.wt_elems;
  *dbtmp = (size_t)p.max_rows;
}

int32_t net_backward(Ctx* c, const NetPlan& p, const double* w, const double* xin, int64_t B, double* const* hs, double* g0,
                     double* g1, double* gw, const NetScratch& s) {
  hipStream_t st = c->stream;
  double* g = g0;      // gradient with respect to the OUTPUT of the layer being processed (device layout)
  double* gn = g1;
  bool delta_ready = false;  // g already holds Delta = dL/d(pre-activation) of the layer (and its db is written)
  for (size_t li = p.L.size(); li-- > 0;) {
    const LayerPlan& q = p.L[li];
    const double* hin = li > 0 ? hs[li - 1] : xin;
    const double* hout = hs[li];
    switch (q.kind) {
      case SI_LAYER_DENSE: {
        // Delta = g .* act'(h) (in place) and db = its row sums, one pass
        launch_mul_dact_rowsum(st, g, hout, q.out_feat, B, q.act, g, s.rspart, q.out_feat, gw + q.b_off);
        launch_backward_weight(st, g, hin, s.bwpart, q.out_feat, q.in_feat, B, c->num_cu, gw + q.w_off);   // dW
        if (li > 0) launch_backward_data(st, w + q.w_off, g, hin, gn, q.out_feat, q.in_feat, B, SI_ACT_IDENTITY);  // W' Delta
        break;
      }
      case SI_LAYER_FLATTEN:
        launch_whcn_to_cwhn(st, g, gn, q.Wi, q.Hi, q.C, q.Cp, B);   // back to channel-fastest, pad channels zero
        break;
      case SI_LAYER_MAXPOOL:
        if (li > 0 && s.pidx && s.pidx[li - 1] && net_grad_fused(p, li - 1)) {
          // the pair ran fused in the forward pass: Delta of the conv layer from (pooled gradient, pooled output, byte index)
          const LayerPlan& cv = p.L[li - 1];
          launch_pool2_bwd_idx(st, g, hout, s.pidx[li - 1], gn, cv.Cop, q.Wo, q.Ho, B, cv.act, s.rspart, cv.Co, gw + cv.b_off);
          delta_ready = true;
        } else if (li > 0 && p.L[li - 1].kind == SI_LAYER_CONV) {
          // the pool's input is a conv layer's output: route the gradient to the window maxima, multiply by act' and sum
          // the conv layer's bias gradient in the same pass
          const LayerPlan& cv = p.L[li - 1];
          launch_maxpool_bwd_dact_rowsum(st, hin, hout, g, gn, q.Cp, q.Wi, q.Hi, q.Wo, q.Ho, q.KW, q.KH, q.sw, q.sh, B, cv.act,
                                         s.rspart, cv.Co, gw + cv.b_off);
          delta_ready = true;
        } else {
          launch_maxpool_bwd(st, hin, hout, g, gn, q.Cp, q.Wi, q.Hi, q.Wo, q.Ho, q.KW, q.KH, q.sw, q.sh, B);
        }
        break;
      default: {  // Conv
        const int64_t npos = (int64_t)q.Wo * q.Ho * B;
        if (!delta_ready) launch_mul_dact_rowsum(st, g, hout, q.Cop, npos, q.act, g, s.rspart, q.Co, gw + q.b_off);
        delta_ready = false;
        int64_t ks;
        const int ns = conv_dw_splits(q.Cop, q.Kp, npos, c->num_cu, &ks);
        launch_conv_backward_weight(st, g, hin, s.bwpart, q.g, q.Cop, q.Kp, npos, ns, ks);
        launch_conv_unpack_dw(st, s.bwpart, ns, gw + q.w_off, q.KW, q.KH, q.C, q.Co, q.Cp, q.Cop, q.Kp);
        if (li > 0) {
          launch_conv_pack_t(st, w + q.w_off, s.wt, q.KW, q.KH, q.C, q.Co, q.Cp, q.Cop, q.KpT);
          launch_conv_backward_data(st, s.wt, g, gn, q.gT, q.Cp, q.KpT, (int64_t)q.Wi * q.Hi * B);
        }
        break;
      }
    }
    std::swap(g, gn);
  }
  hipError_t e = hipGetLastError();
  if (e != hipSuccess) return fail(c, SI_ERR_HIP, std::string("net_backward: ") + hipGetErrorString(e));
  return SI_OK;
}

}  // namespace si
