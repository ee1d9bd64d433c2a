// Backward kernels of the two MLPs (what loss.backward() computes through model/network/mlp.py via autograd in the
// reference; the explicit sweeps are SURVEY.md appendix A.3/A.4).  Same execution model as the forward kernels:
// 32 points per wave, activations/adjoints in registers, weights streamed through LDS.
//   sdf_bwd : sweep 1 (adjoint of the d sdf/dx chain, bottom-up, forward-direction weights) followed by
//             sweep 2 (ordinary backward, top-down, transposed weights).  Emits per layer the operands of the
//             weight-gradient GEMMs (G(ubar_l), G(a_l)); the GEMMs themselves run in wgrad.hip.
//   rgb_bwd : first-order backward of the radiance net; emits G(a_l) and the feature gradient fbar.
#include "ksplit.h"
#include "mlp_args.h"

using namespace i2sdf;

int i2sdf_hip_check(hipError_t e, const char* what);


namespace {

__host__ __device__ constexpr int sdf_fwd_hidden_stages(int H, int PEC, int L, bool has_skip) {
  int c = op_chunks(H / 32, PEC);
  for (int l = 1; l < L - 1; ++l) c += op_chunks(H / 32, H / 8);
  if (has_skip) c += op_chunks(H / 32, H / 8 + PEC) - op_chunks(H / 32, H / 8);
  return c / SC;
}
// reverse stream: [w_sdf][W_feat^T][w_sdf][W_{L-2}^T .. W_1^T]  (W_0^T is not needed: the points carry no gradient)
__host__ __device__ constexpr int sdf_rev_bwd_stages(int H, int F, int PEC, int L, bool has_skip) {
  const int PT = cdiv(PEC * 8, 32);
  int c = bwd_op_chunks(H / 32, F / 8) + 2 * rowvec_chunks(H / 8, 1);
  for (int l = L - 2; l >= 1; --l) c += bwd_op_chunks(H / 32, H / 8);
  if (has_skip) c += bwd_op_chunks(H / 32 + PT, H / 8) - bwd_op_chunks(H / 32, H / 8);
  return c / SC;
}

template <int H, int F, int LF>
__global__ __launch_bounds__(256) void sdf_bwd_kernel(SdfBwdArgs a) {
  constexpr int NT = H / 32, KC = H / 8, PEC = PE<LF>::PEC, PT = cdiv(PEC * 8, 32), FC = F / 8;
  extern __shared__ __attribute__((aligned(16))) float lds[];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, hi = lane >> 5;
  const int64_t m = ((int64_t)(blockIdx.x + a.wg0) * 4 + wave) * 32 + (lane & 31);
  const bool valid = m < a.M;
  const int64_t mc = valid ? m : a.M - 1;
  const int64_t lstride = a.Mp * H;
  float gp[PEC * 4];
  {
    float px, py, pz, full[PEC * 8], coef[PEC * 8], nb[3] = {0.f, 0.f, 0.f};
    fetch_point(a.pts, mc, px, py, pz);
    pe_full<LF>(px, py, pz, full);
    pe_coef<LF>(full, coef);
    if (a.nbar) { nb[0] = a.nbar[mc * 3 + 0]; nb[1] = a.nbar[mc * 3 + 1]; nb[2] = a.nbar[mc * 3 + 2]; }
    pe_j_apply<LF>(coef, nb, hi, gp);
    store_regs<PEC>(a.gpbar + m * (PEC * 8), hi, valid, gp);
  }
  constexpr bool PRE = KC >= SC;
  WStream ws;
  // ------------------------------ sweep 1: bottom-up ------------------------------
  ws.begin(a.fwd, lds, a.n_fwd, tid);
  float gh[KC * 4];
  f32x16 acc[NT];
  for (int l = 0; l < a.L - 1; ++l) {
    // epilogue: G(hbar_{l+1}) = G(abar_l) * sigma_l ; G2(a_l) = G(abar_l) * abar_l * 100 (1 - sigma_l)
    const float* hrow = a.hs + l * lstride + mc * H;
    const float* arow = a.abars + l * lstride + mc * H;
    float* g2row = a.gas + l * lstride + m * H;
    float* gurow = a.gus + (l + 1) * lstride + m * H;
    if (l == 0) {
      Sweep1Epi<false> e{hrow, arow, g2row, gurow, hi, valid};
      dense_op_epi<NT, PEC, NT * 4, 1, 0, Sweep1Epi<false>>(ws, gp, acc, e, tid);
    } else if (l == a.skip) {
      float u[(KC + PEC) * 4];
#pragma unroll
      for (int i = 0; i < KC * 4; ++i) u[i] = gh[i] * RS2;
#pragma unroll
      for (int i = 0; i < PEC * 4; ++i) u[KC * 4 + i] = gp[i] * RS2;
      Sweep1Epi<PRE> e{hrow, arow, g2row, gurow, hi, valid};
      dense_op_epi<NT, KC + PEC, NT * 4, 1, 0, Sweep1Epi<PRE>>(ws, u, acc, e, tid);
    } else {
      Sweep1Epi<PRE> e{hrow, arow, g2row, gurow, hi, valid};
      dense_op_epi<NT, KC, NT * 4, 1, 0, Sweep1Epi<PRE>>(ws, gh, acc, e, tid);
    }
    commit_tiles<NT>(acc, gh);
  }
  // ------------------------------ sweep 2: top-down ------------------------------
  __syncthreads();
  ws.begin(a.rev, lds, a.n_rev, tid);
  float ga[KC * 4];
  {
    float wv[KC * 4];
    f32x4 sc;
    rowvec_load<KC>(ws, wv, sc, tid);
    float fb[FC * 4];
    const bool hasf = a.fbar != nullptr && mc < a.m_fbar;
    if (hasf) load_regs<FC>(a.fbar + mc * F, hi, fb);
    else {
#pragma unroll
      for (int i = 0; i < FC * 4; ++i) fb[i] = 0.f;
    }
    const float sb = a.sbar ? a.sbar[mc] : 0.f;
    if (valid && hi == 0) {
      *reinterpret_cast<f32x4*>(a.ga_last4 + m * 4) = f32x4{sb, 0.f, 0.f, 0.f};
      *reinterpret_cast<f32x4*>(a.ones4 + m * 4) = f32x4{1.f, 0.f, 0.f, 0.f};
    }
    const int lt = a.L - 2;
    constexpr bool PRET = FC >= SC;
    Sweep2TopEpi<NT, PRET> te{a.hs + lt * lstride + mc * H, a.gas + lt * lstride + mc * H, a.gas + lt * lstride + m * H, hi, valid, sb, wv};
    dense_op_epi<NT, FC, 0, 1, 0, Sweep2TopEpi<NT, PRET>>(ws, fb, acc, te, tid);      // -> G(a_{L-2})
    commit_tiles<NT>(acc, ga);
    ws.skip(rowvec_chunks(KC, 1) / SC, tid);                                             // the chain's copy of w_sdf
  }
  for (int l = a.L - 2; l >= 1; --l) {
    // G(h_l) = W_l^T G(a_l); epilogue -> G(a_{l-1}) = G(h_l) * sigma_{l-1} + G2(a_{l-1})
    const float* hrow = a.hs + (l - 1) * lstride + mc * H;
    const float* g2row = a.gas + (l - 1) * lstride + mc * H;
    float* grow = a.gas + (l - 1) * lstride + m * H;
    if (l == a.skip) {
      f32x16 as[NT + PT];
      Sweep2Epi<NT, PRE> e{hrow, g2row, grow, hi, valid, RS2};
      dense_op_epi<NT + PT, KC, 0, 1, 0, Sweep2Epi<NT, PRE>>(ws, ga, as, e, tid);
#pragma unroll
      for (int i = 0; i < KC * 4; ++i) ga[i] = as[i / 16][i % 16];
    } else {
      Sweep2Epi<NT, PRE> e{hrow, g2row, grow, hi, valid, 1.0f};
      dense_op_epi<NT, KC, 0, 1, 0, Sweep2Epi<NT, PRE>>(ws, ga, acc, e, tid);
      commit_tiles<NT>(acc, ga);
    }
  }
}

// Split-K variant of sdf_bwd_kernel for the last partial round (ksplit.h): one 32-point tile per workgroup.
template <int H, int F, int LF>
__global__ __launch_bounds__(256) void sdf_bwd_split_kernel(SdfBwdArgs a, int64_t m0) {
  constexpr int NT = H / 32, KC = H / 8, PEC = PE<LF>::PEC, PT = cdiv(PEC * 8, 32), FC = F / 8;
  static_assert(NT == 8 && KC == SC && FC == SC, "split-K kernels are built for 256-wide layers");
  extern __shared__ __attribute__((aligned(16))) float lds[];
  float* xlds = lds + 2 * STAGE_FLOATS;
  float* slds = xlds + KS_X_FLOATS;
  float* glds = slds + KS_S_FLOATS;
  const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6, hi = lane >> 5;
  const int64_t m = m0 + (int64_t)blockIdx.x * 32 + (lane & 31);
  const bool valid = m < a.M;
  const int64_t mc = valid ? m : a.M - 1;
  const int64_t lstride = a.Mp * H;
  float gp[PEC * 4];
  {
    float px, py, pz, full[PEC * 8], coef[PEC * 8], nb[3] = {0.f, 0.f, 0.f};
    fetch_point(a.pts, mc, px, py, pz);
    pe_full<LF>(px, py, pz, full);
    pe_coef<LF>(full, coef);
    if (a.nbar) { nb[0] = a.nbar[mc * 3 + 0]; nb[1] = a.nbar[mc * 3 + 1]; nb[2] = a.nbar[mc * 3 + 2]; }
    pe_j_apply<LF>(coef, nb, hi, gp);
    if (w == 0) store_regs<PEC>(a.gpbar + m * (PEC * 8), hi, valid, gp);
  }
  WStream ws;
  // ------------------------------ sweep 1 ------------------------------
  ws.begin(a.fwd, lds, a.n_fwd, tid);
  float ghq[32];
  for (int l = 0; l < a.L - 1; ++l) {
    const float* hrow = a.hs + l * lstride + mc * H;
    const float* arow = a.abars + l * lstride + mc * H;
    float* g2row = a.gas + l * lstride + m * H;
    float* gurow = a.gus + (l + 1) * lstride + m * H;
    if (l == 0 || l == a.skip) {
      // reduction length 5 / 37 chunks: not stage aligned -> every wave computes the layer (stores masked to its own tiles)
      f32x16 acc[NT];
      Sweep1Epi<false> e{hrow, arow, g2row, gurow, hi, valid};
      e.own = w;
      if (l == 0) {
        dense_op_epi<NT, PEC, NT * 4, 1, 0, Sweep1Epi<false>>(ws, gp, acc, e, tid);
      } else {
        float u[(KC + PEC) * 4];
        {
          float gh[NT * 16];
          gather_full(ghq, gh, glds, tid);
#pragma unroll
          for (int i = 0; i < KC * 4; ++i) u[i] = gh[i] * RS2;
        }
#pragma unroll
        for (int i = 0; i < PEC * 4; ++i) u[KC * 4 + i] = gp[i] * RS2;
        dense_op_epi<NT, KC + PEC, NT * 4, 1, 0, Sweep1Epi<false>>(ws, u, acc, e, tid);
      }
      float gh2[NT * 16];
      commit_tiles<NT>(acc, gh2);
      take_quarter(gh2, ghq, w);
    } else {
      Sweep1Epi<true> e{hrow, arow, g2row, gurow, hi, valid};
      float nq[32];
      dense_op_ksplit<NT, NT * 4, 1, Sweep1Epi<true>>(ws, ghq, nq, nullptr, e, xlds, tid);
#pragma unroll
      for (int i = 0; i < 32; ++i) ghq[i] = nq[i];
    }
  }
  // ------------------------------ sweep 2 ------------------------------
  __syncthreads();
  ws.begin(a.rev, lds, a.n_rev, tid);
  float gaq[32];
  {
    float wq[32];
    f32x4 sc;
    rowvec_load_quarter(ws, wq, sc, tid);
    float fbq[32];
    const bool hasf = a.fbar != nullptr && mc < a.m_fbar;
    if (hasf) load_quarter(a.fbar + mc * F, w, hi, fbq);
    else {
#pragma unroll
      for (int i = 0; i < 32; ++i) fbq[i] = 0.f;
    }
    const float sb = a.sbar ? a.sbar[mc] : 0.f;
    if (valid && hi == 0 && w == 0) {
      *reinterpret_cast<f32x4*>(a.ga_last4 + m * 4) = f32x4{sb, 0.f, 0.f, 0.f};
      *reinterpret_cast<f32x4*>(a.ones4 + m * 4) = f32x4{1.f, 0.f, 0.f, 0.f};
    }
    const int lt = a.L - 2;
    Sweep2TopEpiQ<true> te{a.hs + lt * lstride + mc * H, a.gas + lt * lstride + mc * H, a.gas + lt * lstride + m * H, hi, valid, sb, wq};
    dense_op_ksplit<NT, 0, 1, Sweep2TopEpiQ<true>>(ws, fbq, gaq, nullptr, te, xlds, tid);
    ws.skip(rowvec_chunks(KC, 1) / SC, tid);
  }
  for (int l = a.L - 2; l >= 1; --l) {
    const float* hrow = a.hs + (l - 1) * lstride + mc * H;
    const float* g2row = a.gas + (l - 1) * lstride + mc * H;
    float* grow = a.gas + (l - 1) * lstride + m * H;
    float nq[32];
    if (l == a.skip) {
      Sweep2Epi<NT, true> e{hrow, g2row, grow, hi, valid, RS2};
      dense_op_ksplit<NT + PT, 0, 1, Sweep2Epi<NT, true>>(ws, gaq, nq, nullptr, e, xlds, tid);
    } else {
      Sweep2Epi<NT, true> e{hrow, g2row, grow, hi, valid, 1.0f};
      dense_op_ksplit<NT, 0, 1, Sweep2Epi<NT, true>>(ws, gaq, nq, nullptr, e, xlds, tid);
    }
#pragma unroll
    for (int i = 0; i < 32; ++i) gaq[i] = nq[i];
  }
}

}  // namespace

// ---- radiance net backward ---------------------------------------------------------------------------------

namespace {

__host__ __device__ constexpr int rgb_rev_stages(int H, int F, int L) {
  int c = rowvec_chunks(H / 8, 3);
  for (int l = L - 2; l >= 1; --l) c += bwd_op_chunks(H / 32, H / 8);
  c += bwd_op_chunks(F / 32, H / 8);
  return c / SC;
}

template <int H, int F>
__global__ __launch_bounds__(256) void rgb_bwd_kernel(RgbBwdArgs a) {
  constexpr int NT = H / 32, KC = H / 8, FT = F / 32;
  extern __shared__ __attribute__((aligned(16))) float lds[];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, hi = lane >> 5;
  const int64_t m = ((int64_t)(blockIdx.x + a.wg0) * 4 + wave) * 32 + (lane & 31);
  const bool valid = m < a.M;
  const int64_t mc = valid ? m : a.M - 1;
  const int64_t lstride = a.Mp * H;
  float g3[3];
#pragma unroll
  for (int j = 0; j < 3; ++j) {
    const float c = a.rgb[mc * 3 + j];
    g3[j] = a.rgb_bar[mc * 3 + j] * c * (1.0f - c);
  }
  if (valid && hi == 0) *reinterpret_cast<f32x4*>(a.ga_last + m * 4) = f32x4{g3[0], g3[1], g3[2], 0.f};
  WStream ws;
  ws.begin(a.rev, lds, a.n_rev, tid);
  float gr[KC * 4];
  {
    // G(r_{L-1}) = W_last^T G(a_last): three row vectors
    constexpr int NW = 3 * KC, TOT = rowvec_chunks(KC, 3), NS = TOT / SC;
#pragma unroll
    for (int i = 0; i < KC * 4; ++i) gr[i] = 0.f;
#pragma unroll
    for (int s = 0; s < NS; ++s) {
      const f32x4* cur = reinterpret_cast<const f32x4*>(ws.advance(tid)) + lane;
#pragma unroll
      for (int j = 0; j < SC; ++j) {
        const int c = s * SC + j;
        if (c < NW) {
          const int row = c / KC, kc = c % KC;
          const f32x4 w = cur[j * 64];
          gr[kc * 4 + 0] = fmaf(w.x, g3[row], gr[kc * 4 + 0]);
          gr[kc * 4 + 1] = fmaf(w.y, g3[row], gr[kc * 4 + 1]);
          gr[kc * 4 + 2] = fmaf(w.z, g3[row], gr[kc * 4 + 2]);
          gr[kc * 4 + 3] = fmaf(w.w, g3[row], gr[kc * 4 + 3]);
        }
      }
    }
  }
  // top mask (exposed, VALU only): G(a_{L-2}) = G(r_{L-1}) where r_{L-1} > 0
  constexpr bool PRE = KC >= SC;
  float ga[KC * 4];
  {
    const int l = a.L - 2;
    const float* rrow = a.rs + l * lstride + mc * H;
    float* grow = a.gar + l * lstride + m * H;
#pragma unroll
    for (int c = 0; c < KC; ++c) {
      const f32x4 rv = *reinterpret_cast<const f32x4*>(rrow + 8 * c + 4 * hi);
      f32x4 o;
#pragma unroll
      for (int t = 0; t < 4; ++t) { o[t] = rv[t] > 0.f ? gr[c * 4 + t] : 0.f; ga[c * 4 + t] = o[t]; }
      if (valid) *reinterpret_cast<f32x4*>(grow + 8 * c + 4 * hi) = o;
    }
  }
  f32x16 acc[NT];
  for (int l = a.L - 2; l >= 1; --l) {
    MaskEpi<PRE> e{a.rs + (l - 1) * lstride + mc * H, a.gar + (l - 1) * lstride + m * H, hi, valid};
    dense_op_epi<NT, KC, 0, 1, 0, MaskEpi<PRE>>(ws, ga, acc, e, tid);
    commit_tiles<NT>(acc, ga);
  }
  {
    f32x16 fa[FT];
    StoreEpi se{a.fbar + m * F, hi, valid};
    dense_op_epi<FT, KC, 0, 1, 0, StoreEpi>(ws, ga, fa, se, tid);
  }
}

template <int H, int F>
__global__ __launch_bounds__(256) void rgb_bwd_split_kernel(RgbBwdArgs a, int64_t m0) {
  constexpr int NT = H / 32, KC = H / 8, FT = F / 32;
  static_assert(NT == 8 && KC == SC && FT == 8, "split-K kernels are built for 256-wide layers");
  extern __shared__ __attribute__((aligned(16))) float lds[];
  float* xlds = lds + 2 * STAGE_FLOATS;
  const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6, hi = lane >> 5;
  const int64_t m = m0 + (int64_t)blockIdx.x * 32 + (lane & 31);
  const bool valid = m < a.M;
  const int64_t mc = valid ? m : a.M - 1;
  const int64_t lstride = a.Mp * H;
  float g3[3];
#pragma unroll
  for (int j = 0; j < 3; ++j) {
    const float c = a.rgb[mc * 3 + j];
    g3[j] = a.rgb_bar[mc * 3 + j] * c * (1.0f - c);
  }
  if (valid && hi == 0 && w == 0) *reinterpret_cast<f32x4*>(a.ga_last + m * 4) = f32x4{g3[0], g3[1], g3[2], 0.f};
  WStream ws;
  ws.begin(a.rev, lds, a.n_rev, tid);
  float gaq[32];
  {
    // this wave's quarter of G(r_{L-1}) = W_last^T G(a_last): row j of the last layer fills stage j
    float grq[32];
#pragma unroll
    for (int i = 0; i < 32; ++i) grq[i] = 0.f;
    constexpr int NS = rowvec_chunks(KC, 3) / SC;
#pragma unroll
    for (int s = 0; s < NS; ++s) {
      const f32x4* cur = reinterpret_cast<const f32x4*>(ws.advance(tid)) + lane;
      if (s < 3) {
#pragma unroll
        for (int i = 0; i < 8; ++i) {
          const f32x4 wv = cur[(8 * w + i) * 64];
          grq[i * 4 + 0] = fmaf(wv.x, g3[s < 3 ? s : 0], grq[i * 4 + 0]);
          grq[i * 4 + 1] = fmaf(wv.y, g3[s < 3 ? s : 0], grq[i * 4 + 1]);
          grq[i * 4 + 2] = fmaf(wv.z, g3[s < 3 ? s : 0], grq[i * 4 + 2]);
          grq[i * 4 + 3] = fmaf(wv.w, g3[s < 3 ? s : 0], grq[i * 4 + 3]);
        }
      }
    }
    const int l = a.L - 2;
    float rq[32];
    load_quarter(a.rs + l * lstride + mc * H, w, hi, rq);
#pragma unroll
    for (int i = 0; i < 32; ++i) gaq[i] = rq[i] > 0.f ? grq[i] : 0.f;
    store_quarter(a.gar + l * lstride + m * H, w, hi, valid, gaq);
  }
  for (int l = a.L - 2; l >= 1; --l) {
    MaskEpi<true> e{a.rs + (l - 1) * lstride + mc * H, a.gar + (l - 1) * lstride + m * H, hi, valid};
    float nq[32];
    dense_op_ksplit<NT, 0, 1, MaskEpi<true>>(ws, gaq, nq, nullptr, e, xlds, tid);
#pragma unroll
    for (int i = 0; i < 32; ++i) gaq[i] = nq[i];
  }
  {
    StoreEpi se{a.fbar + m * F, hi, valid};
    float dq[32];
    dense_op_ksplit<FT, 0, 1, StoreEpi>(ws, gaq, dq, nullptr, se, xlds, tid);
  }
}

}  // namespace

extern "C" int i2sdf_sdf_backward(const i2sdf_plan* p, const float* packed, const float* points, const float* cam, const float* dirs,
                                  const float* z, int64_t ldz, int32_t n_per_ray, int64_t n_ray_pts, int64_t M, int64_t Mp,
                                  const float* hs, const float* abars, const float* sbar, const float* fbar, int64_t m_fbar,
                                  const float* nbar, float* gus, float* gpbar, float* gas, float* ga_last4, float* ones4, void* stream) {
  if (M == 0) return I2SDF_OK;                 // empty batch: nothing to validate, nothing to launch
  if (!p || !packed || !hs || !abars || !gus || !gpbar || !gas || !ga_last4 || !ones4 || M < 0) return I2SDF_EINVAL;
  if (Mp < M || Mp % PTS_PER_WG) return I2SDF_EINVAL;
  if (n_ray_pts < 0 || n_ray_pts > M || (n_ray_pts > 0 && (!cam || !dirs || !z || n_per_ray <= 0)) || (n_ray_pts < M && !points))
    return I2SDF_EINVAL;
  const i2sdf_mlp_desc& d = p->sdf.d;
  if (d.multires != 6) return I2SDF_EINVAL;
  SdfBwdArgs a{};
  const float* base = packed + p->scale_floats;
  a.fwd = base + p->sdf.fwd_chunk0 * CHUNK_FLOATS;
  a.rev = base + p->sdf.rev_chunk0 * CHUNK_FLOATS;
  a.L = d.n_lin; a.skip = d.skip_layer;
  a.pts = PointSpec{points, cam, dirs, z, ldz, n_ray_pts, n_per_ray > 0 ? n_per_ray : 1};
  a.M = M; a.Mp = Mp; a.hs = hs; a.abars = abars; a.sbar = sbar; a.fbar = fbar; a.m_fbar = m_fbar; a.nbar = nbar;
  a.gus = gus; a.gpbar = gpbar; a.gas = gas; a.ga_last4 = ga_last4; a.ones4 = ones4;
  hipStream_t st = (hipStream_t)stream;
  const unsigned grid = (unsigned)((M + PTS_PER_WG - 1) / PTS_PER_WG);
  const bool has_skip = d.skip_layer > 0;
  if (p->H == 256 && p->F == 256) {
    a.n_fwd = sdf_fwd_hidden_stages(256, PE<6>::PEC, d.n_lin, has_skip);
    a.n_rev = sdf_rev_bwd_stages(256, 256, PE<6>::PEC, d.n_lin, has_skip);
    const int64_t bulk = split_bulk_points(M, p->n_cu);
    // full workgroups in bf16x3 split arithmetic (two launches); needs at least one plain hidden layer above the skip layer
    const bool x3 = p->sdf_bwd_bf16x3 != 0 && d.n_lin >= 4 && d.skip_layer != d.n_lin - 2;
    ChainGuard guard(p, st, x3 && i2sdf_parts_on(p));
    auto launch3 = [&](unsigned g) {
      SdfBwdArgs a3 = a;
      a3.fwd = base + p->sdf.fwd3_chunk0 * CHUNK_FLOATS;
      a3.rev = base + p->sdf.rev3_chunk0 * CHUNK_FLOATS;
      a3.n_fwd = sdf_fwd3_hidden_stages(256, PE<6>::DIM, d.n_lin, has_skip);
      a3.n_rev = sdf_rev3_bwd_stages(256, 256, PE<6>::PEC, d.n_lin, has_skip);
      a3.kcs = sdf_blocked_points(p, M, Mp) > 0 ? KCS_BLK : KCS_PM;
      a3.p24 = (sdf_saves24(p) && a3.kcs == KCS_BLK) ? 1 : 0;
      i2sdf_launch_sdf_bwd3(a3, g, st);
    };
    if (x3 && i2sdf_parts_on(p)) {      // point ranges (plan.h: PartRun): both sweeps of a range on the range's stream
      PartRun pr;
      i2sdf_parts_begin(p, st, M, &pr);
      for (int q = 0; q < pr.n; ++q) {
        if (pr.hi[q] <= pr.lo[q]) continue;
        st = pr.st[q];
        a.wg0 = (int)(pr.lo[q] / PTS_PER_WG);
        launch3((unsigned)((pr.hi[q] - pr.lo[q] + PTS_PER_WG - 1) / PTS_PER_WG));
      }
      st = (hipStream_t)stream;
      a.wg0 = 0;
      i2sdf_parts_end(p, st, &pr);
    } else if (bulk > 0) {          // full rounds + the partial last round as split-K workgroups (ksplit.h)
      hipStream_t ts = i2sdf_tail_fork(p, st);       // tail first, on the side stream when the overlap is on (plan.h)
      launch_lds_bytes(KS_LDS_BYTES, sdf_bwd_split_kernel<256, 256, 6>, (unsigned)((M - bulk + 31) / 32), ts, a, bulk);
      a.M = bulk;
      if (x3) launch3((unsigned)(bulk / PTS_PER_WG));
      else launch_lds(sdf_bwd_kernel<256, 256, 6>, (unsigned)(bulk / PTS_PER_WG), st, a);
      a.M = M;
      i2sdf_tail_join(p, st, ts);
    } else if (x3) {
      launch3(grid);
    } else {
      launch_lds(sdf_bwd_kernel<256, 256, 6>, grid, st, a);
    }
  } else if (p->H == 64 && p->F == 64) {
    a.n_fwd = sdf_fwd_hidden_stages(64, PE<6>::PEC, d.n_lin, has_skip);
    a.n_rev = sdf_rev_bwd_stages(64, 64, PE<6>::PEC, d.n_lin, has_skip);
    launch_lds(sdf_bwd_kernel<64, 64, 6>, grid, st, a);
  } else return I2SDF_EINVAL;
  return i2sdf_hip_check(hipGetLastError(), "sdf_backward launch");
}

extern "C" int i2sdf_rgb_backward(const i2sdf_plan* p, const float* packed, const float* rgb, const float* rgb_bar, const float* rs,
                                  int64_t M, int64_t Mp, float* gar, float* ga_last, float* fbar, void* stream) {
  if (M == 0) return I2SDF_OK;                 // empty batch: nothing to validate, nothing to launch
  if (!p || !packed || !rgb || !rgb_bar || !rs || !gar || !ga_last || !fbar || M < 0) return I2SDF_EINVAL;
  if (Mp < M || Mp % PTS_PER_WG) return I2SDF_EINVAL;
  const i2sdf_mlp_desc& d = p->rgb.d;
  RgbBwdArgs a{};
  a.rev = packed + p->scale_floats + p->rgb.rev_chunk0 * CHUNK_FLOATS;
  a.L = d.n_lin; a.M = M; a.Mp = Mp; a.rgb = rgb; a.rgb_bar = rgb_bar; a.rs = rs; a.gar = gar; a.ga_last = ga_last; a.fbar = fbar;
  hipStream_t st = (hipStream_t)stream;
  const unsigned grid = (unsigned)((M + PTS_PER_WG - 1) / PTS_PER_WG);
  ChainGuard guard(p, st, d.hidden == 256 && p->F == 256 && p->rgb_bf16x3 && i2sdf_parts_on(p));
  if (d.hidden == 256 && p->F == 256) {
    a.n_rev = rgb_rev_stages(256, 256, d.n_lin);
    const int64_t bulk = split_bulk_points(M, p->n_cu);
    auto full = [&](const RgbBwdArgs& x, unsigned g) {
      if (p->rgb_bf16x3) {
        RgbBwdArgs x3 = x;
        x3.rev = packed + p->scale_floats + p->rgb.rev3h_chunk0 * CHUNK_FLOATS;          // 16-point waves (x3h.h)
        x3.n_rev = rgb_rev3h_stages(256, 256, d.n_lin);
        x3.kcs = rgb_blocked_points(p, M, Mp) > 0 ? KCS_BLK : KCS_PM;
        i2sdf_launch_rgb_bwd3h(x3, g, st);
      } else {
        launch_lds(rgb_bwd_kernel<256, 256>, g, st, x);
      }
    };
    if (p->rgb_bf16x3 && i2sdf_parts_on(p)) {      // point ranges (plan.h: PartRun)
      PartRun pr;
      i2sdf_parts_begin(p, st, M, &pr);
      for (int q = 0; q < pr.n; ++q) {
        if (pr.hi[q] <= pr.lo[q]) continue;
        st = pr.st[q];
        RgbBwdArgs b = a;
        b.wg0 = (int)(pr.lo[q] / PTS_PER_WG);
        full(b, (unsigned)((pr.hi[q] - pr.lo[q] + PTS_PER_WG - 1) / PTS_PER_WG));
      }
      st = (hipStream_t)stream;
      i2sdf_parts_end(p, st, &pr);
    } else if (bulk > 0) {
      a.M = bulk;
      full(a, (unsigned)(bulk / PTS_PER_WG));
      a.M = M;
      launch_lds_bytes(KS_LDS_BYTES, rgb_bwd_split_kernel<256, 256>, (unsigned)((M - bulk + 31) / 32), st, a, bulk);
    } else {
      full(a, grid);
    }
  } else if (d.hidden == 64 && p->F == 64) {
    a.n_rev = rgb_rev_stages(64, 64, d.n_lin);
    launch_lds(rgb_bwd_kernel<64, 64>, grid, st, a);
  } else return I2SDF_EINVAL;
  return i2sdf_hip_check(hipGetLastError(), "rgb_backward launch");
}
