// Training/eval-mode forward kernels:
//   sdf_train_fwd : ImplicitNetwork.get_outputs (model/network/mlp.py:123-143) = forward + d sdf/dx, where the
//                   reference builds d sdf/dx with torch.autograd.grad(create_graph=True); here the reverse chain
//                    is evaluated explicitly in the same kernel, in registers.
//                   Saves h_l = softplus(a_{l-1}) and abar_l = d sdf / d a_l for the backward kernels.
//   rgb_fwd       : RenderingNetwork.forward, 'nerf' mode (mlp.py:208-229), saves the post-ReLU activations.
#include "ksplit.h"
#include "mlp_args.h"

using namespace i2sdf;

int i2sdf_hip_check(hipError_t e, const char* what);


namespace {

template <int H, int F, int LF, bool GRAD>
__global__ __launch_bounds__(256) void sdf_train_fwd_kernel(SdfTrainFwdArgs a) {
  constexpr int NT = H / 32, KC = H / 8, PEC = PE<LF>::PEC, PT = cdiv(PEC * 8, 32), FT = F / 32;
  extern __shared__ __attribute__((aligned(16))) float lds[];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, hi = lane >> 5;
  const int64_t m = ((int64_t)(blockIdx.x + a.wg0) * 4 + wave) * 32 + (lane & 31);
  const bool valid = m < a.M;
  const int64_t mc = valid ? m : a.M - 1;
  float px, py, pz;
  fetch_point(a.pts, mc, px, py, pz);
  float pe[PEC * 4];
  {
    float full[PEC * 8];
    pe_full<LF>(px, py, pz, full);
    to_b_layout<PEC>(full, pe, hi);
  }
  if (a.pe_save) store_regs<PEC>(a.pe_save + m * (PEC * 8), hi, valid, pe);
  const int64_t lstride = a.Mp * H;
  constexpr bool PRE = KC >= SC;          // at most one tile completes per LDS stage -> prefetching epilogues are legal
  WStream ws;
  ws.begin(a.fwd, lds, a.n_fwd, tid);
  f32x16 acc[NT];
  float h[NT * 16];
  SoftplusEpi fe{a.hs ? a.hs + m * H : nullptr, hi, valid};
  dense_op_epi<NT, PEC, NT * 4, 0, 0, SoftplusEpi>(ws, pe, acc, fe, tid);
  commit_tiles<NT>(acc, h);
  for (int l = 1; l < a.L - 1; ++l) {
    fe.row = a.hs ? a.hs + l * lstride + m * H : nullptr;
    if (l == a.skip) {
      float u[(KC + PEC) * 4];
#pragma unroll
      for (int i = 0; i < KC * 4; ++i) u[i] = h[i] * RS2;
#pragma unroll
      for (int i = 0; i < PEC * 4; ++i) u[KC * 4 + i] = pe[i] * RS2;
      dense_op_epi<NT, KC + PEC, NT * 4, 0, 0, SoftplusEpi>(ws, u, acc, fe, tid);
    } else {
      dense_op_epi<NT, KC, NT * 4, 0, 0, SoftplusEpi>(ws, h, acc, fe, tid);
    }
    commit_tiles<NT>(acc, h);
  }
  {
    float s[1];
    rowvec_op<1, KC>(ws, h, s, tid);
    if (valid && hi == 0) a.sdf[m] = s[0];
  }
  if (a.feat != nullptr) {
    f32x16 fa[FT];
    StoreEpi se{a.feat + m * F, hi, valid};
    dense_op_epi<FT, KC, FT * 4, 0, 0, StoreEpi>(ws, h, fa, se, tid);
  }
  if (!GRAD) return;
  // ---------------- reverse chain: n = d sdf / d x  (appendix A.2) ----------------
  __syncthreads();                      // every wave is done with the forward stream's LDS buffers
  ws.begin(a.rev, lds, a.n_rev, tid);
  float ab[KC * 4];
  {
    float wv[KC * 4];
    f32x4 sc;
    rowvec_load<KC>(ws, wv, sc, tid);
#pragma unroll
    for (int i = 0; i < KC * 4; ++i) ab[i] = wv[i] * sp_sigma_from_h(h[i]);     // abar_{L-2} = w_sdf (.) sigma_{L-2}
  }
  if (a.abars) store_regs<KC>(a.abars + (a.L - 2) * lstride + m * H, hi, valid, ab);
  f32x16 pt[PT];
#pragma unroll
  for (int i = 0; i < PT; ++i)
#pragma unroll
    for (int r = 0; r < 16; ++r) pt[i][r] = 0.f;
  for (int l = a.L - 2; l >= 1; --l) {
    const float* hrow = a.hs + (l - 1) * lstride + mc * H;
    float* abrow = a.abars ? a.abars + (l - 1) * lstride + m * H : nullptr;
    if (l == a.skip) {
      f32x16 as[NT + PT];
      RevEpi<NT, PT, PRE> re{hrow, abrow, hi, valid, RS2, pt};
      dense_op_epi<NT + PT, KC, 0, 1, 0, RevEpi<NT, PT, PRE>>(ws, ab, as, re, tid);
#pragma unroll
      for (int i = 0; i < KC * 4; ++i) ab[i] = as[i / 16][i % 16];
    } else {
      f32x16 a2[NT];
      RevEpi<NT, PT, PRE> re{hrow, abrow, hi, valid, 1.0f, pt};
      dense_op_epi<NT, KC, 0, 1, 0, RevEpi<NT, PT, PRE>>(ws, ab, a2, re, tid);
      commit_tiles<NT>(a2, ab);
    }
  }
  dense_op_nobias<PT, KC, 2>(ws, ab, pt, tid);       // pbar += W_0^T abar_0
  {
    float full[PEC * 8], coef[PEC * 8], n[3];
    pe_full<LF>(px, py, pz, full);
    pe_coef<LF>(full, coef);
    pe_jt_apply<LF, PT>(coef, pt, hi, n);
    if (valid && hi == 0) { a.grad[m * 3 + 0] = n[0]; a.grad[m * 3 + 1] = n[1]; a.grad[m * 3 + 2] = n[2]; }
  }
}

// Split-K variant of sdf_train_fwd_kernel for the last partial round (ksplit.h): one 32-point tile per workgroup.
template <int H, int F, int LF, bool GRAD>
__global__ __launch_bounds__(256) void sdf_train_fwd_split_kernel(SdfTrainFwdArgs a, int64_t m0) {
  constexpr int NT = H / 32, KC = H / 8, PEC = PE<LF>::PEC, PT = cdiv(PEC * 8, 32), FT = F / 32;
  static_assert(NT == 8 && KC == SC && FT == 8, "split-K kernels are built for 256-wide layers");
  extern __shared__ __attribute__((aligned(16))) float lds[];
  float* xlds = lds + 2 * STAGE_FLOATS;
  float* slds = xlds + KS_X_FLOATS;
  float* glds = slds + KS_S_FLOATS;
  const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6, hi = lane >> 5;
  const int64_t m = m0 + (int64_t)blockIdx.x * 32 + (lane & 31);
  const bool valid = m < a.M;
  const int64_t mc = valid ? m : a.M - 1;
  float px, py, pz;
  fetch_point(a.pts, mc, px, py, pz);
  float pe[PEC * 4];
  {
    float full[PEC * 8];
    pe_full<LF>(px, py, pz, full);
    to_b_layout<PEC>(full, pe, hi);
  }
  if (a.pe_save && w == 0) store_regs<PEC>(a.pe_save + m * (PEC * 8), hi, valid, pe);
  const int64_t lstride = a.Mp * H;
  WStream ws;
  ws.begin(a.fwd, lds, a.n_fwd, tid);
  float hq[32];
  {
    f32x16 acc[NT];
    float h[NT * 16];
    SoftplusEpi fe{a.hs ? a.hs + m * H : nullptr, hi, valid};
    fe.own = w;
    dense_op_epi<NT, PEC, NT * 4, 0, 0, SoftplusEpi>(ws, pe, acc, fe, tid);         // tiny: every wave computes all of layer 0
    commit_tiles<NT>(acc, h);
    take_quarter(h, hq, w);
  }
  for (int l = 1; l < a.L - 1; ++l) {
    float* hrow = a.hs ? a.hs + l * lstride + m * H : nullptr;
    if (l == a.skip) {
      // 37 chunks per tile: not stage aligned -> gather the full input, every wave computes the layer, keeps its quarter
      float u[(KC + PEC) * 4];
      {
        float h[NT * 16];
        gather_full(hq, h, glds, tid);
#pragma unroll
        for (int i = 0; i < KC * 4; ++i) u[i] = h[i] * RS2;
      }
#pragma unroll
      for (int i = 0; i < PEC * 4; ++i) u[KC * 4 + i] = pe[i] * RS2;
      f32x16 acc[NT];
      SoftplusEpi fe{hrow, hi, valid};
      fe.own = w;
      dense_op_epi<NT, KC + PEC, NT * 4, 0, 0, SoftplusEpi>(ws, u, acc, fe, tid);
      float h2[NT * 16];
      commit_tiles<NT>(acc, h2);
      take_quarter(h2, hq, w);
    } else {
      SoftplusEpi fe{hrow, hi, valid};
      float nq[32];
      dense_op_ksplit<NT, NT * 4, 0, SoftplusEpi>(ws, hq, nq, nullptr, fe, xlds, tid);
#pragma unroll
      for (int i = 0; i < 32; ++i) hq[i] = nq[i];
    }
  }
  {
    float s[1];
    rowvec_ksplit<1>(ws, hq, s, slds, tid);
    if (valid && hi == 0 && w == 0) a.sdf[m] = s[0];
  }
  if (a.feat != nullptr) {
    StoreEpi se{a.feat + m * F, hi, valid};
    float dq[32];
    dense_op_ksplit<FT, FT * 4, 0, StoreEpi>(ws, hq, dq, nullptr, se, xlds, tid);
  }
  if (!GRAD) return;
  // ---------------- reverse chain ----------------
  __syncthreads();
  ws.begin(a.rev, lds, a.n_rev, tid);
  float abq[32];
  {
    float wq[32];
    f32x4 sc;
    rowvec_load_quarter(ws, wq, sc, tid);
#pragma unroll
    for (int i = 0; i < 32; ++i) abq[i] = wq[i] * sp_sigma_from_h(hq[i]);
  }
  if (a.abars) store_quarter(a.abars + (a.L - 2) * lstride + m * H, w, hi, valid, abq);
  f32x16 pt[PT];
#pragma unroll
  for (int i = 0; i < PT; ++i)
#pragma unroll
    for (int r = 0; r < 16; ++r) pt[i][r] = 0.f;
  for (int l = a.L - 2; l >= 1; --l) {
    const float* hrow = a.hs + (l - 1) * lstride + mc * H;
    float* abrow = a.abars ? a.abars + (l - 1) * lstride + m * H : nullptr;
    float nq[32];
    if (l == a.skip) {
      RevEpi<NT, PT, true> re{hrow, abrow, hi, valid, RS2, pt};
      dense_op_ksplit<NT + PT, 0, 1, RevEpi<NT, PT, true>>(ws, abq, nq, nullptr, re, xlds, tid);
    } else {
      RevEpi<NT, PT, true> re{hrow, abrow, hi, valid, 1.0f, pt};
      dense_op_ksplit<NT, 0, 1, RevEpi<NT, PT, true>>(ws, abq, nq, nullptr, re, xlds, tid);
    }
#pragma unroll
    for (int i = 0; i < 32; ++i) abq[i] = nq[i];
  }
  {
    NoEpi ne;
    float oq[32];
#pragma unroll
    for (int i = 0; i < 32; ++i) oq[i] = 0.f;
    dense_op_ksplit<PT, 0, 1, NoEpi>(ws, abq, oq, nullptr, ne, xlds, tid);        // W_0^T abar_0: both tiles land in wave 0
    if (w == 0) {
#pragma unroll
      for (int i = 0; i < PT; ++i)
#pragma unroll
        for (int r = 0; r < 16; ++r) pt[i][r] += oq[i * 16 + r];
      float full[PEC * 8], coef[PEC * 8], n[3];
      pe_full<LF>(px, py, pz, full);
      pe_coef<LF>(full, coef);
      pe_jt_apply<LF, PT>(coef, pt, hi, n);
      if (valid && hi == 0) { a.grad[m * 3 + 0] = n[0]; a.grad[m * 3 + 1] = n[1]; a.grad[m * 3 + 2] = n[2]; }
    }
  }
}

}  // namespace

// ---- radiance network ---------------------------------------------------------------------------------------

namespace {

__host__ __device__ constexpr int rgb_fwd_stages(int H, int F, int PECV, int L) {
  int c = op_chunks(H / 32, PECV + F / 8);
  for (int l = 1; l < L - 1; ++l) c += op_chunks(H / 32, H / 8);
  c += rowvec_chunks(H / 8, 3);
  return c / SC;
}

template <int H, int F, int LFV>
__global__ __launch_bounds__(256) void rgb_fwd_kernel(RgbFwdArgs a) {
  constexpr int NT = H / 32, KC = H / 8, PECV = PE<LFV>::PEC, FC = F / 8;
  extern __shared__ __attribute__((aligned(16))) float lds[];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, hi = lane >> 5;
  const int64_t m = ((int64_t)(blockIdx.x + a.wg0) * 4 + wave) * 32 + (lane & 31);
  const bool valid = m < a.M;
  const int64_t mc = valid ? m : a.M - 1;
  const int64_t ray = mc / a.n_per_ray;
  float in0[(PECV + FC) * 4];
  {
    float full[PECV * 8], pv[PECV * 4];
    pe_full<LFV>(a.dirs[ray * 3 + 0], a.dirs[ray * 3 + 1], a.dirs[ray * 3 + 2], full);
    to_b_layout<PECV>(full, pv, hi);
    if (a.pev_save) store_regs<PECV>(a.pev_save + m * (PECV * 8), hi, valid, pv);
    float ft[FC * 4];
    load_regs<FC>(a.feat + mc * F, hi, ft);
#pragma unroll
    for (int i = 0; i < PECV * 4; ++i) in0[i] = pv[i];
#pragma unroll
    for (int i = 0; i < FC * 4; ++i) in0[PECV * 4 + i] = ft[i];
  }
  const int64_t lstride = a.Mp * H;
  WStream ws;
  ws.begin(a.fwd, lds, a.n_fwd, tid);
  f32x16 acc[NT];
  float r[NT * 16];
  ReluEpi re{a.rs ? a.rs + m * H : nullptr, hi, valid};
  dense_op_epi<NT, PECV + FC, NT * 4, 0, 0, ReluEpi>(ws, in0, acc, re, tid);
  commit_tiles<NT>(acc, r);
  for (int l = 1; l < a.L - 1; ++l) {
    re.row = a.rs ? a.rs + l * lstride + m * H : nullptr;
    dense_op_epi<NT, KC, NT * 4, 0, 0, ReluEpi>(ws, r, acc, re, tid);
    commit_tiles<NT>(acc, r);
  }
  float o[3];
  rowvec_op<3, KC>(ws, r, o, tid);
  if (valid && hi == 0) {
#pragma unroll
    for (int i = 0; i < 3; ++i) a.rgb[m * 3 + i] = 1.0f / (1.0f + expf(-o[i]));
  }
}

// Split-K variant for the last partial round (ksplit.h): one 32-point tile per workgroup, points m0 + 32*blockIdx.x ...
template <int H, int F, int LFV>
__global__ __launch_bounds__(256) void rgb_fwd_split_kernel(RgbFwdArgs a, int64_t m0) {
  constexpr int NT = H / 32, KC = H / 8, PECV = PE<LFV>::PEC, FC = F / 8;
  static_assert(NT == 8 && KC == SC && FC == SC, "split-K kernels are built for 256-wide layers");
  extern __shared__ __attribute__((aligned(16))) float lds[];
  float* xlds = lds + 2 * STAGE_FLOATS;
  float* slds = xlds + KS_X_FLOATS;
  const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6, hi = lane >> 5;
  const int64_t m = m0 + (int64_t)blockIdx.x * 32 + (lane & 31);
  const bool valid = m < a.M;
  const int64_t mc = valid ? m : a.M - 1;
  const int64_t ray = mc / a.n_per_ray;
  float in0[(PECV + FC) * 4];
  {
    float full[PECV * 8], pv[PECV * 4];
    pe_full<LFV>(a.dirs[ray * 3 + 0], a.dirs[ray * 3 + 1], a.dirs[ray * 3 + 2], full);
    to_b_layout<PECV>(full, pv, hi);
    if (a.pev_save && w == 0) store_regs<PECV>(a.pev_save + m * (PECV * 8), hi, valid, pv);
    float ft[FC * 4];
    load_regs<FC>(a.feat + mc * F, hi, ft);
#pragma unroll
    for (int i = 0; i < PECV * 4; ++i) in0[i] = pv[i];
#pragma unroll
    for (int i = 0; i < FC * 4; ++i) in0[PECV * 4 + i] = ft[i];
  }
  const int64_t lstride = a.Mp * H;
  WStream ws;
  ws.begin(a.fwd, lds, a.n_fwd, tid);
  float rq[32];
  {
    // layer 0 (reduction length 36 chunks: not stage aligned) is computed by every wave; each keeps its quarter
    f32x16 acc[NT];
    float r[NT * 16];
    ReluEpi re{a.rs ? a.rs + m * H : nullptr, hi, valid};
    re.own = w;
    dense_op_epi<NT, PECV + FC, NT * 4, 0, 0, ReluEpi>(ws, in0, acc, re, tid);
    commit_tiles<NT>(acc, r);
    take_quarter(r, rq, w);
  }
  for (int l = 1; l < a.L - 1; ++l) {
    ReluEpi re{a.rs ? a.rs + l * lstride + m * H : nullptr, hi, valid};
    float nq[32];
    dense_op_ksplit<NT, NT * 4, 0, ReluEpi>(ws, rq, nq, nullptr, re, xlds, tid);
#pragma unroll
    for (int i = 0; i < 32; ++i) rq[i] = nq[i];
  }
  float o[3];
  rowvec_ksplit<3>(ws, rq, o, slds, tid);
  if (valid && hi == 0 && w == 0) {
#pragma unroll
    for (int i = 0; i < 3; ++i) a.rgb[m * 3 + i] = 1.0f / (1.0f + expf(-o[i]));
  }
}

}  // namespace

// leading points of a batch whose saved 256-wide tensors are in the blocked layout (I2SDF_OPT_BLOCKED_SAVES): which = 0 the SDF
// tensors (hs, abars, gus, gas) of a batch of M points, 1 the radiance tensors (rs, gar); which = 2: the leading points whose abars / gus / gas
// are packed 24-bit records (I2SDF_OPT_SAVES24: all of them, Mp, or none)
extern "C" int64_t i2sdf_blocked_points(const i2sdf_plan* p, int32_t which, int64_t M, int64_t Mp, int32_t has_feat) {
  if (!p || M <= 0) return 0;
  if (which == 2) return (sdf_saves24(p) && sdf_blocked_points(p, M, Mp, has_feat != 0) == Mp) ? Mp : 0;
  return which == 0 ? sdf_blocked_points(p, M, Mp, has_feat != 0) : rgb_blocked_points(p, M, Mp);
}

// =============================================================================================================
extern "C" int i2sdf_sdf_forward_grad(const i2sdf_plan* p, const float* packed, const float* points, const float* cam,
                                      const float* dirs, const float* z, int64_t ldz, int32_t n_per_ray, int64_t n_ray_pts, int64_t M,
                                      int64_t Mp, float* sdf, float* feat, float* grad, float* hs, float* abars, float* pe_save,
                                      void* stream) {
  if (M == 0) return I2SDF_OK;                 // empty batch: nothing to validate, nothing to launch
  if (!p || !packed || M < 0 || !sdf) return I2SDF_EINVAL;
  if (Mp < M || Mp % PTS_PER_WG) return I2SDF_EINVAL;
  if (n_ray_pts < 0 || n_ray_pts > M) return I2SDF_EINVAL;
  if (n_ray_pts > 0 && (!cam || !dirs || !z || n_per_ray <= 0)) return I2SDF_EINVAL;
  if (n_ray_pts < M && !points) return I2SDF_EINVAL;
  if (grad && !hs) return I2SDF_EINVAL;          // the reverse chain re-reads h_l
  const i2sdf_mlp_desc& d = p->sdf.d;
  if (d.multires != 6) return I2SDF_EINVAL;
  SdfTrainFwdArgs a{};
  const float* base = packed + p->scale_floats;
  a.fwd = base + p->sdf.fwd_chunk0 * CHUNK_FLOATS;
  a.rev = base + p->sdf.rev_wsdf_chunk * CHUNK_FLOATS;
  a.L = d.n_lin; a.skip = d.skip_layer;
  a.pts = PointSpec{points, cam, dirs, z, ldz, n_ray_pts, n_per_ray > 0 ? n_per_ray : 1};
  a.M = M; a.Mp = Mp; a.sdf = sdf; a.feat = feat; a.grad = grad; a.hs = hs; a.abars = abars; a.pe_save = pe_save;
  hipStream_t st = (hipStream_t)stream;
  const unsigned grid = (unsigned)((M + PTS_PER_WG - 1) / PTS_PER_WG);
  const bool has_skip = d.skip_layer > 0;
#define LAUNCH(HH, FF, G_)                                                                           \
  do {                                                                                               \
    a.n_fwd = sdf_fwd_stages(HH, FF, PE<6>::PEC, d.n_lin, has_skip, feat != nullptr);                \
    a.n_rev = sdf_rev_stages(HH, PE<6>::PEC, d.n_lin, has_skip);                                     \
    if (grad) launch_lds(sdf_train_fwd_kernel<HH, FF, 6, true>, G_, st, a);                          \
    else launch_lds(sdf_train_fwd_kernel<HH, FF, 6, false>, G_, st, a);                              \
  } while (0)
  // full workgroups in bf16x3 split arithmetic: the forward with saves on 16-point waves (x3h.h, mlp_x3h.hip), the d sdf/dx chain on
  // 32-point waves (x3.h, mlp_x3.hip) -- both read / write the same saved tensors; the split-K tail keeps the fp32 MFMA kernel
#define LAUNCH3(G_)                                                                                  \
  do {                                                                                               \
    SdfTrainFwdArgs a3 = a;                                                                          \
    a3.fwd = base + p->sdf.fwd3h_chunk0 * CHUNK_FLOATS;                                              \
    a3.rev = base + p->sdf.rev3_wsdf_chunk * CHUNK_FLOATS;                                           \
    a3.n_fwd = sdf_fwd3h_train_stages(256, 256, PE<6>::DIM, d.n_lin, has_skip, feat != nullptr);     \
    a3.n_rev = sdf_rev3_stages(256, PE<6>::PEC, d.n_lin, has_skip);                                  \
    a3.kcs = sdf_blocked_points(p, M, Mp, feat != nullptr) > 0 ? KCS_BLK : KCS_PM;                   \
    a3.p24 = (sdf_saves24(p) && a3.kcs == KCS_BLK && abars != nullptr) ? 1 : 0;                      \
    i2sdf_launch_train_fwd3h(a3, G_, st);                                                            \
    if (grad) i2sdf_launch_igrad3(a3, G_, st);                                                       \
  } while (0)
  const bool x3 = p->train_fwd_bf16x3 != 0 && p->H == 256 && p->F == 256 && d.n_lin >= 4 && d.skip_layer != d.n_lin - 2;
  ChainGuard guard(p, st, x3 && i2sdf_parts_on(p));
  if (x3 && i2sdf_parts_on(p)) {
    // point ranges (plan.h: PartRun): one launch pair per range, each on the range's own stream; no split-K tail
    PartRun pr;
    i2sdf_parts_begin(p, st, M, &pr);
    for (int q = 0; q < pr.n; ++q) {
      if (pr.hi[q] <= pr.lo[q]) continue;
      st = pr.st[q];
      a.wg0 = (int)(pr.lo[q] / PTS_PER_WG);
      LAUNCH3((unsigned)((pr.hi[q] - pr.lo[q] + PTS_PER_WG - 1) / PTS_PER_WG));
    }
    st = (hipStream_t)stream;
    a.wg0 = 0;
    i2sdf_parts_end(p, st, &pr);
  } else if (p->H == 256 && p->F == 256) {
    const int64_t bulk = split_bulk_points(M, p->n_cu);
    if (bulk > 0 && feat != nullptr) {      // full rounds + the partial last round as split-K workgroups (ksplit.h)
      const int64_t M_all = a.M;
      const unsigned tg = (unsigned)((M_all - bulk + 31) / 32);
      {                                      // tail first, on the side stream when the overlap is on (plan.h)
        auto t = a;
        t.n_fwd = sdf_fwd_stages(256, 256, PE<6>::PEC, d.n_lin, has_skip, feat != nullptr);
        t.n_rev = sdf_rev_stages(256, PE<6>::PEC, d.n_lin, has_skip);
        hipStream_t ts = i2sdf_tail_fork(p, st);
        if (grad) launch_lds_bytes(KS_LDS_BYTES, sdf_train_fwd_split_kernel<256, 256, 6, true>, tg, ts, t, bulk);
        else launch_lds_bytes(KS_LDS_BYTES, sdf_train_fwd_split_kernel<256, 256, 6, false>, tg, ts, t, bulk);
        a.M = bulk;
        if (x3) LAUNCH3((unsigned)(bulk / PTS_PER_WG));
        else LAUNCH(256, 256, (unsigned)(bulk / PTS_PER_WG));
        a.M = M_all;
        i2sdf_tail_join(p, st, ts);
      }
    } else if (x3) {
      LAUNCH3(grid);
    } else {
      LAUNCH(256, 256, grid);
    }
  } else if (p->H == 64 && p->F == 64) LAUNCH(64, 64, grid);
  else return I2SDF_EINVAL;
#undef LAUNCH
#undef LAUNCH3
  return i2sdf_hip_check(hipGetLastError(), "sdf_forward_grad launch");
}

extern "C" int i2sdf_rgb_forward(const i2sdf_plan* p, const float* packed, const float* dirs, int32_t n_per_ray, const float* feat,
                                 int64_t M, int64_t Mp, float* rgb, float* rs, float* pev_save, void* stream) {
  if (M == 0) return I2SDF_OK;                 // empty batch: nothing to validate, nothing to launch
  if (!p || !packed || !dirs || !feat || !rgb || M < 0 || n_per_ray <= 0) return I2SDF_EINVAL;
  if (Mp < M || Mp % PTS_PER_WG) return I2SDF_EINVAL;
  const i2sdf_mlp_desc& d = p->rgb.d;
  if (d.multires != 4) return I2SDF_EINVAL;
  RgbFwdArgs a{};
  a.fwd = packed + p->scale_floats + p->rgb.fwd_chunk0 * CHUNK_FLOATS;
  a.L = d.n_lin; a.dirs = dirs; a.n_per_ray = n_per_ray; a.feat = feat; a.M = M; a.Mp = Mp; a.rgb = rgb; a.rs = rs; a.pev_save = pev_save;
  hipStream_t st = (hipStream_t)stream;
  const unsigned grid = (unsigned)((M + PTS_PER_WG - 1) / PTS_PER_WG);
  ChainGuard guard(p, st, d.hidden == 256 && p->F == 256 && p->rgb_bf16x3 && i2sdf_parts_on(p));
  if (d.hidden == 256 && p->F == 256) {
    a.n_fwd = rgb_fwd_stages(256, 256, PE<4>::PEC, d.n_lin);
    const int64_t bulk = split_bulk_points(M, p->n_cu);
    // full workgroups optionally in bf16x3 split arithmetic (mlp_x3.hip); the split-K tail keeps the fp32 MFMA kernel
    auto full = [&](const RgbFwdArgs& x, unsigned g) {
      if (p->rgb_bf16x3) {
        RgbFwdArgs x3 = x;
        x3.kcs = rgb_blocked_points(p, M, Mp) > 0 ? KCS_BLK : KCS_PM;
        x3.fwd = packed + p->scale_floats + p->rgb.fwd3h_chunk0 * CHUNK_FLOATS;
        x3.n_fwd = rgb_fwd3h_stages(256, 256, PE<4>::DIM, d.n_lin);
        i2sdf_launch_rgb_fwd3h(x3, g, st);
      } else {
        launch_lds(rgb_fwd_kernel<256, 256, 4>, g, st, x);
      }
    };
    if (p->rgb_bf16x3 && i2sdf_parts_on(p)) {      // point ranges (plan.h: PartRun)
      PartRun pr;
      i2sdf_parts_begin(p, st, M, &pr);
      for (int q = 0; q < pr.n; ++q) {
        if (pr.hi[q] <= pr.lo[q]) continue;
        st = pr.st[q];
        RgbFwdArgs b = a;
        b.wg0 = (int)(pr.lo[q] / PTS_PER_WG);
        full(b, (unsigned)((pr.hi[q] - pr.lo[q] + PTS_PER_WG - 1) / PTS_PER_WG));
      }
      st = (hipStream_t)stream;
      i2sdf_parts_end(p, st, &pr);
    } else if (bulk > 0) {          // full rounds with one 32-point tile per wave, the partial last round as split-K workgroups
      RgbFwdArgs b = a;
      b.M = bulk;
      full(b, (unsigned)(bulk / PTS_PER_WG));
      // (no side stream here: this tail is 32 short workgroups, the fork/join costs more than the overlap gains -- measured)
      launch_lds_bytes(KS_LDS_BYTES, rgb_fwd_split_kernel<256, 256, 4>, (unsigned)((M - bulk + 31) / 32), st, a, bulk);
    } else {
      full(a, grid);
    }
  } else if (d.hidden == 64 && p->F == 64) {
    a.n_fwd = rgb_fwd_stages(64, 64, PE<4>::PEC, d.n_lin);
    launch_lds(rgb_fwd_kernel<64, 64, 4>, grid, st, a);
  } else return I2SDF_EINVAL;
  return i2sdf_hip_check(hipGetLastError(), "rgb_forward launch");
}

// =============================================================================================================
// light-mask head -- model/network/__init__.py:29-32,162-170: lm = sigmoid(W1 softplus100(W0 relu(feature).detach() + b0) + b1)
// (an ImplicitNetwork without encoding, geometric_init False, sigmoid output).  The feature input is detached
// (detach_light_feature=True, the default), so the backward stops at the head's own parameters.
struct LightArgs {
  const float* fwd; int n_fwd; const float* rev; int n_rev;
  const float* feat; int64_t M, Mp;
  float* lm;            // (M)
  float* hl;            // (Mp, HL) softplus activations
  const float* lm_bar;  // (M)   backward only
  float* gal0;          // (Mp, HL)  G(a_0)
  float* gal_last;      // (Mp, 4)   {G(a_1),0,0,0}
};

namespace {

template <int HL, int F>
__global__ __launch_bounds__(256) void light_fwd_kernel(LightArgs a) {
  constexpr int NT = HL / 32, KC = HL / 8, FC = F / 8;
  extern __shared__ __attribute__((aligned(16))) float lds[];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, hi = lane >> 5;
  const int64_t m = ((int64_t)blockIdx.x * 4 + wave) * 32 + (lane & 31);
  const bool valid = m < a.M;
  const int64_t mc = valid ? m : a.M - 1;
  float in0[FC * 4];
  load_regs<FC>(a.feat + mc * F, hi, in0);
#pragma unroll
  for (int i = 0; i < FC * 4; ++i) in0[i] = fmaxf(in0[i], 0.f);
  WStream ws;
  ws.begin(a.fwd, lds, a.n_fwd, tid);
  f32x16 acc[NT];
  float h[NT * 16];
  SoftplusEpi le{a.hl ? a.hl + m * HL : nullptr, hi, valid};
  dense_op_epi<NT, FC, NT * 4, 0, 0, SoftplusEpi>(ws, in0, acc, le, tid);
  commit_tiles<NT>(acc, h);
  float o[1];
  rowvec_op<1, KC>(ws, h, o, tid);
  if (valid && hi == 0) a.lm[m] = 1.0f / (1.0f + expf(-o[0]));
}

template <int HL>
__global__ __launch_bounds__(256) void light_bwd_kernel(LightArgs a) {
  constexpr int KC = HL / 8;
  extern __shared__ __attribute__((aligned(16))) float lds[];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, hi = lane >> 5;
  const int64_t m = ((int64_t)blockIdx.x * 4 + wave) * 32 + (lane & 31);
  const bool valid = m < a.M;
  const int64_t mc = valid ? m : a.M - 1;
  const float l = a.lm[mc];
  const float g1 = a.lm_bar[mc] * l * (1.0f - l);
  if (valid && hi == 0) *reinterpret_cast<f32x4*>(a.gal_last + m * 4) = f32x4{g1, 0.f, 0.f, 0.f};
  WStream ws;
  ws.begin(a.rev, lds, a.n_rev, tid);
  float wv[KC * 4], hv[KC * 4];
  f32x4 sc;
  rowvec_load<KC>(ws, wv, sc, tid);
  load_regs<KC>(a.hl + mc * HL, hi, hv);
#pragma unroll
  for (int i = 0; i < KC * 4; ++i) wv[i] = g1 * wv[i] * sp_sigma_from_h(hv[i]);
  store_regs<KC>(a.gal0 + m * HL, hi, valid, wv);
}

}  // namespace

extern "C" int i2sdf_light_forward(const i2sdf_plan* p, const float* packed, const float* feat, int64_t M, int64_t Mp, float* lm, float* hl,
                                   void* stream) {
  if (M == 0) return I2SDF_OK;                 // empty batch: nothing to validate, nothing to launch
  if (!p || !packed || !feat || !lm || M < 0 || p->light.d.n_lin == 0) return I2SDF_EINVAL;
  if (Mp < M || Mp % PTS_PER_WG) return I2SDF_EINVAL;
  LightArgs a{};
  const int HL = p->light.d.hidden, F = p->F;
  a.fwd = packed + p->scale_floats + p->light.fwd_chunk0 * CHUNK_FLOATS;
  a.n_fwd = (op_chunks(HL / 32, F / 8) + rowvec_chunks(HL / 8, 1)) / SC;
  a.feat = feat; a.M = M; a.Mp = Mp; a.lm = lm; a.hl = hl;
  hipStream_t st = (hipStream_t)stream;
  // the head's kernels are never cut into point ranges: called INSIDE a chain (a C caller may; the module calls it outside) the entry
  // point joins the ranges into `st` first -- `feat` is written by the ranges of i2sdf_sdf_forward_grad -- and fences them behind itself
  ChainGuard guard(p, st, false);
  const unsigned grid = (unsigned)((M + PTS_PER_WG - 1) / PTS_PER_WG);
  if (p->rgb_bf16x3 && HL == 128 && F == 256 && p->light.fwd3h_chunks > 0) {      // bf16x3 split arithmetic with the radiance net's option
    LightFwd3hArgs x{};
    x.fwd = packed + p->scale_floats + p->light.fwd3h_chunk0 * CHUNK_FLOATS;
    x.n_fwd = (int)(p->light.fwd3h_chunks / SCH);
    x.feat = feat; x.M = M; x.lm = lm; x.hl = hl;
    i2sdf_launch_light_fwd3h(x, (unsigned)((M + 8 * HP - 1) / (8 * HP)), st);
    return i2sdf_hip_check(hipGetLastError(), "light_forward launch");
  }
  if (HL == 128 && F == 256) launch_lds(light_fwd_kernel<128, 256>, grid, st, a);
  else if (HL == 32 && F == 64) launch_lds(light_fwd_kernel<32, 64>, grid, st, a);
  else return I2SDF_EINVAL;
  return i2sdf_hip_check(hipGetLastError(), "light_forward launch");
}

extern "C" int i2sdf_light_backward(const i2sdf_plan* p, const float* packed, const float* lm, const float* lm_bar, const float* hl, int64_t M,
                                    int64_t Mp, float* gal0, float* gal_last, void* stream) {
  if (M == 0) return I2SDF_OK;                 // empty batch: nothing to validate, nothing to launch
  if (!p || !packed || !lm || !lm_bar || !hl || !gal0 || !gal_last || M < 0 || p->light.d.n_lin == 0) return I2SDF_EINVAL;
  if (Mp < M || Mp % PTS_PER_WG) return I2SDF_EINVAL;
  LightArgs a{};
  const int HL = p->light.d.hidden;
  a.rev = packed + p->scale_floats + p->light.rev_chunk0 * CHUNK_FLOATS;
  a.n_rev = rowvec_chunks(HL / 8, 1) / SC;
  a.M = M; a.Mp = Mp; a.lm = const_cast<float*>(lm); a.lm_bar = lm_bar; a.hl = const_cast<float*>(hl); a.gal0 = gal0; a.gal_last = gal_last;
  hipStream_t st = (hipStream_t)stream;
  ChainGuard guard(p, st, false);         // as in i2sdf_light_forward: whole-batch launches on `st`, joined / fenced when called inside a chain
  const unsigned grid = (unsigned)((M + PTS_PER_WG - 1) / PTS_PER_WG);
  if (HL == 128) launch_lds(light_bwd_kernel<128>, grid, st, a);
  else if (HL == 32) launch_lds(light_bwd_kernel<32>, grid, st, a);
  else return I2SDF_EINVAL;
  return i2sdf_hip_check(hipGetLastError(), "light_backward launch");
}
