// bf16x3 split-arithmetic twins of the SDF training kernels (x3.h): forward with saves, d sdf/dx chain, backward sweeps.
// Kept in their own translation unit: their fully unrolled K-outer stage loops need -mllvm -pragma-unroll-threshold (build.sh),
// which the fp32 MFMA kernels must not be compiled with (it changes their unrolling and costs them ~4 %).
#ifndef I2SDF_NO_RELU_ASM      // (A/B builds)
#define I2SDF_RELU_ASM 1      // common.h: relu0
#endif
#include "mlp_args.h"

using namespace i2sdf;

namespace {

// bf16x3 variant of the sdf-only forward (sampler passes, grid queries; fp32 twin: mlp_fwd.hip): same result to fp32 rounding level
// at 3/8 of the matrix-pipe cycles.
template <int H, int LF>
__global__ __launch_bounds__(256) void sdf_fwd3_kernel(const float* __restrict__ stream, int n_stages, int L, int skip, PointSpec ps,
                                                        const int* __restrict__ skip_flag, int64_t M, float* __restrict__ sdf_out) {
  constexpr int NT = H / 32, KC = H / 8, KH16 = H / 16, PED = PE<LF>::DIM, PE16 = cdiv(PED, 16), NPE = PE16 * 8;
  extern __shared__ __attribute__((aligned(16))) float lds[];
  if (skip_flag != nullptr && skip_flag[0] != 0) return;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, hi = lane >> 5;
  const int64_t m = ((int64_t)blockIdx.x * 4 + wave) * 32 + (lane & 31);
  const bool valid = m < M;
  const int64_t mc = valid ? m : M - 1;
  float px, py, pz;
  fetch_point(ps, mc, px, py, pz);
  float pe[NPE];
  {
    float full[PE<LF>::PEC * 8], pad[PE16 * 16];
    pe_full<LF>(px, py, pz, full);
#pragma unroll
    for (int i = 0; i < PE16 * 16; ++i) pad[i] = (i < PED) ? full[i] : 0.f;
    x3_select_pe<PE16>(pad, pe, hi);
  }
  WStream ws;
  ws.begin(stream, lds, n_stages, tid);
  f32x16 accP[NT], accN[NT];
  {
    X3FwdSrc<NT, 0, NPE, false> src{accN, pe, nullptr, hi, valid};
    dense_x3g<NT, PE16, 1>(ws, src, accP, tid);
  }
  for (int l = 1; l < L - 1; ++l) {
    X3FwdSrc<NT, KH16, NPE, false> src{accP, pe, nullptr, hi, valid};
    if (l == skip) dense_x3g<NT, KH16 + PE16, 1>(ws, src, accN, tid);
    else dense_x3g<NT, KH16, 1>(ws, src, accN, tid);
#pragma unroll
    for (int nt = 0; nt < NT; ++nt) accP[nt] = accN[nt];
  }
  float h[NT * 16];
#pragma unroll
  for (int nt = 0; nt < NT; ++nt)
#pragma unroll
    for (int r = 0; r < 16; ++r) h[nt * 16 + r] = softplus100(accP[nt][r]);
  float s[1];
  rowvec_op<1, KC>(ws, h, s, tid);
  if (valid && hi == 0) sdf_out[m] = s[0];
}

// bf16x3 variant (x3.h): same outputs and saved tensors, K-outer loops on the bf16 matrix pipe.  The activations of a layer
// are produced (softplus, store, split) as the B operand of the NEXT op, one k-chunk ahead of their use; two accumulator
// sets alternate between "previous layer" and "this layer".
template <int H, int F, int LF, bool GRAD>
__global__ __launch_bounds__(256) void sdf_train_fwd3_kernel(SdfTrainFwdArgs a) {
  constexpr int NT = H / 32, KC = H / 8, KH16 = H / 16, PEC = PE<LF>::PEC, PED = PE<LF>::DIM, PT = cdiv(PEC * 8, 32), FT = F / 32;
  constexpr int PE16 = cdiv(PED, 16), NPE = PE16 * 8;
  static_assert(NT == FT, "feature tiles reuse the hidden accumulator set");
  extern __shared__ __attribute__((aligned(16))) float lds[];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, hi = lane >> 5;
  const int64_t m = ((int64_t)(blockIdx.x + a.wg0) * 4 + wave) * 32 + (lane & 31);
  const bool valid = m < a.M;
  const int64_t mc = valid ? m : a.M - 1;
  float px, py, pz;
  fetch_point(a.pts, mc, px, py, pz);
  float pe[NPE];
  {
    float full[PEC * 8], pad[PE16 * 16], reg[PEC * 4];
    pe_full<LF>(px, py, pz, full);
    if (a.pe_save) { to_b_layout<PEC>(full, reg, hi); store_regs<PEC>(a.pe_save + m * (PEC * 8), hi, valid, reg); }
#pragma unroll
    for (int i = 0; i < PE16 * 16; ++i) pad[i] = (i < PED) ? full[i] : 0.f;
    x3_select_pe<PE16>(pad, pe, hi);
  }
  const int64_t lstride = a.Mp * H;
  const int kcs = a.kcs;
  const int64_t mrow = save_row_off(m, kcs), mcrow = save_row_off(mc, kcs);     // this point's row in the saved tensors
  WStream ws;
  ws.begin(a.fwd, lds, a.n_fwd, tid);
  f32x16 accA[NT], accB[NT];
  {
    X3FwdSrc<NT, 0, NPE> src{accB, pe, nullptr, hi, valid};
    dense_x3g<NT, PE16, 1>(ws, src, accA, tid);
  }
  // hidden layers: layer l reads accA (pre-activations of layer l-1, whose softplus is h_l -> hs[l-1]) and writes accB
  for (int l = 1; l < a.L - 1; ++l) {
    float* hrow = a.hs ? a.hs + (l - 1) * lstride + mrow : nullptr;
    if (l == a.skip) {
      X3FwdSrc<NT, KH16, NPE> src{accA, pe, hrow, hi, valid, kcs};
      dense_x3g<NT, KH16 + PE16, 1>(ws, src, accB, tid);
    } else {
      X3FwdSrc<NT, KH16, NPE> src{accA, pe, hrow, hi, valid, kcs};
      dense_x3g<NT, KH16, 1>(ws, src, accB, tid);
    }
#pragma unroll
    for (int nt = 0; nt < NT; ++nt) accA[nt] = accB[nt];
  }
  float h[KC * 4];                     // h_{L-1} in the fp32 kernels' B layout: sdf row, feature op, top of the reverse chain
#pragma unroll
  for (int nt = 0; nt < NT; ++nt)
#pragma unroll
    for (int r = 0; r < 16; ++r) h[nt * 16 + r] = softplus100(accA[nt][r]);
  if (a.hs) store_regs<KC>(a.hs + (a.L - 2) * lstride + mrow, hi, valid, h, kcs);
  {
    float s[1];
    rowvec_op<1, KC>(ws, h, s, tid);
    if (valid && hi == 0) a.sdf[m] = s[0];
  }
  if (a.feat != nullptr) {
    X3RegSrc<KC * 4> src{h};
    dense_x3g<FT, KH16, 1>(ws, src, accB, tid);
    store_tile<FT>(a.feat + mc * F, hi, valid, accB);
  }
}

// d sdf/dx chain of the bf16x3 path as its own launch (appendix A.2): the forward kernel above and this one each fit the
// register file without spills; h_{L-1} is re-read from the tensor the forward saved.
template <int H, int LF>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(1, 1))) void sdf_igrad3_kernel(SdfTrainFwdArgs a) {
  constexpr int NT = H / 32, KC = H / 8, KH16 = H / 16, PEC = PE<LF>::PEC, PT = cdiv(PEC * 8, 32);
  extern __shared__ __attribute__((aligned(16))) float lds[];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, hi = lane >> 5;
  const int64_t m = ((int64_t)(blockIdx.x + a.wg0) * 4 + wave) * 32 + (lane & 31);
  const bool valid = m < a.M;
  const int64_t mc = valid ? m : a.M - 1;
  const int64_t lstride = a.Mp * H;
  const int kcs = a.kcs;
  const int64_t mrow = save_row_off(m, kcs), mcrow = save_row_off(mc, kcs);     // this point's row in the saved tensors
  float px, py, pz;
  fetch_point(a.pts, mc, px, py, pz);
  f32x16 accA[NT], accB[NT];
  float h[KC * 4];
  load_regs<KC>(a.hs + (a.L - 2) * lstride + mcrow, hi, h, kcs);
  WStream ws;
  ws.begin(a.rev, lds, a.n_rev, tid);
  {
    float wv[KC * 4];
    f32x4 sc;
    rowvec_load<KC>(ws, wv, sc, tid);
#pragma unroll
    for (int i = 0; i < KC * 4; ++i) h[i] = wv[i] * sp_sigma_from_h(h[i]);       // abar_{L-2} = w_sdf (.) sigma_{L-2}
  }
  if (a.abars) store_regs<KC>(a.abars + (a.L - 2) * lstride + mrow, hi, valid, h, kcs);
  f32x16 pt[PT];
#pragma unroll
  for (int i = 0; i < PT; ++i)
#pragma unroll
    for (int r = 0; r < 16; ++r) pt[i][r] = 0.f;
  auto zero = [&](f32x16 (&x)[NT]) __attribute__((always_inline)) {
#pragma unroll
    for (int nt = 0; nt < NT; ++nt)
#pragma unroll
      for (int r = 0; r < 16; ++r) x[nt][r] = 0.f;
  };
  // op(l) = W_l^T abar_l.  abar_{L-2} comes from registers; for l < L-2 it is (accumulators of op(l+1)) * sigma(h_{l+1}),
  // made (and stored to abars[l]) by the B preparation of op(l).  The skip layer's op is two ops over the same B operand:
  // the hidden part and the PE part (which adds into pbar; its B preparation is recomputed, without the store).
  {
    X3RegSrc<KC * 4> src{h};
    zero(accA);
    dense_x3g<NT, KH16, 0>(ws, src, accA, tid);       // l = L-2 (never the skip layer, checked by the host)
  }
  for (int l = a.L - 3; l >= 1; --l) {
    const float* hrow = a.hs + l * lstride + mcrow;
    X3RevSrc<NT> src{accA, hrow, a.abars ? a.abars + l * lstride + mrow : nullptr, hi, valid, kcs};
    zero(accB);
    dense_x3g<NT, KH16, 0>(ws, src, accB, tid);
    if (l == a.skip) {
      X3RevSrc<NT> src2{accA, hrow, nullptr, hi, valid, kcs};
      dense_x3g<PT, KH16, 0>(ws, src2, pt, tid);
    }
#pragma unroll
    for (int nt = 0; nt < NT; ++nt) accA[nt] = accB[nt];
  }
  {
    X3RevSrc<NT> src{accA, a.hs + mcrow, a.abars ? a.abars + mrow : nullptr, hi, valid, kcs};     // abar_0 = (.) * sigma(h_1)
    dense_x3g<PT, KH16, 0>(ws, src, pt, tid);       // pbar += W_0^T abar_0
  }
  {
    float full[PEC * 8], coef[PEC * 8], n[3];
    pe_full<LF>(px, py, pz, full);
    pe_coef<LF>(full, coef);
    pe_jt_apply<LF, PT>(coef, pt, hi, n);
    if (valid && hi == 0) { a.grad[m * 3 + 0] = n[0]; a.grad[m * 3 + 1] = n[1]; a.grad[m * 3 + 2] = n[2]; }
  }
}

// ---------------------------------------------------------------------------------------------------------------
// bf16x3 variants (x3.h) of the two sweeps, one launch each (each fits the register file without spills).  K-outer: the
// per-element epilogue of an op (loads of the saved tensors, sigma products, stores) is the B preparation of the next op;
// the last op of a sweep is followed by a drain that only runs the epilogue.
// ---------------------------------------------------------------------------------------------------------------
template <int H, int LF>
__global__ __launch_bounds__(256) void sdf_bwd3_sweep1_kernel(SdfBwdArgs a) {
  constexpr int NT = H / 32, KH16 = H / 16, PEC = PE<LF>::PEC, PED = PE<LF>::DIM, PE16 = cdiv(PED, 16), NGP = PE16 * 8;
  extern __shared__ __attribute__((aligned(16))) float lds[];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, hi = lane >> 5;
  const int64_t m = ((int64_t)(blockIdx.x + a.wg0) * 4 + wave) * 32 + (lane & 31);
  const bool valid = m < a.M;
  const int64_t mc = valid ? m : a.M - 1;
  const int64_t lstride = a.Mp * H;
  const int kcs = a.kcs;
  const int64_t mrow = save_row_off(m, kcs), mcrow = save_row_off(mc, kcs);     // this point's row in the saved tensors
  float gpx[NGP];                       // G(pbar) in the fp32 kernels' B layout, zero padded to whole 16-chunks
  {
    float gp[PEC * 4];
    float px, py, pz, full[PEC * 8], coef[PEC * 8], nb[3] = {0.f, 0.f, 0.f};
    fetch_point(a.pts, mc, px, py, pz);
    pe_full<LF>(px, py, pz, full);
    pe_coef<LF>(full, coef);
    if (a.nbar) { nb[0] = a.nbar[mc * 3 + 0]; nb[1] = a.nbar[mc * 3 + 1]; nb[2] = a.nbar[mc * 3 + 2]; }
    pe_j_apply<LF>(coef, nb, hi, gp);
    store_regs<PEC>(a.gpbar + m * (PEC * 8), hi, valid, gp);
#pragma unroll
    for (int i = 0; i < NGP; ++i) gpx[i] = i < PEC * 4 ? gp[i < PEC * 4 ? i : 0] : 0.f;
  }
  WStream ws;
  ws.begin(a.fwd, lds, a.n_fwd, tid);
  f32x16 accA[NT], accB[NT];
  {
    X3Sweep1Src<NT, 0, NGP> src{accB, gpx, nullptr, nullptr, nullptr, nullptr, hi, valid};
    dense_x3g<NT, PE16, 2>(ws, src, accA, tid);
  }
  for (int l = 1; l < a.L - 1; ++l) {
    // the B preparation of layer l is the epilogue of layer l-1: G(hbar_l) -> gus[l], G2(a_{l-1}) -> gas[l-1]
    X3Sweep1Src<NT, KH16, NGP> src{accA, gpx, a.hs + (l - 1) * lstride + mcrow, a.abars + (l - 1) * lstride + mcrow,
                                   a.gas + (l - 1) * lstride + mrow, a.gus + l * lstride + mrow, hi, valid, kcs};
    if (l == a.skip) dense_x3g<NT, KH16 + PE16, 2>(ws, src, accB, tid);
    else dense_x3g<NT, KH16, 2>(ws, src, accB, tid);
#pragma unroll
    for (int nt = 0; nt < NT; ++nt) accA[nt] = accB[nt];
  }
  {
    const int l = a.L - 1;
    X3Sweep1Src<NT, KH16, NGP> src{accA, gpx, a.hs + (l - 1) * lstride + mcrow, a.abars + (l - 1) * lstride + mcrow,
                                   a.gas + (l - 1) * lstride + mrow, a.gus + l * lstride + mrow, hi, valid, kcs};
    x3_drain<KH16>(src);
  }
}

template <int H, int F, int LF>
__global__ __launch_bounds__(256) void sdf_bwd3_sweep2_kernel(SdfBwdArgs a) {
  constexpr int NT = H / 32, KC = H / 8, KH16 = H / 16, PEC = PE<LF>::PEC, PT = cdiv(PEC * 8, 32);
  extern __shared__ __attribute__((aligned(16))) float lds[];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, hi = lane >> 5;
  const int64_t m = ((int64_t)(blockIdx.x + a.wg0) * 4 + wave) * 32 + (lane & 31);
  const bool valid = m < a.M;
  const int64_t mc = valid ? m : a.M - 1;
  const int64_t lstride = a.Mp * H;
  const int kcs = a.kcs;
  const int64_t mrow = save_row_off(m, kcs), mcrow = save_row_off(mc, kcs);     // this point's row in the saved tensors
  const float sb = a.sbar ? a.sbar[mc] : 0.f;
  if (valid && hi == 0) {
    *reinterpret_cast<f32x4*>(a.ga_last4 + m * 4) = f32x4{sb, 0.f, 0.f, 0.f};
    *reinterpret_cast<f32x4*>(a.ones4 + m * 4) = f32x4{1.f, 0.f, 0.f, 0.f};
  }
  WStream ws;
  ws.begin(a.rev, lds, a.n_rev, tid);
  f32x16 accA[NT], accB[NT];
  auto zero = [&](f32x16 (&x)[NT]) __attribute__((always_inline)) {
#pragma unroll
    for (int nt = 0; nt < NT; ++nt)
#pragma unroll
      for (int r = 0; r < 16; ++r) x[nt][r] = 0.f;
  };
  ws.skip(rowvec_chunks(KC, 1) / SC, tid);            // w_sdf is read straight from the packed buffer (X3Sweep2Src<TOP>)
  {
    X3RowSrc src{a.fbar ? a.fbar + mc * F : nullptr, hi, a.fbar != nullptr && mc < a.m_fbar};
    zero(accA);
    dense_x3g<NT, F / 16, 0>(ws, src, accA, tid);     // W_feat^T fbar
  }
  ws.skip(rowvec_chunks(KC, 1) / SC, tid);            // the d sdf/dx chain's copy of w_sdf
  {
    const int l = a.L - 2;                            // G(a_{L-2}) = (W_feat^T fbar + sbar w_sdf) sigma + G2, then W_{L-2}^T G(a_{L-2})
    X3Sweep2Src<NT, true> src{accA, a.hs + l * lstride + mcrow, a.gas + l * lstride + mcrow, a.gas + l * lstride + mrow, hi, valid,
                              sb, a.rev + lane * 4, kcs};
    zero(accB);
    dense_x3g<NT, KH16, 0>(ws, src, accB, tid);
#pragma unroll
    for (int nt = 0; nt < NT; ++nt) accA[nt] = accB[nt];
  }
  for (int l = a.L - 3; l >= 1; --l) {
    X3Sweep2Src<NT, false> src{accA, a.hs + l * lstride + mcrow, a.gas + l * lstride + mcrow, a.gas + l * lstride + mrow, hi, valid,
                               0.f, nullptr, kcs};
    zero(accB);
    dense_x3g<NT, KH16, 0>(ws, src, accB, tid);
    if (l == a.skip) ws.skip(x3_bwd_chunks(PT, KH16) / SC, tid);      // the PE rows of W_skip^T are not needed here
#pragma unroll
    for (int nt = 0; nt < NT; ++nt) accA[nt] = accB[nt];
  }
  {
    X3Sweep2Src<NT, false> src{accA, a.hs + mcrow, a.gas + mcrow, a.gas + mrow, hi, valid, 0.f, nullptr, kcs};     // G(a_0)
    x3_drain<KH16>(src);
  }
}

// ---------------------------------------------------------------------------------------------------------------
// Radiance net (RenderingNetwork, 'nerf' mode: mlp.py:208-229) forward and backward, bf16x3 twins of rgb_fwd_kernel /
// rgb_bwd_kernel.
// ---------------------------------------------------------------------------------------------------------------
template <int H, int F, int LFV>
__global__ __launch_bounds__(256) void rgb_fwd3_kernel(RgbFwdArgs a) {
  constexpr int NT = H / 32, KC = H / 8, KH16 = H / 16, PECV = PE<LFV>::PEC, PEDV = PE<LFV>::DIM, PV16 = cdiv(PEDV, 16);
  extern __shared__ __attribute__((aligned(16))) float lds[];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, hi = lane >> 5;
  const int64_t m = ((int64_t)(blockIdx.x + a.wg0) * 4 + wave) * 32 + (lane & 31);
  const bool valid = m < a.M;
  const int64_t mc = valid ? m : a.M - 1;
  const int64_t ray = mc / a.n_per_ray;
  float pev[PV16 * 8];
  {
    float full[PECV * 8], pv[PECV * 4], pad[PV16 * 16];
    pe_full<LFV>(a.dirs[ray * 3 + 0], a.dirs[ray * 3 + 1], a.dirs[ray * 3 + 2], full);
    to_b_layout<PECV>(full, pv, hi);
    if (a.pev_save) store_regs<PECV>(a.pev_save + m * (PECV * 8), hi, valid, pv);
#pragma unroll
    for (int i = 0; i < PV16 * 16; ++i) pad[i] = (i < PEDV) ? full[i < PECV * 8 ? i : 0] : 0.f;
    x3_select_pe<PV16>(pad, pev, hi);
  }
  const int64_t lstride = a.Mp * H;
  const int kcs = a.kcs;
  const int64_t mrow = save_row_off(m, kcs), mcrow = save_row_off(mc, kcs);     // this point's row in the saved tensors
  WStream ws;
  ws.begin(a.fwd, lds, a.n_fwd, tid);
  f32x16 accA[NT], accB[NT];
  {
    X3PeRowSrc<PV16> src{pev, a.feat + mc * F, hi};
    dense_x3g<NT, PV16 + F / 16, 1>(ws, src, accA, tid);
  }
  for (int l = 1; l < a.L - 1; ++l) {
    X3ReluSrc<NT> src{accA, a.rs ? a.rs + (l - 1) * lstride + mrow : nullptr, hi, valid, kcs};
    dense_x3g<NT, KH16, 1>(ws, src, accB, tid);
#pragma unroll
    for (int nt = 0; nt < NT; ++nt) accA[nt] = accB[nt];
  }
  float r[KC * 4];
#pragma unroll
  for (int nt = 0; nt < NT; ++nt)
#pragma unroll
    for (int q = 0; q < 16; ++q) r[nt * 16 + q] = fmaxf(accA[nt][q], 0.f);
  if (a.rs) store_regs<KC>(a.rs + (a.L - 2) * lstride + mrow, hi, valid, r, kcs);
  float o[3];
  rowvec_op<3, KC>(ws, r, o, tid);
  if (valid && hi == 0) {
#pragma unroll
    for (int i = 0; i < 3; ++i) a.rgb[m * 3 + i] = 1.0f / (1.0f + expf(-o[i]));
  }
}

template <int H, int F>
__global__ __launch_bounds__(256) void rgb_bwd3_kernel(RgbBwdArgs a) {
  constexpr int NT = H / 32, KC = H / 8, KH16 = H / 16, FT = F / 32;
  static_assert(FT == NT, "feature tiles reuse the hidden accumulator set");
  extern __shared__ __attribute__((aligned(16))) float lds[];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, hi = lane >> 5;
  const int64_t m = ((int64_t)(blockIdx.x + a.wg0) * 4 + wave) * 32 + (lane & 31);
  const bool valid = m < a.M;
  const int64_t mc = valid ? m : a.M - 1;
  const int64_t lstride = a.Mp * H;
  const int kcs = a.kcs;
  const int64_t mrow = save_row_off(m, kcs), mcrow = save_row_off(mc, kcs);     // this point's row in the saved tensors
  float g3[3];
#pragma unroll
  for (int j = 0; j < 3; ++j) {
    const float c = a.rgb[mc * 3 + j];
    g3[j] = a.rgb_bar[mc * 3 + j] * c * (1.0f - c);
  }
  if (valid && hi == 0) *reinterpret_cast<f32x4*>(a.ga_last + m * 4) = f32x4{g3[0], g3[1], g3[2], 0.f};
  WStream ws;
  ws.begin(a.rev, lds, a.n_rev, tid);
  float ga[KC * 4];
  {
    // G(r_{L-1}) = W_last^T G(a_last): three row vectors, then the top mask r_{L-1} > 0 -> G(a_{L-2})
    constexpr int NW = 3 * KC, TOT = rowvec_chunks(KC, 3), NS = TOT / SC;
#pragma unroll
    for (int i = 0; i < KC * 4; ++i) ga[i] = 0.f;
#pragma unroll
    for (int s = 0; s < NS; ++s) {
      const f32x4* cur = reinterpret_cast<const f32x4*>(ws.advance(tid)) + lane;
#pragma unroll
      for (int j = 0; j < SC; ++j) {
        const int c = s * SC + j;
        if (c < NW) {
          const int row = c / KC, kc = c % KC;
          const f32x4 w = cur[j * 64];
          ga[kc * 4 + 0] = fmaf(w.x, g3[row], ga[kc * 4 + 0]);
          ga[kc * 4 + 1] = fmaf(w.y, g3[row], ga[kc * 4 + 1]);
          ga[kc * 4 + 2] = fmaf(w.z, g3[row], ga[kc * 4 + 2]);
          ga[kc * 4 + 3] = fmaf(w.w, g3[row], ga[kc * 4 + 3]);
        }
      }
    }
    const int l = a.L - 2;
    const float* rrow = a.rs + l * lstride + mcrow;
    float* grow = a.gar + l * lstride + mrow;
#pragma unroll
    for (int c = 0; c < KC; ++c) {
      const f32x4 rv = *reinterpret_cast<const f32x4*>(rrow + (c >> 1) * kcs + (c & 1) * 8 + 4 * hi);
      f32x4 o;
#pragma unroll
      for (int t = 0; t < 4; ++t) { o[t] = rv[t] > 0.f ? ga[c * 4 + t] : 0.f; ga[c * 4 + t] = o[t]; }
      if (valid) *reinterpret_cast<f32x4*>(grow + (c >> 1) * kcs + (c & 1) * 8 + 4 * hi) = o;
    }
  }
  f32x16 accA[NT], accB[NT];
  auto zero = [&](f32x16 (&x)[NT]) __attribute__((always_inline)) {
#pragma unroll
    for (int nt = 0; nt < NT; ++nt)
#pragma unroll
      for (int r = 0; r < 16; ++r) x[nt][r] = 0.f;
  };
  {
    X3RegSrc<KC * 4> src{ga};
    zero(accA);
    dense_x3g<NT, KH16, 0>(ws, src, accA, tid);            // W_{L-2}^T G(a_{L-2})
  }
  for (int l = a.L - 3; l >= 1; --l) {
    X3MaskSrc<NT> src{accA, a.rs + l * lstride + mcrow, a.gar + l * lstride + mrow, hi, valid, kcs};
    zero(accB);
    dense_x3g<NT, KH16, 0>(ws, src, accB, tid);
#pragma unroll
    for (int nt = 0; nt < NT; ++nt) accA[nt] = accB[nt];
  }
  {
    X3MaskSrc<NT> src{accA, a.rs + mcrow, a.gar + mrow, hi, valid, kcs};     // G(a_0), then the feature rows of W_0^T
    zero(accB);
    dense_x3g<FT, KH16, 0>(ws, src, accB, tid);
    store_tile<FT>(a.fbar + mc * F, hi, valid, accB);
  }
}

}  // namespace

void i2sdf_launch_sdf_fwd3(int H, const float* stream, int n_stages, int L, int skip, const PointSpec& ps, const int* skip_flag, int64_t M,
                           float* sdf_out, unsigned grid, hipStream_t st) {
  if (H == 256) launch_lds(sdf_fwd3_kernel<256, 6>, grid, st, stream, n_stages, L, skip, ps, skip_flag, M, sdf_out);
  else launch_lds(sdf_fwd3_kernel<64, 6>, grid, st, stream, n_stages, L, skip, ps, skip_flag, M, sdf_out);
}
void i2sdf_launch_train_fwd3(const SdfTrainFwdArgs& a, bool fwd, bool grad, unsigned grid, hipStream_t st) {
  if (fwd) launch_lds(sdf_train_fwd3_kernel<256, 256, 6, false>, grid, st, a);
  if (grad) launch_lds(sdf_igrad3_kernel<256, 6>, grid, st, a);
}
void i2sdf_launch_sdf_bwd3(const SdfBwdArgs& a, unsigned grid, hipStream_t st) {
  launch_lds(sdf_bwd3_sweep1_kernel<256, 6>, grid, st, a);
  launch_lds(sdf_bwd3_sweep2_kernel<256, 256, 6>, grid, st, a);
}
void i2sdf_launch_rgb_fwd3(const RgbFwdArgs& a, unsigned grid, hipStream_t st) { launch_lds(rgb_fwd3_kernel<256, 256, 4>, grid, st, a); }
void i2sdf_launch_rgb_bwd3(const RgbBwdArgs& a, unsigned grid, hipStream_t st) { launch_lds(rgb_bwd3_kernel<256, 256>, grid, st, a); }
