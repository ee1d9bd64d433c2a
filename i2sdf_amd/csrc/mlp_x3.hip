// bf16x3 split-arithmetic kernels on 32-point waves (x3.h): sdf-only forward of 64-wide nets, d sdf/dx chain, backward sweeps (the other
// bf16x3 kernels run on 16-point waves: mlp_x3h.hip).
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

// d sdf/dx chain of the bf16x3 path as its own launch (appendix A.2): the forward kernel above and this one each fit the
// register file without spills; h_{L-1} is re-read from the tensor the forward saved.
// SV: a.abars is given (a backward will follow) -> its stores are unconditional instructions, counted by the stage waits (x3.h)
// P24: abars as packed 24-bit records (x3.h; only with SV)
template <int H, int LF, bool SV, bool P24 = false>
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
  const int64_t arow_ = P24 ? p24_row_off(m) : mrow;                             // ... and in abars
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
  if (P24) {
#pragma unroll
    for (int kc = 0; kc < KH16; ++kc) {
      float v8[8];
#pragma unroll
      for (int u = 0; u < 8; ++u) v8[u] = h[8 * kc + u];
      p24_store8(a.abars + (a.L - 2) * lstride + arow_, kc, hi, v8);
    }
  } else if (a.abars) store_regs<KC>(a.abars + (a.L - 2) * lstride + mrow, hi, valid, h, kcs);
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
    X3RevSrc<NT, SV, P24> src{accA, hrow, (SV || a.abars) ? a.abars + l * lstride + arow_ : nullptr, hi, valid, kcs};
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
    X3RevSrc<NT, SV, P24> src{accA, a.hs + mcrow, (SV || a.abars) ? a.abars + arow_ : nullptr, hi, valid, kcs};     // abar_0 = (.) * sigma(h_1)
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
template <int H, int LF, bool P24 = false>
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
    X3Sweep1Src<NT, 0, NGP, X3_AHEAD, P24> src{accB, gpx, nullptr, nullptr, hi};
    dense_x3g<NT, PE16, 2>(ws, src, accA, tid);
  }
  for (int l = 1; l < a.L - 1; ++l) {
    // the B preparation of layer l is the epilogue of layer l-1: G(hbar_l) -> gus[l]  (G2(a_{l-1}) is formed by sweep 2 from it, x3.h)
    X3Sweep1Src<NT, KH16, NGP, X3_AHEAD, P24> src{accA, gpx, a.hs + (l - 1) * lstride + mcrow, a.gus + l * lstride + (P24 ? p24_row_off(m) : mrow), hi, kcs};
    if (l == a.skip) dense_x3g<NT, KH16 + PE16, 2>(ws, src, accB, tid);
    else dense_x3g<NT, KH16, 2>(ws, src, accB, tid);
#pragma unroll
    for (int nt = 0; nt < NT; ++nt) accA[nt] = accB[nt];
  }
  {
    const int l = a.L - 1;
    // (G(hbar_{L-1}) stays fp32 also with packed records: x3.h X3Sweep2Src)
    X3Sweep1Src<NT, KH16, NGP, X3_DRAIN_AHEAD, false> src{accA, gpx, a.hs + (l - 1) * lstride + mcrow, a.gus + l * lstride + mrow, hi, kcs};
    x3_drain<KH16>(src);
  }
}

template <int H, int F, int LF, bool P24 = false>
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
  // rows of gus / abars (read) and gas (written: this point, padding rows included) -- fp32 blocked or packed 24-bit records.  fp32 rows are read at the CLAMPED
  // point; packed rows at the lane's OWN point, padding lanes included: p24_load8 / p24_store8 address the mid-byte part through the lane's index in its 32-point
  // block (threadIdx.x & 31 == m & 31), and the padding rows of abars / gus exist and hold what the producers' padding lanes wrote (unconditional stores, Mp rows)
  const int64_t srow = P24 ? p24_row_off(m) : mcrow, wrow = P24 ? p24_row_off(m) : mrow;
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
    X3Sweep2Src<NT, true, X3_SW2_AHEAD, P24, false> src{accA, a.hs + l * lstride + mcrow, a.gus + (l + 1) * lstride + mcrow, a.abars + l * lstride + srow,
                                                 a.gas + l * lstride + wrow, hi, sb, a.rev + lane * 4, kcs};
    zero(accB);
    dense_x3g<NT, KH16, 0>(ws, src, accB, tid);
#pragma unroll
    for (int nt = 0; nt < NT; ++nt) accA[nt] = accB[nt];
  }
  for (int l = a.L - 3; l >= 1; --l) {
    X3Sweep2Src<NT, false, X3_SW2_AHEAD, P24> src{accA, a.hs + l * lstride + mcrow, a.gus + (l + 1) * lstride + srow, a.abars + l * lstride + srow,
                                                  a.gas + l * lstride + wrow, hi, 0.f, nullptr, kcs};
    zero(accB);
    dense_x3g<NT, KH16, 0>(ws, src, accB, tid);
    if (l == a.skip) ws.skip(x3_bwd_chunks(PT, KH16) / SC, tid);      // the PE rows of W_skip^T are not needed here
#pragma unroll
    for (int nt = 0; nt < NT; ++nt) accA[nt] = accB[nt];
  }
  {
    X3Sweep2Src<NT, false, X3_DRAIN_AHEAD, P24> src{accA, a.hs + mcrow, a.gus + lstride + srow, a.abars + srow, a.gas + wrow, hi, 0.f, nullptr, kcs};     // G(a_0)
    x3_drain<KH16>(src);
  }
}

}  // namespace

// The packing instantiations (I2SDF_OPT_SAVES24: abars / gus / gas as packed 24-bit records) are compiled in a SECOND translation unit, mlp_x3p.hip, which
// includes this file with I2SDF_X3_P24_TU defined: the three kernels with and without packing in one unit took hipcc 8 minutes, single-threaded; as two
// units the build stays at ~4.5 minutes (build.sh compiles the units in parallel).
#ifdef I2SDF_X3_P24_TU
void i2sdf_launch_igrad3_p24(const SdfTrainFwdArgs& a, unsigned grid, hipStream_t st) { launch_lds(sdf_igrad3_kernel<256, 6, true, true>, grid, st, a); }
void i2sdf_launch_sdf_bwd3_p24(const SdfBwdArgs& a, unsigned grid, hipStream_t st) {
  launch_lds(sdf_bwd3_sweep1_kernel<256, 6, true>, grid, st, a);
  launch_lds(sdf_bwd3_sweep2_kernel<256, 256, 6, true>, grid, st, a);
}
#else
void i2sdf_launch_igrad3_p24(const SdfTrainFwdArgs& a, unsigned grid, hipStream_t st);      // mlp_x3p.hip
void i2sdf_launch_sdf_bwd3_p24(const SdfBwdArgs& a, unsigned grid, hipStream_t st);
// 64-wide nets only: the 256-wide sdf-only forward runs on 16-point waves (mlp_x3h.hip)
void i2sdf_launch_sdf_fwd3(int H, const float* stream, int n_stages, int L, int skip, const PointSpec& ps, const int* skip_flag, int64_t M,
                           float* sdf_out, unsigned grid, hipStream_t st) {
  (void)H;
  launch_lds(sdf_fwd3_kernel<64, 6>, grid, st, stream, n_stages, L, skip, ps, skip_flag, M, sdf_out);
}
void i2sdf_launch_igrad3(const SdfTrainFwdArgs& a, unsigned grid, hipStream_t st) {
  if (a.abars && a.p24) i2sdf_launch_igrad3_p24(a, grid, st);
  else if (a.abars) launch_lds(sdf_igrad3_kernel<256, 6, true>, grid, st, a);
  else launch_lds(sdf_igrad3_kernel<256, 6, false>, grid, st, a);
}
void i2sdf_launch_sdf_bwd3(const SdfBwdArgs& a, unsigned grid, hipStream_t st) {
  if (a.p24) { i2sdf_launch_sdf_bwd3_p24(a, grid, st); return; }
  launch_lds(sdf_bwd3_sweep1_kernel<256, 6>, grid, st, a);
  launch_lds(sdf_bwd3_sweep2_kernel<256, 256, 6>, grid, st, a);
}
#endif
