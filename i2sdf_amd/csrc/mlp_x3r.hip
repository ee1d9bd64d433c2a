// bf16x3 kernels whose saved-tensor reads go through the per-wave LDS ring (x3r.h): backward sweeps, d sdf/dx chain, radiance
// backward.  Same arguments, same arithmetic and same results as their twins in mlp_x3.hip (which keep the VGPR-load form of
// the sources); selected by I2SDF_OPT_SRC_RING.  Own translation unit: each of these fully unrolled kernels takes minutes to compile.
#ifndef I2SDF_NO_RELU_ASM      // (A/B builds)
#define I2SDF_RELU_ASM 1      // common.h: relu0
#endif
#include "mlp_args.h"
#include "x3r.h"

using namespace i2sdf;

namespace {

__device__ __forceinline__ float* wave_ring(float* lds, int wave) { return lds + 2 * STAGE_FLOATS + wave * X3R_RING_FLOATS; }

template <int NT>
__device__ __forceinline__ void zero_tiles(f32x16 (&x)[NT]) {
#pragma unroll
  for (int nt = 0; nt < NT; ++nt)
#pragma unroll
    for (int r = 0; r < 16; ++r) x[nt][r] = 0.f;
}

// d sdf/dx chain (appendix A.2), ring twin of sdf_igrad3_kernel
template <int H, int LF>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(1, 1))) void sdf_igrad3r_kernel(SdfTrainFwdArgs a) {
  constexpr int NT = H / 32, KC = H / 8, KH16 = H / 16, PEC = PE<LF>::PEC, PT = cdiv(PEC * 8, 32);
  extern __shared__ __attribute__((aligned(16))) float lds[];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, hi = lane >> 5;
  float* ring = wave_ring(lds, wave);
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
  zero_tiles<PT>(pt);
  {
    X3rRegSrc<KC * 4> src{h};
    zero_tiles<NT>(accA);
    dense_x3r<NT, KH16, 0>(ws, src, accA, ring, tid);       // l = L-2 (never the skip layer, checked by the host)
  }
  for (int l = a.L - 3; l >= 1; --l) {
    const float* hrow = a.hs + l * lstride + mcrow;
    X3rRevSrc<NT> src{accA, hrow, a.abars ? a.abars + l * lstride + mrow : nullptr, hi, valid, kcs};
    zero_tiles<NT>(accB);
    dense_x3r<NT, KH16, 0>(ws, src, accB, ring, tid);
    if (l == a.skip) {
      X3rRevSrc<NT> src2{accA, hrow, nullptr, hi, valid, kcs};
      dense_x3r<PT, KH16, 0>(ws, src2, pt, ring, tid);
    }
#pragma unroll
    for (int nt = 0; nt < NT; ++nt) accA[nt] = accB[nt];
  }
  {
    X3rRevSrc<NT> src{accA, a.hs + mcrow, a.abars ? a.abars + mrow : nullptr, hi, valid, kcs};     // abar_0 = (.) * sigma(h_1)
    dense_x3r<PT, KH16, 0>(ws, src, pt, ring, tid);       // pbar += W_0^T abar_0
  }
  {
    float full[PEC * 8], coef[PEC * 8], n[3];
    pe_full<LF>(px, py, pz, full);
    pe_coef<LF>(full, coef);
    pe_jt_apply<LF, PT>(coef, pt, hi, n);
    if (valid && hi == 0) { a.grad[m * 3 + 0] = n[0]; a.grad[m * 3 + 1] = n[1]; a.grad[m * 3 + 2] = n[2]; }
  }
}

// backward sweep 1 (appendix A.3 step 1), ring twin of sdf_bwd3_sweep1_kernel
template <int H, int LF>
__global__ __launch_bounds__(256) void sdf_bwd3r_sweep1_kernel(SdfBwdArgs a) {
  constexpr int NT = H / 32, KH16 = H / 16, PEC = PE<LF>::PEC, PED = PE<LF>::DIM, PE16 = cdiv(PED, 16), NGP = PE16 * 8;
  extern __shared__ __attribute__((aligned(16))) float lds[];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, hi = lane >> 5;
  float* ring = wave_ring(lds, wave);
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
    X3rSweep1Src<NT, 0, NGP> src{accB, gpx, nullptr, nullptr, nullptr, nullptr, hi, valid};
    dense_x3r<NT, PE16, 2>(ws, src, accA, ring, tid);
  }
  for (int l = 1; l < a.L - 1; ++l) {
    // the B preparation of layer l is the epilogue of layer l-1: G(hbar_l) -> gus[l], G2(a_{l-1}) -> gas[l-1]
    X3rSweep1Src<NT, KH16, NGP> src{accA, gpx, a.hs + (l - 1) * lstride + mcrow, a.abars + (l - 1) * lstride + mcrow,
                                    a.gas + (l - 1) * lstride + mrow, a.gus + l * lstride + mrow, hi, valid, kcs};
    if (l == a.skip) dense_x3r<NT, KH16 + PE16, 2>(ws, src, accB, ring, tid);
    else dense_x3r<NT, KH16, 2>(ws, src, accB, ring, tid);
#pragma unroll
    for (int nt = 0; nt < NT; ++nt) accA[nt] = accB[nt];
  }
  {
    const int l = a.L - 1;
    X3rSweep1Src<NT, KH16, NGP> src{accA, gpx, a.hs + (l - 1) * lstride + mcrow, a.abars + (l - 1) * lstride + mcrow,
                                    a.gas + (l - 1) * lstride + mrow, a.gus + l * lstride + mrow, hi, valid, kcs};
    x3r_drain<KH16>(src, ring, tid);
  }
}

// backward sweep 2 (appendix A.3 step 2), ring twin of sdf_bwd3_sweep2_kernel
template <int H, int F, int LF>
__global__ __launch_bounds__(256) void sdf_bwd3r_sweep2_kernel(SdfBwdArgs a) {
  constexpr int NT = H / 32, KC = H / 8, KH16 = H / 16, PEC = PE<LF>::PEC, PT = cdiv(PEC * 8, 32);
  extern __shared__ __attribute__((aligned(16))) float lds[];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, hi = lane >> 5;
  float* ring = wave_ring(lds, wave);
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
  ws.skip(rowvec_chunks(KC, 1) / SC, tid);            // first copy of w_sdf (the fp32 kernels' feature op needs it up front)
  {
    X3rRowSrc src{(a.fbar ? a.fbar : a.hs) + mc * F, hi, a.fbar != nullptr && mc < a.m_fbar};
    zero_tiles<NT>(accA);
    dense_x3r<NT, F / 16, 0>(ws, src, accA, ring, tid);     // W_feat^T fbar
  }
  {
    // the d sdf/dx chain's copy of w_sdf, B layout; chunk 4*nt+q of the B layout is D-layout tile nt, registers 4q..4q+3:
    // (W_feat^T fbar + sbar w_sdf) is the upstream of the top hidden layer
    float wv[KC * 4];
    f32x4 sc;
    rowvec_load<KC>(ws, wv, sc, tid);
#pragma unroll
    for (int nt = 0; nt < NT; ++nt)
#pragma unroll
      for (int r = 0; r < 16; ++r) accA[nt][r] = fmaf(sb, wv[16 * nt + r], accA[nt][r]);
  }
  for (int l = a.L - 2; l >= 1; --l) {
    // G(a_l) = upstream * sigma_l + G2(a_l), then W_l^T G(a_l)
    X3rSweep2Src<NT, false> src{accA, a.hs + l * lstride + mcrow, a.gas + l * lstride + mcrow, a.gas + l * lstride + mrow, hi, valid,
                                0.f, nullptr, kcs};
    zero_tiles<NT>(accB);
    dense_x3r<NT, KH16, 0>(ws, src, accB, ring, tid);
    if (l == a.skip) ws.skip(x3_bwd_chunks(PT, KH16) / SC, tid);      // the PE rows of W_skip^T are not needed here
#pragma unroll
    for (int nt = 0; nt < NT; ++nt) accA[nt] = accB[nt];
  }
  {
    X3rSweep2Src<NT, false> src{accA, a.hs + mcrow, a.gas + mcrow, a.gas + mrow, hi, valid, 0.f, nullptr, kcs};     // G(a_0)
    x3r_drain<KH16>(src, ring, tid);
  }
}

// radiance backward, ring twin of rgb_bwd3_kernel
template <int H, int F>
__global__ __launch_bounds__(256) void rgb_bwd3r_kernel(RgbBwdArgs a) {
  constexpr int NT = H / 32, KC = H / 8, KH16 = H / 16, FT = F / 32;
  static_assert(FT == NT, "feature tiles reuse the hidden accumulator set");
  extern __shared__ __attribute__((aligned(16))) float lds[];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, hi = lane >> 5;
  float* ring = wave_ring(lds, wave);
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
  float ga[KC * 4];
  {
    // top mask operand r_{L-1}: loaded before the weight stream starts (ordinary loads must not follow a DMA in flight)
    const int l = a.L - 2;
    const float* rrow = a.rs + l * lstride + mcrow;
    load_regs<KC>(rrow, hi, ga, kcs);
  }
  WStream ws;
  ws.begin(a.rev, lds, a.n_rev, tid);
  {
    // G(r_{L-1}) = W_last^T G(a_last): three row vectors, then the top mask r_{L-1} > 0 -> G(a_{L-2})
    constexpr int NW = 3 * KC, TOT = rowvec_chunks(KC, 3), NS = TOT / SC;
    float acc1[KC * 4];
#pragma unroll
    for (int i = 0; i < KC * 4; ++i) acc1[i] = 0.f;
#pragma unroll
    for (int s = 0; s < NS; ++s) {
      const f32x4* cur = reinterpret_cast<const f32x4*>(ws.advance(tid)) + lane;
#pragma unroll
      for (int j = 0; j < SC; ++j) {
        const int c = s * SC + j;
        if (c < NW) {
          const int row = c / KC, kc = c % KC;
          const f32x4 w = cur[j * 64];
          acc1[kc * 4 + 0] = fmaf(w.x, g3[row], acc1[kc * 4 + 0]);
          acc1[kc * 4 + 1] = fmaf(w.y, g3[row], acc1[kc * 4 + 1]);
          acc1[kc * 4 + 2] = fmaf(w.z, g3[row], acc1[kc * 4 + 2]);
          acc1[kc * 4 + 3] = fmaf(w.w, g3[row], acc1[kc * 4 + 3]);
        }
      }
    }
    const int l = a.L - 2;
    float* grow = a.gar + l * lstride + mrow;
#pragma unroll
    for (int c = 0; c < KC; ++c) {
      f32x4 o;
#pragma unroll
      for (int t = 0; t < 4; ++t) { o[t] = ga[c * 4 + t] > 0.f ? acc1[c * 4 + t] : 0.f; ga[c * 4 + t] = o[t]; }
      if (valid) *reinterpret_cast<f32x4*>(grow + (c >> 1) * kcs + (c & 1) * 8 + 4 * hi) = o;
    }
  }
  f32x16 accA[NT], accB[NT];
  {
    X3rRegSrc<KC * 4> src{ga};
    zero_tiles<NT>(accA);
    dense_x3r<NT, KH16, 0>(ws, src, accA, ring, tid);            // W_{L-2}^T G(a_{L-2})
  }
  for (int l = a.L - 3; l >= 1; --l) {
    X3rMaskSrc<NT> src{accA, a.rs + l * lstride + mcrow, a.gar + l * lstride + mrow, hi, valid, kcs};
    zero_tiles<NT>(accB);
    dense_x3r<NT, KH16, 0>(ws, src, accB, ring, tid);
#pragma unroll
    for (int nt = 0; nt < NT; ++nt) accA[nt] = accB[nt];
  }
  {
    X3rMaskSrc<NT> src{accA, a.rs + mcrow, a.gar + mrow, hi, valid, kcs};     // G(a_0), then the feature rows of W_0^T
    zero_tiles<NT>(accB);
    dense_x3r<FT, KH16, 0>(ws, src, accB, ring, tid);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    store_tile<FT>(a.fbar + mc * F, hi, valid, accB);
  }
}

}  // namespace

void i2sdf_launch_igrad3r(const SdfTrainFwdArgs& a, unsigned grid, hipStream_t st) {
  launch_lds_bytes(LDS_BYTES_R, sdf_igrad3r_kernel<256, 6>, grid, st, a);
}
void i2sdf_launch_sdf_bwd3r(const SdfBwdArgs& a, unsigned grid, hipStream_t st) {
  launch_lds_bytes(LDS_BYTES_R, sdf_bwd3r_sweep1_kernel<256, 6>, grid, st, a);
  launch_lds_bytes(LDS_BYTES_R, sdf_bwd3r_sweep2_kernel<256, 256, 6>, grid, st, a);
}
void i2sdf_launch_rgb_bwd3r(const RgbBwdArgs& a, unsigned grid, hipStream_t st) {
  launch_lds_bytes(LDS_BYTES_R, rgb_bwd3r_kernel<256, 256>, grid, st, a);
}
