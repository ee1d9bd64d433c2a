// The bf16x3 kernels that run on 16-point waves (x3h.h), two waves per SIMD: the sdf-only forward (sampler passes, grid queries), the SDF
// forward with saves, the radiance net forward and backward -- the families that measured 4-8 % faster than their 32-point-wave
// predecessors (round 4, profiles/r4_wave16_experiments.txt; the d sdf/dx chain and the backward sweeps did not and stay in mlp_x3.hip).
// Same saved-tensor layouts as the 32-point kernels, so both families mix along a chain.  Built with the lifted unroll cap (build.sh).
#define I2SDF_RELU_ASM 1      // common.h: relu0
#include "mlp_args.h"
#include "x3h.h"

using namespace i2sdf;

namespace {

template <int LF>
__device__ __forceinline__ void pe_select_h(float px, float py, float pz, float (&pe)[cdiv(PE<LF>::DIM, 32) * 8], int kg) {
  constexpr int PED = PE<LF>::DIM, PE32 = cdiv(PED, 32);
  float full[PE<LF>::PEC * 8], pad[PE32 * 32];
  pe_full<LF>(px, py, pz, full);
#pragma unroll
  for (int i = 0; i < PE32 * 32; ++i) pad[i] = (i < PED) ? full[i < PE<LF>::PEC * 8 ? i : 0] : 0.f;
  x3h_select<PE32>(pad, pe, kg);
}

// sdf-only forward (sampler passes, grid queries) of the 256-wide net; 64-wide nets: sdf_fwd3_kernel, mlp_x3.hip
// PL = 2: the sampler's passes with two split planes per operand (x3h.h: dense_x3h; its own weight stream of two planes)
template <int H, int LF, int NW, int PL>
__global__ __launch_bounds__(NW * 64, 2) void sdf_fwd3h_kernel(const float* __restrict__ stream, int n_stages, int L, int skip, PointSpec ps,
                                                                const int* __restrict__ skip_flag, int64_t M, float* __restrict__ sdf_out) {
  constexpr int NT = H / 16, KH32 = H / 32, PE32 = cdiv(PE<LF>::DIM, 32), NPE = PE32 * 8;
  extern __shared__ __attribute__((aligned(16))) float lds[];
  if (skip_flag != nullptr && skip_flag[0] != 0) return;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, kg = lane >> 4;
  const int64_t m = ((int64_t)blockIdx.x * NW + wave) * HP + (lane & 15);
  const bool valid = m < M;
  const int64_t mc = valid ? m : M - 1;
  float px, py, pz;
  fetch_point(ps, mc, px, py, pz);
  float pe[NPE];
  pe_select_h<LF>(px, py, pz, pe, kg);
  WStreamH<NW> ws;
  ws.begin(stream, lds, n_stages, tid);
  f32x4 accP[NT], accN[NT];
  {
    XhFwdSrc<NT, 0, NPE, false> src{accN, pe, nullptr, kg, valid};
    dense_x3h<NT, PE32, 1, NW, XhFwdSrc<NT, 0, NPE, false>, PL>(ws, src, accP, tid);
  }
  for (int l = 1; l < L - 1; ++l) {
    XhFwdSrc<NT, KH32, NPE, false> src{accP, pe, nullptr, kg, valid};
    if (l == skip) dense_x3h<NT, KH32 + PE32, 1, NW, XhFwdSrc<NT, KH32, NPE, false>, PL>(ws, src, accN, tid);
    else dense_x3h<NT, KH32, 1, NW, XhFwdSrc<NT, KH32, NPE, false>, PL>(ws, src, accN, tid);
#pragma unroll
    for (int nt = 0; nt < NT; ++nt) accP[nt] = accN[nt];
  }
  float h[NT * 4];
#pragma unroll
  for (int nt = 0; nt < NT; ++nt)
#pragma unroll
    for (int r = 0; r < 4; ++r) h[nt * 4 + r] = softplus100(accP[nt][r]);
  float s[1];
  rowvec_h<1, NT, NW>(ws, h, s, tid);
  if (valid && kg == 0) sdf_out[m] = s[0];
}


// this lane's PE-space values (natural index k) -> the point-major (PEC*8)-float row of a saved PE / G(pbar) tensor: lane (p, kg) holds
// sel[8c + 4e + t] = x[32c + 16e + 4kg + t] and writes the f32x4 at k = 16 j + 4 kg for j = 2c + e
template <int PEC, int NSEL>
__device__ __forceinline__ void store_pe_row_h(float* __restrict__ row, int kg, bool valid, const float (&sel)[NSEL]) {
  if (!valid) return;
#pragma unroll
  for (int j = 0; j < cdiv(PEC * 8, 16); ++j)
    if (16 * j + 4 * kg < PEC * 8)
      *reinterpret_cast<f32x4*>(row + 16 * j + 4 * kg) = f32x4{sel[8 * (j >> 1) + 4 * (j & 1)], sel[8 * (j >> 1) + 4 * (j & 1) + 1],
                                                               sel[8 * (j >> 1) + 4 * (j & 1) + 2], sel[8 * (j >> 1) + 4 * (j & 1) + 3]};
}

// SDF forward with saves: hidden activations -> hs, sdf, feature rows (the d sdf/dx chain is sdf_igrad3_kernel, mlp_x3.hip)
// (its saved-tensor stores stay predicated and uncounted by the stage waits, x3.h: as unconditional instructions they cost 8 more registers
// at the 256-register cap of a two-waves-per-SIMD kernel -- 132 B of scratch, 310 -> 336 us per launch, profiles/r5_mfma_paced.txt)
template <int H, int F, int LF, int NW>
__global__ __launch_bounds__(NW * 64, 2) void sdf_train_fwd3h_kernel(SdfTrainFwdArgs a) {
  constexpr int NT = H / 16, KH32 = H / 32, PEC = PE<LF>::PEC, PE32 = cdiv(PE<LF>::DIM, 32), NPE = PE32 * 8, FT = F / 16;
  static_assert(NT == FT, "feature tiles reuse the hidden accumulator set");
  extern __shared__ __attribute__((aligned(16))) float lds[];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, kg = lane >> 4;
  const int64_t m = ((int64_t)(blockIdx.x + a.wg0) * NW + wave) * HP + (lane & 15);
  const bool valid = m < a.M;
  const int64_t mc = valid ? m : a.M - 1;
  float px, py, pz;
  fetch_point(a.pts, mc, px, py, pz);
  float pe[NPE];
  pe_select_h<LF>(px, py, pz, pe, kg);
  if (a.pe_save) store_pe_row_h<PEC>(a.pe_save + m * (PEC * 8), kg, valid, pe);
  const int64_t lstride = a.Mp * H;
  const int kcs = a.kcs;
  const int64_t mrow = save_row_off(m, kcs);
  WStreamH<NW> ws;
  ws.begin(a.fwd, lds, a.n_fwd, tid);
  f32x4 accA[NT], accB[NT];
  {
    XhFwdSrc<NT, 0, NPE> src{accB, pe, nullptr, kg, valid};
    dense_x3h<NT, PE32, 1, NW>(ws, src, accA, tid);
  }
  for (int l = 1; l < a.L - 1; ++l) {
    float* hrow = a.hs ? a.hs + (l - 1) * lstride + mrow : nullptr;
    XhFwdSrc<NT, KH32, NPE> src{accA, pe, hrow, kg, valid, kcs};
    if (l == a.skip) dense_x3h<NT, KH32 + PE32, 1, NW>(ws, src, accB, tid);
    else dense_x3h<NT, KH32, 1, NW>(ws, src, accB, tid);
#pragma unroll
    for (int nt = 0; nt < NT; ++nt) accA[nt] = accB[nt];
  }
  float h[NT * 4];                     // h_{L-1} in D-layout order: sdf row, feature op, top of the reverse chain
#pragma unroll
  for (int nt = 0; nt < NT; ++nt)
#pragma unroll
    for (int r = 0; r < 4; ++r) h[nt * 4 + r] = softplus100(accA[nt][r]);
  if (a.hs) store_regs_h<NT>(a.hs + (a.L - 2) * lstride + mrow, kg, valid, h, kcs);
  {
    float s[1];
    rowvec_h<1, NT, NW>(ws, h, s, tid);
    if (a.sdf != nullptr && valid && kg == 0) a.sdf[m] = s[0];
  }
  if (a.feat != nullptr) {
    XhRegSrc<NT * 4> src{h};
    dense_x3h<FT, KH32, 1, NW>(ws, src, accB, tid);
    store_tile_h<FT>(a.feat + mc * (a.ldf ? a.ldf : (int64_t)F), kg, valid, accB);
  }
}

// radiance net (RenderingNetwork, 'nerf' mode: mlp.py:208-229) forward and backward
template <int H, int F, int LFV, int NW>
__global__ __launch_bounds__(NW * 64, 2) void rgb_fwd3h_kernel(RgbFwdArgs a) {
  constexpr int NT = H / 16, KH32 = H / 32, PECV = PE<LFV>::PEC, PV32 = cdiv(PE<LFV>::DIM, 32);
  extern __shared__ __attribute__((aligned(16))) float lds[];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, kg = lane >> 4;
  const int64_t m = ((int64_t)(blockIdx.x + a.wg0) * NW + wave) * HP + (lane & 15);
  const bool valid = m < a.M;
  const int64_t mc = valid ? m : a.M - 1;
  const int64_t ray = mc / a.n_per_ray;
  float pev[PV32 * 8];
  pe_select_h<LFV>(a.dirs[ray * 3 + 0], a.dirs[ray * 3 + 1], a.dirs[ray * 3 + 2], pev, kg);
  if (a.pev_save) store_pe_row_h<PECV>(a.pev_save + m * (PECV * 8), kg, valid, pev);
  const int64_t lstride = a.Mp * H;
  const int kcs = a.kcs;
  const int64_t mrow = save_row_off(m, kcs);
  WStreamH<NW> ws;
  ws.begin(a.fwd, lds, a.n_fwd, tid);
  f32x4 accA[NT], accB[NT];
  {
    XhPeRowSrc<PV32> src{pev, a.feat + mc * F, kg};
    dense_x3h<NT, PV32 + F / 32, 1, NW>(ws, src, accA, tid);
  }
  for (int l = 1; l < a.L - 1; ++l) {
    XhReluSrc<NT> src{accA, a.rs ? a.rs + (l - 1) * lstride + mrow : nullptr, kg, valid, kcs};
    dense_x3h<NT, KH32, 1, NW>(ws, src, accB, tid);
#pragma unroll
    for (int nt = 0; nt < NT; ++nt) accA[nt] = accB[nt];
  }
  float r[NT * 4];
#pragma unroll
  for (int nt = 0; nt < NT; ++nt)
#pragma unroll
    for (int q = 0; q < 4; ++q) r[nt * 4 + q] = fmaxf(accA[nt][q], 0.f);
  if (a.rs) store_regs_h<NT>(a.rs + (a.L - 2) * lstride + mrow, kg, valid, r, kcs);
  float o[3];
  rowvec_h<3, NT, NW>(ws, r, o, tid);
  if (valid && kg == 0) {
#pragma unroll
    for (int i = 0; i < 3; ++i) a.rgb[m * 3 + i] = 1.0f / (1.0f + expf(-o[i]));
  }
}

// light-mask head (model/network/__init__.py:29-32,162-170; fp32-MFMA twin: mlp_train.hip light_fwd_kernel): HL hidden units from F features
template <int HL, int F, int NW>
__global__ __launch_bounds__(NW * 64, 2) void light_fwd3h_kernel(LightFwd3hArgs a) {
  constexpr int NT = HL / 16, KF32 = F / 32;
  extern __shared__ __attribute__((aligned(16))) float lds[];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, kg = lane >> 4;
  const int64_t m = ((int64_t)blockIdx.x * NW + wave) * HP + (lane & 15);
  const bool valid = m < a.M;
  const int64_t mc = valid ? m : a.M - 1;
  WStreamH<NW> ws;
  ws.begin(a.fwd, lds, a.n_fwd, tid);
  f32x4 acc[NT];
  {
    XhRowSrc<true> src{a.feat + mc * F, kg};
    dense_x3h<NT, KF32, 1, NW>(ws, src, acc, tid);
  }
  float h[NT * 4];
#pragma unroll
  for (int nt = 0; nt < NT; ++nt)
#pragma unroll
    for (int r = 0; r < 4; ++r) h[nt * 4 + r] = softplus100(acc[nt][r]);
  if (a.hl) store_regs_h<NT>(a.hl + m * HL, kg, valid, h, 16);
  float o[1];
  rowvec_h<1, NT, NW>(ws, h, o, tid);
  if (valid && kg == 0) a.lm[m] = 1.0f / (1.0f + expf(-o[0]));
}

template <int H, int F, int NW>
__global__ __launch_bounds__(NW * 64, 2) void rgb_bwd3h_kernel(RgbBwdArgs a) {
  constexpr int NT = H / 16, KH32 = H / 32, FT = F / 16;
  static_assert(FT == NT, "feature tiles reuse the hidden accumulator set");
  extern __shared__ __attribute__((aligned(16))) float lds[];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, kg = lane >> 4;
  const int64_t m = ((int64_t)(blockIdx.x + a.wg0) * NW + wave) * HP + (lane & 15);
  const bool valid = m < a.M;
  const int64_t mc = valid ? m : a.M - 1;
  const int64_t lstride = a.Mp * H;
  const int kcs = a.kcs;
  const int64_t mrow = save_row_off(m, kcs), mcrow = save_row_off(mc, kcs);
  float g3[3];
#pragma unroll
  for (int j = 0; j < 3; ++j) {
    const float c = a.rgb[mc * 3 + j];
    g3[j] = a.rgb_bar[mc * 3 + j] * c * (1.0f - c);
  }
  if (valid && kg == 0) *reinterpret_cast<f32x4*>(a.ga_last + m * 4) = f32x4{g3[0], g3[1], g3[2], 0.f};
  WStreamH<NW> ws;
  ws.begin(a.rev, lds, a.n_rev, tid);
  float ga[NT * 4];
  {
    // G(r_{L-1}) = W_last^T G(a_last): three row vectors, then the top mask r_{L-1} > 0 -> G(a_{L-2})
    constexpr int NWC = 3 * NT, TOT = rowvec_h_chunks(NT, 3), NS = TOT / SCH;
#pragma unroll
    for (int i = 0; i < NT * 4; ++i) ga[i] = 0.f;
#pragma unroll
    for (int s = 0; s < NS; ++s) {
      const f32x4* cur = reinterpret_cast<const f32x4*>(ws.advance(tid)) + lane;
#pragma unroll
      for (int j = 0; j < SCH; ++j) {
        const int c = s * SCH + j;
        if (c < NWC) {
          const int row = c / NT, nt = c % NT;
          const f32x4 w = cur[j * 64];
          ga[nt * 4 + 0] = fmaf(w.x, g3[row], ga[nt * 4 + 0]);
          ga[nt * 4 + 1] = fmaf(w.y, g3[row], ga[nt * 4 + 1]);
          ga[nt * 4 + 2] = fmaf(w.z, g3[row], ga[nt * 4 + 2]);
          ga[nt * 4 + 3] = fmaf(w.w, g3[row], ga[nt * 4 + 3]);
        }
      }
    }
    const int l = a.L - 2;
    const float* rrow = a.rs + l * lstride + mcrow;
    float* grow = a.gar + l * lstride + mrow;
#pragma unroll
    for (int nt = 0; nt < NT; ++nt) {
      const f32x4 rv = *reinterpret_cast<const f32x4*>(rrow + nt * kcs + 4 * kg);
      f32x4 o;
#pragma unroll
      for (int t = 0; t < 4; ++t) { o[t] = rv[t] > 0.f ? ga[nt * 4 + t] : 0.f; ga[nt * 4 + t] = o[t]; }
      if (valid) *reinterpret_cast<f32x4*>(grow + nt * kcs + 4 * kg) = o;
    }
  }
  f32x4 accA[NT], accB[NT];
  auto zero = [&](f32x4 (&x)[NT]) __attribute__((always_inline)) {
#pragma unroll
    for (int nt = 0; nt < NT; ++nt) x[nt] = f32x4{0.f, 0.f, 0.f, 0.f};
  };
  {
    XhRegSrc<NT * 4> src{ga};
    zero(accA);
    dense_x3h<NT, KH32, 0, NW>(ws, src, accA, tid);            // W_{L-2}^T G(a_{L-2})
  }
  for (int l = a.L - 3; l >= 1; --l) {
    XhMaskSrc<NT> src{accA, a.rs + l * lstride + mcrow, a.gar + l * lstride + mrow, kg, valid, kcs};
    zero(accB);
    dense_x3h<NT, KH32, 0, NW>(ws, src, accB, tid);
#pragma unroll
    for (int nt = 0; nt < NT; ++nt) accA[nt] = accB[nt];
  }
  {
    XhMaskSrc<NT> src{accA, a.rs + mcrow, a.gar + mrow, kg, valid, kcs};     // G(a_0), then the feature rows of W_0^T
    zero(accB);
    dense_x3h<FT, KH32, 0, NW>(ws, src, accB, tid);
    store_tile_h<FT>(a.fbar + mc * F, kg, valid, accB);
  }
}

}  // namespace

// launches over workgroups of 128 points (eight 16-point waves)
void i2sdf_launch_sdf_fwd3h(const float* stream, int n_stages, int L, int skip, const PointSpec& ps, const int* skip_flag, int64_t M, float* sdf_out,
                            int planes, hipStream_t st) {
  const unsigned grid = (unsigned)((M + 8 * HP - 1) / (8 * HP));
  if (planes == 2) launch_lds_threads(512, sdf_fwd3h_kernel<256, 6, 8, 2>, grid, st, stream, n_stages, L, skip, ps, skip_flag, M, sdf_out);
  else launch_lds_threads(512, sdf_fwd3h_kernel<256, 6, 8, 3>, grid, st, stream, n_stages, L, skip, ps, skip_flag, M, sdf_out);
}
void i2sdf_launch_train_fwd3h(const SdfTrainFwdArgs& a, unsigned grid, hipStream_t st) { launch_lds_threads(512, sdf_train_fwd3h_kernel<256, 256, 6, 8>, grid, st, a); }
void i2sdf_launch_rgb_fwd3h(const RgbFwdArgs& a, unsigned grid, hipStream_t st) { launch_lds_threads(512, rgb_fwd3h_kernel<256, 256, 4, 8>, grid, st, a); }
void i2sdf_launch_light_fwd3h(const LightFwd3hArgs& a, unsigned grid, hipStream_t st) { launch_lds_threads(8 * 64, light_fwd3h_kernel<128, 256, 8>, grid, st, a); }
void i2sdf_launch_rgb_bwd3h(const RgbBwdArgs& a, unsigned grid, hipStream_t st) { launch_lds_threads(512, rgb_bwd3h_kernel<256, 256, 8>, grid, st, a); }
