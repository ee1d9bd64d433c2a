// Pieces shared by the training-mode MLP kernels (forward-with-saves, igrad chain, backward sweeps).
#pragma once
#include "plan.h"

#include <mutex>
#include <unordered_set>

namespace i2sdf {

// launch a 256-thread kernel with the double-buffered weight stage in dynamic LDS (> 64 KB needs the attribute once)
template <class K, class... A>
inline void launch_lds_bytes(int lds_bytes, K kern, unsigned grid, hipStream_t st, A... args) {
  static std::mutex mu;
  static std::unordered_set<const void*> done;
  {
    std::lock_guard<std::mutex> lk(mu);
    if (!done.count((const void*)kern)) {
      (void)hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, lds_bytes);
      done.insert((const void*)kern);
    }
  }
  kern<<<grid, 256, lds_bytes, st>>>(args...);
}
template <class K, class... A>
inline void launch_lds(K kern, unsigned grid, hipStream_t st, A... args) { launch_lds_bytes(LDS_BYTES, kern, grid, st, args...); }
// the same for a kernel of `threads` threads per workgroup (16-point-wave kernels, x3h.h: 512 = 8 waves, 256 = 4 waves)
template <class K, class... A>
inline void launch_lds_threads(int threads, K kern, unsigned grid, hipStream_t st, A... args) {
  static std::mutex mu;
  static std::unordered_set<const void*> done;
  {
    std::lock_guard<std::mutex> lk(mu);
    if (!done.count((const void*)kern)) {
      (void)hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, LDS_BYTES_H);
      done.insert((const void*)kern);
    }
  }
  kern<<<grid, threads, LDS_BYTES_H, st>>>(args...);
}

// Split of a launch over M points into full rounds of 128-point workgroups (one per CU) and a short tail that is run by
// the split-K kernels (ksplit.h): returns the number of points of the bulk part (0 = no split).
inline int64_t split_bulk_points(int64_t M, int n_cu) {
  const int64_t n_wg = (M + PTS_PER_WG - 1) / PTS_PER_WG;
  const int64_t full = (n_wg / n_cu) * n_cu, rem = n_wg - full;
  if (full == 0 || rem == 0 || rem * 4 > n_cu) return 0;      // nothing to gain, or the tail would not fit one round
  // the bulk is also where the blocked layout of the saved tensors ends, and the weight-gradient kernels pick an operand's layout once
  // per split-M chunk: a chunk must not straddle that boundary (n_cu * 128 is a multiple of the 2048-point chunk only when n_cu % 16 == 0
  // -- true for the 256 CUs of an MI355X, not for every partition of it).  No split then: every point goes through the full
  // workgroups, every saved tensor is blocked throughout.
  if ((full * PTS_PER_WG) % I2SDF_WG_CH != 0) return 0;
  return full * PTS_PER_WG;
}

// ---- layouts of the saved per-point tensors of width 256 (hs, abars, gus, gas; rs, gar) ------------------------------
//   point-major  [Mp][256]                          row of point m at m*256, k-chunk kc (16 floats) at +16*kc
//   blocked      [Mp/32][16 k-chunks][32 points][16] row of point m at (m/32)*8192 + (m%32)*16, k-chunk kc at +512*kc
// The K-outer bf16x3 kernels touch 16 floats of every point per k-chunk: in the point-major layout a wave instruction
// (16 B per lane) lands in 32 different 1 KB rows, in the blocked layout in one contiguous 2 KB run -- 3.9 vs 5.0 TB/s at the
// occupancy of these kernels (a pure access-pattern microbenchmark, round 2).  I2SDF_OPT_BLOCKED_SAVES stores the points handled by the bf16x3
// full workgroups blocked; the fp32 split-K tail workgroups (the points behind them) and the fp32 kernels keep point-major rows.
// Both ranges are tile aligned, so one tensor holds both; every producer / consumer derives the blocked prefix from (plan, M).
constexpr int KCS_PM = 16, KCS_BLK = 512;
__host__ __device__ inline int64_t save_row_off(int64_t m, int kcs) { return kcs == KCS_PM ? m * 256 : (m >> 5) * 8192 + (m & 31) * 16; }
// packed 24-bit records (I2SDF_OPT_SAVES24; x3.h: p24_load8 / p24_store8, wgrad.hip: the p24 loaders): abars, gus, gas of the points that go through the
// bf16x3 kernels when EVERY point does (point ranges on).  A 32-point block of a layer is 16 k-chunks of 1536 B: [32 points][2 lanes][16 B] upper halves,
// then [32 points][2 lanes][8 B] mid bytes -- 6144 floats, dense; the layer stride stays Mp * 256 floats (the tensors keep their fp32 allocation).
// (constants and p24_row_off: x3.h)

inline bool sdf_x3_path(const i2sdf_plan* p) {
  const i2sdf_mlp_desc& d = p->sdf.d;
  return p->train_fwd_bf16x3 != 0 && p->sdf_bwd_bf16x3 != 0 && p->H == 256 && p->F == 256 && d.n_lin >= 4 && d.skip_layer != d.n_lin - 2;
}
// leading points of a batch of M points whose saved SDF tensors are in the blocked layout (0 = none)
inline int64_t sdf_blocked_points(const i2sdf_plan* p, int64_t M, int64_t Mp, bool has_feat = true) {
  if (!p->blocked_saves || !sdf_x3_path(p)) return 0;
  if (i2sdf_parts_on(p)) return Mp;                           // point ranges: no split-K tail, everything blocked
  const int64_t bulk = split_bulk_points(M, p->n_cu);
  return (bulk > 0 && has_feat) ? bulk : Mp;
}
// abars / gus / gas as packed 24-bit records?  Needs every producer and consumer on the blocked bf16x3 path for EVERY point (point ranges: no split-K
// tail) and the two-plane weight-gradient GEMMs (which keep 16 significant bits per operand anyway); the fp32-equivalent mode keeps fp32 storage.
inline bool sdf_saves24(const i2sdf_plan* p) {
  return p->saves24 != 0 && p->wgrad_bf16x3 != 0 && p->wgrad_bf16x2 != 0 && p->blocked_saves != 0 && sdf_x3_path(p) && i2sdf_parts_on(p);
}
inline int64_t rgb_blocked_points(const i2sdf_plan* p, int64_t M, int64_t Mp) {
  if (!p->blocked_saves || !p->rgb_bf16x3 || p->rgb.d.hidden != 256 || p->F != 256) return 0;
  if (i2sdf_parts_on(p)) return Mp;
  const int64_t bulk = split_bulk_points(M, p->n_cu);
  return bulk > 0 ? bulk : Mp;
}

constexpr float RS2 = 0.70710678118654752440f;

template <int N>
__device__ __forceinline__ void softplus_tiles(const f32x16 (&acc)[N], float (&h)[N * 16]) {
#pragma unroll
  for (int nt = 0; nt < N; ++nt)
#pragma unroll
    for (int r = 0; r < 16; ++r) h[nt * 16 + r] = softplus100(acc[nt][r]);
}

// stage counts -- must mirror plan.cpp
__host__ __device__ constexpr int sdf_fwd_stages(int H, int F, int PEC, int L, bool has_skip, bool full) {
  int c = op_chunks(H / 32, PEC);
  for (int l = 1; l < L - 1; ++l) c += op_chunks(H / 32, H / 8);
  if (has_skip) c += op_chunks(H / 32, H / 8 + PEC) - op_chunks(H / 32, H / 8);
  c += rowvec_chunks(H / 8, 1);
  if (full) c += op_chunks(F / 32, H / 8);
  return c / SC;
}
__host__ __device__ constexpr int sdf_fwd3_stages(int H, int PED, int L, bool has_skip) {
  const int PE16 = cdiv(PED, 16);
  int c = x3_op_chunks(H / 32, PE16);
  for (int l = 1; l < L - 1; ++l) c += x3_op_chunks(H / 32, H / 16);
  if (has_skip) c += x3_op_chunks(H / 32, H / 16 + PE16) - x3_op_chunks(H / 32, H / 16);
  c += rowvec_chunks(H / 8, 1);
  return c / SC;
}
__host__ __device__ constexpr int x3_bwd_chunks(int KT, int KC16) { return round_up(KC16 * KT * 3, SC); }
__host__ __device__ constexpr int sdf_rev3_stages(int H, int PEC, int L, bool has_skip) {
  const int PT = cdiv(PEC * 8, 32);
  int c = rowvec_chunks(H / 8, 1);
  for (int l = L - 2; l >= 1; --l) c += x3_bwd_chunks(H / 32, H / 16);
  if (has_skip) c += x3_bwd_chunks(PT, H / 16);
  c += x3_bwd_chunks(PT, H / 16);
  return c / SC;
}
// backward sweep 1 walks the hidden layers of the bf16x3 forward stream; sweep 2 the reverse stream from its start down to W_1^T
__host__ __device__ constexpr int sdf_fwd3_hidden_stages(int H, int PED, int L, bool has_skip) {
  return sdf_fwd3_stages(H, PED, L, has_skip) - rowvec_chunks(H / 8, 1) / SC;
}
__host__ __device__ constexpr int sdf_rev3_bwd_stages(int H, int F, int PEC, int L, bool has_skip) {
  const int PT = cdiv(PEC * 8, 32);
  int c = 2 * rowvec_chunks(H / 8, 1) + x3_bwd_chunks(H / 32, F / 16);
  for (int l = L - 2; l >= 1; --l) c += x3_bwd_chunks(H / 32, H / 16);
  if (has_skip) c += x3_bwd_chunks(PT, H / 16);
  return c / SC;
}
// ---- 16-point-wave family (x3h.h): same ops, 16-row tiles / 32-wide k-chunks -- must mirror plan.cpp
__host__ __device__ constexpr int sdf_fwd3h_stages(int H, int PED, int L, bool has_skip, int PL = 3) {
  const int PE32 = cdiv(PED, 32);
  int c = x3h_op_chunks(H / 16, PE32, PL);
  for (int l = 1; l < L - 1; ++l) c += x3h_op_chunks(H / 16, H / 32, PL);
  if (has_skip) c += x3h_op_chunks(H / 16, H / 32 + PE32, PL) - x3h_op_chunks(H / 16, H / 32, PL);
  c += rowvec_h_chunks(H / 16, 1);
  return c / SCH;
}
__host__ __device__ constexpr int sdf_fwd3h_train_stages(int H, int F, int PED, int L, bool has_skip, bool full) {
  return sdf_fwd3h_stages(H, PED, L, has_skip) + (full ? x3h_op_chunks(F / 16, H / 32) / SCH : 0);
}
__host__ __device__ constexpr int rgb_fwd3h_stages(int H, int F, int PEDV, int L) {
  return (x3h_op_chunks(H / 16, cdiv(PEDV, 32) + F / 32) + (L - 2) * x3h_op_chunks(H / 16, H / 32) + rowvec_h_chunks(H / 16, 3)) / SCH;
}
__host__ __device__ constexpr int rgb_rev3h_stages(int H, int F, int L) {
  return (rowvec_h_chunks(H / 16, 3) + (L - 2) * x3h_bwd_chunks(H / 16, H / 32) + x3h_bwd_chunks(F / 16, H / 32)) / SCH;
}
__host__ __device__ constexpr int bwd_op_chunks(int KT, int NC) { return round_up(KT * NC, SC); }
// reverse stream from the w_sdf row vector to W_0^T (the d sdf/dx chain); PT = tiles of the PE space
__host__ __device__ constexpr int sdf_rev_stages(int H, int PEC, int L, bool has_skip) {
  const int PT = cdiv(PEC * 8, 32);
  int c = rowvec_chunks(H / 8, 1);
  for (int l = L - 2; l >= 1; --l) c += bwd_op_chunks(H / 32, H / 8);
  if (has_skip) c += bwd_op_chunks(H / 32 + PT, H / 8) - bwd_op_chunks(H / 32, H / 8);
  c += bwd_op_chunks(PT, H / 8);
  return c / SC;
}
__host__ __device__ constexpr int sdf_rev_feat_stages(int H, int F) { return bwd_op_chunks(H / 32, F / 8) / SC; }

// transposed dense op without bias: stream layout [KT*NC weight chunks] padded to stages
//   MODE 1: acc = W^T in     MODE 2: acc += W^T in
template <int KT, int NC, int MODE>
__device__ __forceinline__ void dense_op_nobias(WStream& ws, const float (&in)[NC * 4], f32x16 (&acc)[KT], int tid) {
  constexpr int NW = KT * NC, TOT = round_up(NW, SC), NS = TOT / SC;
  const int lane = tid & 63;
  if (MODE == 1) {
#pragma unroll
    for (int kt = 0; kt < KT; ++kt)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[kt][r] = 0.f;
  }
#pragma unroll
  for (int s = 0; s < NS; ++s) {
    const f32x4* cur = reinterpret_cast<const f32x4*>(ws.advance(tid)) + lane;
    const int j1 = (NW - s * SC < SC) ? NW - s * SC : SC;
    f32x4 ab[PF];
#pragma unroll
    for (int i = 0; i < PF; ++i)
      if (i < j1) ab[i] = cur[i * 64];
#pragma unroll
    for (int j = 0; j < SC; ++j) {
      if (j < j1) {
        const int w = s * SC + j;
        const int kt = w / NC, nc = w % NC;
        f32x4 a = ab[j % PF];
        if (j + PF < j1) ab[j % PF] = cur[(j + PF) * 64];
        acc[kt] = mfma(a.x, in[nc * 4 + 0], acc[kt]);
        acc[kt] = mfma(a.y, in[nc * 4 + 1], acc[kt]);
        acc[kt] = mfma(a.z, in[nc * 4 + 2], acc[kt]);
        acc[kt] = mfma(a.w, in[nc * 4 + 3], acc[kt]);
      }
    }
#pragma unroll
    for (int i = 0; i < PF; ++i)
      if (i < j1) I2SDF_SGB(I2SDF_MASK_DSREAD, 1);
#pragma unroll
    for (int j = 0; j < SC; ++j) {
      if (j < j1) {
        I2SDF_SGB(I2SDF_MASK_MFMA, 1);
        if (j + PF < j1) I2SDF_SGB(I2SDF_MASK_DSREAD, 1);
        I2SDF_SGB(I2SDF_MASK_MFMA, 3);
      }
    }
  }
}

// read a row vector (w_row[k] in B layout) from a rowvec op: regs[kc*4+t] = w[8kc+4hi+t]; also returns the scalar chunk
template <int KC>
__device__ __forceinline__ void rowvec_load(WStream& ws, float (&w)[KC * 4], f32x4& scalars, int tid) {
  constexpr int TOT = rowvec_chunks(KC, 1), NS = TOT / SC;
  const int lane = tid & 63;
#pragma unroll
  for (int s = 0; s < NS; ++s) {
    const f32x4* cur = reinterpret_cast<const f32x4*>(ws.advance(tid)) + lane;
#pragma unroll
    for (int j = 0; j < SC; ++j) {
      const int c = s * SC + j;
      if (c < KC) {
        f32x4 v = cur[j * 64];
        w[c * 4 + 0] = v.x; w[c * 4 + 1] = v.y; w[c * 4 + 2] = v.z; w[c * 4 + 3] = v.w;
      } else if (c == KC) {
        scalars = cur[j * 64];
      }
    }
  }
}

// point m of a batch laid out as [n_ray_pts points generated from rays | explicit points]:
//   m <  n_ray_pts : x = cam[r] + z[r*ldz + j] * dirs[r], r = m / n_per_ray, j = m % n_per_ray  (mul then add, two
//                    roundings, as `cam_loc + z_vals * ray_dirs` does, model/network/__init__.py:103)
//   m >= n_ray_pts : x = points[m - n_ray_pts]
struct PointSpec {
  const float* points; const float* cam; const float* dirs; const float* z;
  int64_t ldz; int64_t n_ray_pts; int32_t n_per_ray;
};
__device__ __forceinline__ void fetch_point(const PointSpec& a, int64_t m, float& x, float& y, float& z) {
  if (m < a.n_ray_pts) {
    const int64_t ray = m / a.n_per_ray;
    const int j = (int)(m - ray * a.n_per_ray);
    const float t = a.z[ray * a.ldz + j];
    x = __fadd_rn(a.cam[ray * 3 + 0], __fmul_rn(t, a.dirs[ray * 3 + 0]));
    y = __fadd_rn(a.cam[ray * 3 + 1], __fmul_rn(t, a.dirs[ray * 3 + 1]));
    z = __fadd_rn(a.cam[ray * 3 + 2], __fmul_rn(t, a.dirs[ray * 3 + 2]));
  } else {
    const int64_t i = m - a.n_ray_pts;
    x = a.points[i * 3 + 0]; y = a.points[i * 3 + 1]; z = a.points[i * 3 + 2];
  }
}

// ---- positional-encoding Jacobian helpers -----------------------------------------------------------
// coefficient d PE_k / d x_axis(k) for every k of the padded PE space, from the PE values themselves:
//   identity: 1 ; sin(f x): f cos(f x) = f * PE[k+3] ; cos(f x): -f sin(f x) = -f * PE[k-3]
template <int LF>
__device__ __forceinline__ void pe_coef(const float (&full)[PE<LF>::PEC * 8], float (&coef)[PE<LF>::PEC * 8]) {
#pragma unroll
  for (int k = 0; k < PE<LF>::PEC * 8; ++k) {
    if (k < 3) coef[k] = 1.f;
    else if (k < PE<LF>::DIM) {
      const int kk = k - 3, j = kk / 6, w = kk % 6;
      const float f = (float)(1 << j);
      coef[k] = (w < 3) ? f * full[k + 3] : -f * full[k - 3];
    } else coef[k] = 0.f;
  }
}
__host__ __device__ constexpr int pe_axis(int k) { return k < 3 ? k : (k - 3) % 3; }

// n = J^T pbar, pbar given as D-layout tiles over the padded PE space (PT tiles)
template <int LF, int PT>
__device__ __forceinline__ void pe_jt_apply(const float (&coef)[PE<LF>::PEC * 8], const f32x16 (&pbar)[PT], int hi, float (&n)[3]) {
  float a0[3] = {0.f, 0.f, 0.f}, a1[3] = {0.f, 0.f, 0.f};
#pragma unroll
  for (int kt = 0; kt < PT; ++kt)
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int k0 = 32 * kt + (r & 3) + 8 * (r >> 2), k1 = k0 + 4;
      if (k0 < PE<LF>::DIM) a0[pe_axis(k0)] = fmaf(coef[k0], pbar[kt][r], a0[pe_axis(k0)]);
      if (k1 < PE<LF>::DIM) a1[pe_axis(k1)] = fmaf(coef[k1], pbar[kt][r], a1[pe_axis(k1)]);
    }
#pragma unroll
  for (int i = 0; i < 3; ++i) {
    const float v = hi ? a1[i] : a0[i];
    n[i] = v + __shfl_xor(v, 32);
  }
}
// G(pbar) = J nbar in B layout (chunk regs over the padded PE space)
template <int LF>
__device__ __forceinline__ void pe_j_apply(const float (&coef)[PE<LF>::PEC * 8], const float (&nb)[3], int hi,
                                           float (&out)[PE<LF>::PEC * 4]) {
#pragma unroll
  for (int c = 0; c < PE<LF>::PEC; ++c)
#pragma unroll
    for (int t = 0; t < 4; ++t) {
      const int k0 = 8 * c + t, k1 = k0 + 4;
      const float v0 = (k0 < PE<LF>::DIM) ? coef[k0] * nb[pe_axis(k0)] : 0.f;
      const float v1 = (k1 < PE<LF>::DIM) ? coef[k1] * nb[pe_axis(k1)] : 0.f;
      out[c * 4 + t] = hi ? v1 : v0;
    }
}

}  // namespace i2sdf
