// Per-tile epilogues for dense_op_epi (common.h): what happens to an output tile between two layers.
// Each works in place on the accumulator tile (D layout: lane = point, register r <-> feature 32*nt + (r&3) + 8*(r>>2) + 4*hi)
// and reads/writes the matching 16-float tile of point-major [m][H] rows in HBM.  With PRE = true the global loads of
// tile nt are issued one LDS stage before they are consumed (two register slots, alternating by tile parity: requires
// at most one tile to complete per stage, i.e. reduction length >= 32 chunks); with PRE = false they are issued in place.
#pragma once
#include "mlp_common.h"

namespace i2sdf {

__device__ __forceinline__ void load_tile16(const float* __restrict__ row, int nt, int hi, float (&v)[16]) {
#pragma unroll
  for (int q = 0; q < 4; ++q) {
    const f32x4 x = *reinterpret_cast<const f32x4*>(row + 32 * nt + 8 * q + 4 * hi);
    v[4 * q] = x.x; v[4 * q + 1] = x.y; v[4 * q + 2] = x.z; v[4 * q + 3] = x.w;
  }
}
__device__ __forceinline__ void store_tile16(float* __restrict__ row, int nt, int hi, bool valid, const f32x16& t) {
  if (!valid) return;
#pragma unroll
  for (int q = 0; q < 4; ++q)
    *reinterpret_cast<f32x4*>(row + 32 * nt + 8 * q + 4 * hi) = f32x4{t[4 * q], t[4 * q + 1], t[4 * q + 2], t[4 * q + 3]};
}
__device__ __forceinline__ void store_tile16(float* __restrict__ row, int nt, int hi, bool valid, const float (&t)[16]) {
  if (!valid) return;
#pragma unroll
  for (int q = 0; q < 4; ++q)
    *reinterpret_cast<f32x4*>(row + 32 * nt + 8 * q + 4 * hi) = f32x4{t[4 * q], t[4 * q + 1], t[4 * q + 2], t[4 * q + 3]};
}
template <int N>
__device__ __forceinline__ void commit_tiles(const f32x16 (&acc)[N], float (&h)[N * 16]) {
#pragma unroll
  for (int nt = 0; nt < N; ++nt)
#pragma unroll
    for (int r = 0; r < 16; ++r) h[nt * 16 + r] = acc[nt][r];
}

__device__ __forceinline__ void store_quad(float* __restrict__ row, int nt, int q, int hi, bool valid, const f32x16& t) {
  if (valid) *reinterpret_cast<f32x4*>(row + 32 * nt + 8 * q + 4 * hi) = f32x4{t[4 * q], t[4 * q + 1], t[4 * q + 2], t[4 * q + 3]};
}
__device__ __forceinline__ void store_quad(float* __restrict__ row, int nt, int q, int hi, bool valid, const float (&t)[16]) {
  if (valid) *reinterpret_cast<f32x4*>(row + 32 * nt + 8 * q + 4 * hi) = f32x4{t[4 * q], t[4 * q + 1], t[4 * q + 2], t[4 * q + 3]};
}
#define I2SDF_APPLY_FROM_ELEM                                             \
  __device__ __forceinline__ void apply(int nt, f32x16& acc) {            \
    _Pragma("unroll") for (int r = 0; r < 16; ++r) elem(nt, acc, r);      \
  }

// h = softplus100(a) [+ store h]                      (forward, SDF net / light head)
struct SoftplusEpi {
  float* row; int hi; bool valid;
  int own = -1;             // >= 0: store only the tiles this wave owns (nt/2 == own); split-K tail kernels
  __device__ __forceinline__ void prefetch(int) {}
  __device__ __forceinline__ void elem(int nt, f32x16& acc, int r) {
    acc[r] = softplus100(acc[r]);
    if (row && (r & 3) == 3) store_quad(row, nt, r >> 2, hi, valid && (own < 0 || (nt >> 1) == own), acc);
  }
  I2SDF_APPLY_FROM_ELEM
};
// r = max(a, 0) [+ store r]                            (forward, radiance net)
struct ReluEpi {
  float* row; int hi; bool valid;
  int own = -1;             // >= 0: store only the tiles this wave owns (nt/2 == own); split-K tail kernels
  __device__ __forceinline__ void prefetch(int) {}
  __device__ __forceinline__ void elem(int nt, f32x16& acc, int r) {
    acc[r] = fmaxf(acc[r], 0.f);
    if (row && (r & 3) == 3) store_quad(row, nt, r >> 2, hi, valid && (own < 0 || (nt >> 1) == own), acc);
  }
  I2SDF_APPLY_FROM_ELEM
};
// plain store                                          (feature tiles)
struct StoreEpi {
  float* row; int hi; bool valid;
  __device__ __forceinline__ void prefetch(int) {}
  __device__ __forceinline__ void elem(int nt, f32x16& acc, int r) {
    if ((r & 3) == 3) store_quad(row, nt, r >> 2, hi, valid, acc);
  }
  I2SDF_APPLY_FROM_ELEM
};

// d sdf/dx chain: abar_{l-1} = (W_l^T abar_l) * scale * sigma(h_l) [store]; tiles >= NT (skip layer) add into pbar
template <int NT, int PT, bool PRE>
struct RevEpi {
  const float* hrow; float* abrow; int hi; bool valid; float scale;
  f32x16 (&pt)[PT];
  float hb[2][16];
  __device__ __forceinline__ void prefetch(int nt) {
    if (PRE && nt < NT) load_tile16(hrow, nt, hi, hb[nt & 1]);
  }
  __device__ __forceinline__ void elem(int nt, f32x16& acc, int r) {
    if (nt < NT) {
      if (!PRE && r == 0) load_tile16(hrow, nt, hi, hb[nt & 1]);
      acc[r] = acc[r] * scale * sp_sigma_from_h(hb[nt & 1][r]);
      if (abrow && (r & 3) == 3) store_quad(abrow, nt, r >> 2, hi, valid, acc);
    } else {
      pt[nt - NT < 0 ? 0 : nt - NT][r] += acc[r] * scale;
    }
  }
  I2SDF_APPLY_FROM_ELEM
};

// backward sweep 1: G(hbar_{l+1}) = G(abar_l) * sigma_l [store, stays in acc]; G2(a_l) = G(abar_l) * abar_l * 100(1-sigma_l) [store]
template <bool PRE>
struct Sweep1Epi {
  const float* hrow; const float* arow; float* g2row; float* gurow; int hi; bool valid;
  int own = -1;             // >= 0: store only the tiles this wave owns (split-K tail kernels, redundantly computed ops)
  float hb[2][16], ab[2][16];
  float g2[16];
  __device__ __forceinline__ void prefetch(int nt) {
    if (PRE) { load_tile16(hrow, nt, hi, hb[nt & 1]); load_tile16(arow, nt, hi, ab[nt & 1]); }
  }
  __device__ __forceinline__ void elem(int nt, f32x16& acc, int r) {
    if (!PRE && r == 0) { load_tile16(hrow, nt, hi, hb[nt & 1]); load_tile16(arow, nt, hi, ab[nt & 1]); }
    const float ga = acc[r];
    const float sg = sp_sigma_from_h(hb[nt & 1][r]);
    acc[r] = ga * sg;
    g2[r] = ga * ab[nt & 1][r] * (100.f * (1.0f - sg));
    if ((r & 3) == 3) {
      const bool v = valid && (own < 0 || (nt >> 1) == own);
      store_quad(g2row, nt, r >> 2, hi, v, g2); store_quad(gurow, nt, r >> 2, hi, v, acc);
    }
  }
  I2SDF_APPLY_FROM_ELEM
};

// backward sweep 2: G(a_l) = (W_{l+1}^T G(a_{l+1})) * scale * sigma_l + G2(a_l) [store]; tiles >= NT ignored
template <int NT, bool PRE>
struct Sweep2Epi {
  const float* hrow; const float* g2row; float* grow; int hi; bool valid; float scale;
  float hb[2][16], gb[2][16];
  __device__ __forceinline__ void prefetch(int nt) {
    if (PRE && nt < NT) { load_tile16(hrow, nt, hi, hb[nt & 1]); load_tile16(g2row, nt, hi, gb[nt & 1]); }
  }
  __device__ __forceinline__ void elem(int nt, f32x16& acc, int r) {
    if (nt < NT) {
      if (!PRE && r == 0) { load_tile16(hrow, nt, hi, hb[nt & 1]); load_tile16(g2row, nt, hi, gb[nt & 1]); }
      acc[r] = fmaf(acc[r] * scale, sp_sigma_from_h(hb[nt & 1][r]), gb[nt & 1][r]);
      if ((r & 3) == 3) store_quad(grow, nt, r >> 2, hi, valid, acc);
    }
  }
  I2SDF_APPLY_FROM_ELEM
};
// top of sweep 2: G(h_{L-1}) = W_feat^T fbar + sbar * w_sdf, then as above
template <int NT, bool PRE>
struct Sweep2TopEpi {
  const float* hrow; const float* g2row; float* grow; int hi; bool valid; float sb;
  const float (&wv)[NT * 16];
  float hb[2][16], gb[2][16];
  __device__ __forceinline__ void prefetch(int nt) {
    if (PRE) { load_tile16(hrow, nt, hi, hb[nt & 1]); load_tile16(g2row, nt, hi, gb[nt & 1]); }
  }
  __device__ __forceinline__ void elem(int nt, f32x16& acc, int r) {
    if (!PRE && r == 0) { load_tile16(hrow, nt, hi, hb[nt & 1]); load_tile16(g2row, nt, hi, gb[nt & 1]); }
    acc[r] = fmaf(fmaf(sb, wv[nt * 16 + r], acc[r]), sp_sigma_from_h(hb[nt & 1][r]), gb[nt & 1][r]);
    if ((r & 3) == 3) store_quad(grow, nt, r >> 2, hi, valid, acc);
  }
  I2SDF_APPLY_FROM_ELEM
};

// the same for the split-K kernels: the owner wave holds only its quarter of w_sdf (local index (nt&1)*16 + r)
template <bool PRE>
struct Sweep2TopEpiQ {
  const float* hrow; const float* g2row; float* grow; int hi; bool valid; float sb;
  const float (&wq)[32];
  float hb[2][16], gb[2][16];
  __device__ __forceinline__ void prefetch(int nt) {
    if (PRE) { load_tile16(hrow, nt, hi, hb[nt & 1]); load_tile16(g2row, nt, hi, gb[nt & 1]); }
  }
  __device__ __forceinline__ void elem(int nt, f32x16& acc, int r) {
    if (!PRE && r == 0) { load_tile16(hrow, nt, hi, hb[nt & 1]); load_tile16(g2row, nt, hi, gb[nt & 1]); }
    acc[r] = fmaf(fmaf(sb, wq[(nt & 1) * 16 + r], acc[r]), sp_sigma_from_h(hb[nt & 1][r]), gb[nt & 1][r]);
    if ((r & 3) == 3) store_quad(grow, nt, r >> 2, hi, valid, acc);
  }
  I2SDF_APPLY_FROM_ELEM
};

// radiance backward: G(a_l) = G(r_{l+1}) masked by r_{l+1} > 0 [store]
template <bool PRE>
struct MaskEpi {
  const float* rrow; float* grow; int hi; bool valid;
  float rb[2][16];
  __device__ __forceinline__ void prefetch(int nt) { if (PRE) load_tile16(rrow, nt, hi, rb[nt & 1]); }
  __device__ __forceinline__ void elem(int nt, f32x16& acc, int r) {
    if (!PRE && r == 0) load_tile16(rrow, nt, hi, rb[nt & 1]);
    acc[r] = rb[nt & 1][r] > 0.f ? acc[r] : 0.f;
    if ((r & 3) == 3) store_quad(grow, nt, r >> 2, hi, valid, acc);
  }
  I2SDF_APPLY_FROM_ELEM
};

}  // namespace i2sdf
