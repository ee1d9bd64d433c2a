// Split-K execution of a dense op for the LAST, partially filled round of workgroups.
//
// 800 workgroups of 128 points on 256 CUs are 3.125 rounds: the fourth round keeps 32 CUs busy and 224 idle for a
// whole round.  Here the 4 waves (= the 4 SIMDs of a CU) share ONE 32-point tile instead of owning one each: wave w
// reduces over a quarter of the layer's input features (chunks kc in [8w, 8w+8) of every LDS stage) for ALL output tiles,
// the four partial tiles are summed through LDS by the tile's owner wave (tile nt -> wave nt/2), which also runs the
// epilogue and keeps the result as its quarter of the next layer's input.  A 128-point workgroup becomes four 32-point
// workgroups that each finish in ~1/3 of the time, so the last round shrinks from 1 to ~0.35 rounds.
// Needs the op's tiles to be aligned with the LDS stages (reduction length = 32 chunks); other ops are computed
// redundantly by every wave (see the kernels).
#pragma once
#include "epi.h"

namespace i2sdf {

constexpr int KS_TILE_FLOATS = 16 * 64;                         // one D-layout tile of one wave
constexpr int KS_X_FLOATS = 2 * 4 * KS_TILE_FLOATS;             // exchange area: 2 buffers x 4 waves
constexpr int KS_G_FLOATS = 8 * KS_TILE_FLOATS;                 // all-gather area: one full 8-tile vector
constexpr int KS_S_FLOATS = 1024;                               // small scratch (row-vector sums)
constexpr int KS_LDS_BYTES = LDS_BYTES + (KS_X_FLOATS + KS_S_FLOATS + KS_G_FLOATS) * 4;

// NT output tiles (tiles >= 8 belong to wave 0 and are returned through `extra`), reduction over exactly SC chunks.
//   NB = NT*4 (bias stage first, MODE 0 adds it, MODE 1 ignores it) or 0
template <int NT, int NB, int MODE, class Epi>
__device__ __forceinline__ void dense_op_ksplit(WStream& ws, const float (&inq)[32], float (&outq)[32], f32x16* extra, Epi& epi,
                                                float* xlds, int tid) {
  static_assert(NB == 0 || NB % SC == 0, "bias chunks must fill whole stages");
  constexpr int NBS = NB / SC;
  const int lane = tid & 63, w = tid >> 6;
  float bq[32];
#pragma unroll
  for (int i = 0; i < 32; ++i) bq[i] = 0.f;
#pragma unroll
  for (int s = 0; s < NBS; ++s) {
    const f32x4* cur = reinterpret_cast<const f32x4*>(ws.advance(tid)) + lane;
#pragma unroll
    for (int j = 0; j < SC; ++j) {
      const int c = s * SC + j, nt = c / 4, q = c % 4;
      if (MODE == 0 && nt < 8 && (nt >> 1) == w) {
        const f32x4 b = cur[j * 64];
        bq[(nt & 1) * 16 + 4 * q] = b.x; bq[(nt & 1) * 16 + 4 * q + 1] = b.y; bq[(nt & 1) * 16 + 4 * q + 2] = b.z; bq[(nt & 1) * 16 + 4 * q + 3] = b.w;
      }
    }
  }
#pragma unroll
  for (int nt = 0; nt < NT; ++nt) {
    const f32x4* cur = reinterpret_cast<const f32x4*>(ws.advance(tid)) + lane + (8 * w) * 64;
    const bool owner = (nt < 8) ? ((nt >> 1) == w) : (w == 0);
    if (owner) epi.prefetch(nt);
    f32x16 part;
#pragma unroll
    for (int r = 0; r < 16; ++r) part[r] = 0.f;
    f32x4 ab[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) ab[i] = cur[i * 64];
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      const f32x4 a = ab[i % 4];
      if (i + 4 < 8) ab[i % 4] = cur[(i + 4) * 64];
      part = mfma(a.x, inq[i * 4 + 0], part);
      part = mfma(a.y, inq[i * 4 + 1], part);
      part = mfma(a.z, inq[i * 4 + 2], part);
      part = mfma(a.w, inq[i * 4 + 3], part);
    }
    float* xb = xlds + ((nt & 1) * 4) * KS_TILE_FLOATS;
    {
      f32x4* dst = reinterpret_cast<f32x4*>(xb + w * KS_TILE_FLOATS) + lane;
#pragma unroll
      for (int q = 0; q < 4; ++q) dst[q * 64] = f32x4{part[4 * q], part[4 * q + 1], part[4 * q + 2], part[4 * q + 3]};
    }
    __syncthreads();
    if (owner) {
      f32x16 tot;
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        f32x4 s4 = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int v = 0; v < 4; ++v) s4 += (reinterpret_cast<const f32x4*>(xb + v * KS_TILE_FLOATS) + lane)[q * 64];
        tot[4 * q] = s4.x; tot[4 * q + 1] = s4.y; tot[4 * q + 2] = s4.z; tot[4 * q + 3] = s4.w;
      }
      if (MODE == 0 && nt < 8) {
#pragma unroll
        for (int r = 0; r < 16; ++r) tot[r] += bq[(nt & 1) * 16 + r];
      }
      if (nt < 8) {
        epi.apply(nt, tot);
#pragma unroll
        for (int r = 0; r < 16; ++r) outq[(nt & 1) * 16 + r] = tot[r];
      } else {
        epi.apply(nt, tot);                       // tiles beyond the 8 hidden ones (PE part of the skip layer): wave 0, handled by the epilogue
        if (extra != nullptr) {
#pragma unroll
          for (int r = 0; r < 16; ++r) extra[nt - 8][r] = tot[r];
        }
      }
    }
  }
}

// this wave's quarter (tiles 2w, 2w+1) of a full 8-tile register vector that every wave holds
__device__ __forceinline__ void take_quarter(const float (&full)[128], float (&q)[32], int w) {
#pragma unroll
  for (int i = 0; i < 32; ++i) {
    const float a = (w & 1) ? full[32 + i] : full[i];
    const float b = (w & 1) ? full[96 + i] : full[64 + i];
    q[i] = (w & 2) ? b : a;
  }
}

// every wave contributes its quarter (tiles 2w, 2w+1), every wave receives the full 8-tile vector
__device__ __forceinline__ void gather_full(const float (&q)[32], float (&full)[128], float* glds, int tid) {
  const int lane = tid & 63, w = tid >> 6;
#pragma unroll
  for (int t = 0; t < 2; ++t) {
    f32x4* dst = reinterpret_cast<f32x4*>(glds + (2 * w + t) * KS_TILE_FLOATS) + lane;
#pragma unroll
    for (int qq = 0; qq < 4; ++qq) dst[qq * 64] = f32x4{q[t * 16 + 4 * qq], q[t * 16 + 4 * qq + 1], q[t * 16 + 4 * qq + 2], q[t * 16 + 4 * qq + 3]};
  }
  __syncthreads();
#pragma unroll
  for (int nt = 0; nt < 8; ++nt) {
    const f32x4* src = reinterpret_cast<const f32x4*>(glds + nt * KS_TILE_FLOATS) + lane;
#pragma unroll
    for (int qq = 0; qq < 4; ++qq) {
      const f32x4 v = src[qq * 64];
      full[nt * 16 + 4 * qq] = v.x; full[nt * 16 + 4 * qq + 1] = v.y; full[nt * 16 + 4 * qq + 2] = v.z; full[nt * 16 + 4 * qq + 3] = v.w;
    }
  }
  __syncthreads();
}

// this wave's quarter of a row vector stored as a rowvec op: regs[i*4+t] = w[8*(8w+i) + 4hi + t]; also the scalar chunk
__device__ __forceinline__ void rowvec_load_quarter(WStream& ws, float (&wq)[32], f32x4& scalars, int tid) {
  constexpr int KC = 32, TOT = rowvec_chunks(KC, 1), NS = TOT / SC;
  const int lane = tid & 63, w = tid >> 6;
#pragma unroll
  for (int s = 0; s < NS; ++s) {
    const f32x4* cur = reinterpret_cast<const f32x4*>(ws.advance(tid)) + lane;
    if (s == 0) {
#pragma unroll
      for (int i = 0; i < 8; ++i) {
        const f32x4 v = cur[(8 * w + i) * 64];
        wq[i * 4] = v.x; wq[i * 4 + 1] = v.y; wq[i * 4 + 2] = v.z; wq[i * 4 + 3] = v.w;
      }
    } else if (s == 1) {
      scalars = cur[0];
    }
  }
}

// store / load this wave's quarter of a point-major [m][256] row (columns 64w .. 64w+63)
__device__ __forceinline__ void store_quarter(float* __restrict__ row, int w, int hi, bool valid, const float (&q)[32]) {
  if (!valid) return;
#pragma unroll
  for (int i = 0; i < 8; ++i) *reinterpret_cast<f32x4*>(row + 64 * w + 8 * i + 4 * hi) = f32x4{q[4 * i], q[4 * i + 1], q[4 * i + 2], q[4 * i + 3]};
}
__device__ __forceinline__ void load_quarter(const float* __restrict__ row, int w, int hi, float (&q)[32]) {
#pragma unroll
  for (int i = 0; i < 8; ++i) {
    const f32x4 v = *reinterpret_cast<const f32x4*>(row + 64 * w + 8 * i + 4 * hi);
    q[4 * i] = v.x; q[4 * i + 1] = v.y; q[4 * i + 2] = v.z; q[4 * i + 3] = v.w;
  }
}

// row-vector op, split-K: NROWS dot products over the wave's quarter, summed across the 4 waves through LDS
//   stream layout as rowvec_op: [NROWS*32 weight chunks][1 scalar chunk], padded
template <int NROWS>
__device__ __forceinline__ void rowvec_ksplit(WStream& ws, const float (&inq)[32], float (&out)[NROWS], float* slds, int tid) {
  constexpr int KC = 32, TOT = rowvec_chunks(KC, NROWS), NS = TOT / SC, NW = NROWS * KC;
  const int lane = tid & 63, w = tid >> 6;
  float part[NROWS];
#pragma unroll
  for (int r = 0; r < NROWS; ++r) part[r] = 0.f;
  f32x4 sc = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
  for (int s = 0; s < NS; ++s) {
    const f32x4* cur = reinterpret_cast<const f32x4*>(ws.advance(tid)) + lane;
    // row `s` occupies the whole stage (KC == SC); this wave's chunks are [8w, 8w+8)
    if (s * SC < NW) {
#pragma unroll
      for (int i = 0; i < 8; ++i) {
        const f32x4 wv = cur[(8 * w + i) * 64];
        part[s < NROWS ? s : 0] = fmaf(wv.x, inq[i * 4 + 0], part[s < NROWS ? s : 0]);
        part[s < NROWS ? s : 0] = fmaf(wv.y, inq[i * 4 + 1], part[s < NROWS ? s : 0]);
        part[s < NROWS ? s : 0] = fmaf(wv.z, inq[i * 4 + 2], part[s < NROWS ? s : 0]);
        part[s < NROWS ? s : 0] = fmaf(wv.w, inq[i * 4 + 3], part[s < NROWS ? s : 0]);
      }
    } else if (s * SC == NW) {
      sc = cur[0];
    }
  }
#pragma unroll
  for (int r = 0; r < NROWS; ++r) {
    const float v = part[r] + __shfl_xor(part[r], 32);
    if (lane < 32) slds[(w * NROWS + r) * 32 + lane] = v;
  }
  __syncthreads();
#pragma unroll
  for (int r = 0; r < NROWS; ++r) {
    float v = 0.f;
#pragma unroll
    for (int u = 0; u < 4; ++u) v += slds[(u * NROWS + r) * 32 + (lane & 31)];
    out[r] = v + (r == 0 ? sc.x : r == 1 ? sc.y : r == 2 ? sc.z : sc.w);
  }
  __syncthreads();
}

}  // namespace i2sdf
