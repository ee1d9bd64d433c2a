// Weight-norm reparametrisation + stream packing.
// Replaces the `W = g * v / ||v||_row` that the reference evaluates inside every `lin(x)` call
// (model/network/mlp.py:71-72 -> torch.nn.utils.weight_norm, dim=0) by one pass per optimizer step, and writes W
// in MFMA consumption order (see common.h).  HBM-bound: ~3.2 MB in, ~2 x 3.4 MB out.
#include "plan.h"

using namespace i2sdf;

int i2sdf_hip_check(hipError_t e, const char* what);

namespace {

struct RowTab {
  int32_t n;
  int32_t rows[3 * I2SDF_MAX_LAYERS], cols[3 * I2SDF_MAX_LAYERS], scale_off[3 * I2SDF_MAX_LAYERS];
  int64_t off_v[3 * I2SDF_MAX_LAYERS], off_g[3 * I2SDF_MAX_LAYERS];
};

// one wave per weight row: scale[row] = g[row] / ||v[row,:]||
__global__ __launch_bounds__(256) void rowscale_kernel(RowTab tab, const float* __restrict__ params, float* __restrict__ scale,
                                                        int n_rows) {
  const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
  const int lane = threadIdx.x & 63;
  if (row >= n_rows) return;
  int e = 0;
  while (e + 1 < tab.n && row >= tab.scale_off[e + 1]) ++e;
  const int r = row - tab.scale_off[e];
  const int cols = tab.cols[e];
  const float* v = params + tab.off_v[e] + (int64_t)r * cols;
  float s = 0.f;
  for (int c = lane; c < cols; c += 64) { float x = v[c]; s = fmaf(x, x, s); }
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) s += __shfl_xor(s, o);
  if (lane == 0) scale[row] = params[tab.off_g[e] + r] / sqrtf(s);
}

__device__ __forceinline__ int map_col(const ColMap& cm, int kp) {
  if (kp < cm.split) return kp < cm.valid0 ? cm.base0 + kp : -1;
  const int j = kp - cm.split;
  return j < cm.valid1 ? cm.base1 + j : -1;
}

// one thread per (chunk, lane): writes 4 floats
__global__ __launch_bounds__(256) void pack_kernel(const Seg* __restrict__ segs, int n_segs, const float* __restrict__ params,
                                                    const float* __restrict__ scale, float* __restrict__ chunks,
                                                    int64_t total_chunks) {
  const int64_t gid = (int64_t)blockIdx.x * 256 + threadIdx.x;
  const int64_t chunk = gid >> 6;
  const int lane = (int)(gid & 63);
  if (chunk >= total_chunks) return;
  int lo = 0, hi_ = n_segs - 1;
  while (lo < hi_) {
    const int mid = (lo + hi_ + 1) >> 1;
    if (segs[mid].chunk0 <= chunk) lo = mid; else hi_ = mid - 1;
  }
  const Seg s = segs[lo];
  const int c = (int)(chunk - s.chunk0);
  const int i32 = lane & 31, hi = lane >> 5;
  f32x4 out = {0.f, 0.f, 0.f, 0.f};
  if (c < s.used) {
    if (s.type == SEG_BIAS) {
      const int nt = c / 4, q = c % 4;
#pragma unroll
      for (int t = 0; t < 4; ++t) {
        const int r = 32 * nt + 8 * q + 4 * hi + t;
        out[t] = (r < s.nrows) ? params[s.off_bias + s.row_off + r] : 0.f;
      }
    } else if (s.type == SEG_WFWD) {
      const int nt = c / s.KC, kc = c % s.KC;
      const int r = 32 * nt + i32;
      if (r < s.nrows) {
        const int row = s.row_off + r;
        const float sc = scale[s.scale_off + row] * s.mult;
        const float* v = params + s.off_v + (int64_t)row * s.cols;
#pragma unroll
        for (int t = 0; t < 4; ++t) {
          const int col = map_col(s.cm, 8 * kc + 4 * hi + t);
          out[t] = col >= 0 ? v[col] * sc : 0.f;
        }
      }
    } else if (s.type == SEG_WFWD3) {
      // chunk c -> pair w = c/2 (e = c&1), k-chunk kc = w / (3G), tile pair g, split plane sp (x3.h)
      const int G = s.NT / 2, w = c >> 1, e = c & 1;
      const int kc = w / (3 * G), g = (w / 3) % G, sp = w % 3;
      const int r = 32 * (2 * g + e) + i32;
      unsigned q[4] = {0u, 0u, 0u, 0u};
      if (r < s.nrows) {
        const int row = s.row_off + r;
        const float sc = scale[s.scale_off + row] * s.mult;
        const float* v = params + s.off_v + (int64_t)row * s.cols;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
          float x[2];
#pragma unroll
          for (int h2 = 0; h2 < 2; ++h2) {
            const int j = 2 * i + h2;
            const int col = map_col(s.cm, 16 * kc + (j & 3) + 8 * (j >> 2) + 4 * hi);
            x[h2] = col >= 0 ? v[col] * sc : 0.f;
          }
          unsigned p0, p1, p2;
          split3_pair(x[0], x[1], p0, p1, p2);
          q[i] = sp == 0 ? p0 : (sp == 1 ? p1 : p2);
        }
      }
      out = __builtin_bit_cast(f32x4, u32x4{q[0], q[1], q[2], q[3]});
    } else if (s.type == SEG_WBWD3) {
      // transposed: A-operand row = input column 32*kt + i32 of the layer, reduction index = output row (x3.h k-order)
      const int G = s.NT / 2, w = c >> 1, e = c & 1;
      const int kc = w / (3 * G), g = (w / 3) % G, sp = w % 3;
      const int col = map_col(s.cm, 32 * (2 * g + e) + i32);
      unsigned q[4] = {0u, 0u, 0u, 0u};
      if (col >= 0) {
#pragma unroll
        for (int i = 0; i < 4; ++i) {
          float x[2];
#pragma unroll
          for (int h2 = 0; h2 < 2; ++h2) {
            const int j = 2 * i + h2;
            const int r = 16 * kc + (j & 3) + 8 * (j >> 2) + 4 * hi;
            x[h2] = 0.f;
            if (r < s.nrows) {
              const int row = s.row_off + r;
              x[h2] = params[s.off_v + (int64_t)row * s.cols + col] * scale[s.scale_off + row] * s.mult;
            }
          }
          unsigned p0, p1, p2;
          split3_pair(x[0], x[1], p0, p1, p2);
          q[i] = sp == 0 ? p0 : (sp == 1 ? p1 : p2);
        }
      }
      out = __builtin_bit_cast(f32x4, u32x4{q[0], q[1], q[2], q[3]});
    } else if (s.type == SEG_BIAS_H) {
      // 16-point-wave family (x3h.h): lane = i16 + 16*kg; D layout of tile nt: register t <-> row 16*nt + 4*kg + t
      const int kg = lane >> 4;
#pragma unroll
      for (int t = 0; t < 4; ++t) {
        const int r = 16 * c + 4 * kg + t;
        out[t] = (r < s.nrows) ? params[s.off_bias + s.row_off + r] : 0.f;
      }
    } else if (s.type == SEG_WFWD3H || s.type == SEG_WBWD3H || s.type == SEG_WFWD2H) {
      // chunk c -> pair w = c/2 (e = c&1), k-chunk kc = w / (PL G), tile pair g, split plane sp < PL (3; SEG_WFWD2H: 2); A row = 16*(2g+e) + i16,
      // element j <-> reduction index 32*kc + 16*(j>>2) + 4*kg + (j&3)
      const int G = s.NT / 2, w = c >> 1, e = c & 1, PL = s.type == SEG_WFWD2H ? 2 : 3;
      const int kc = w / (PL * G), g = (w / PL) % G, sp = w % PL;
      const int i16 = lane & 15, kg = lane >> 4;
      const int arow = 16 * (2 * g + e) + i16;
      const bool fwd = s.type != SEG_WBWD3H;
      const int acol = fwd ? -1 : map_col(s.cm, arow);            // transposed: the A row is an input column of the layer
      unsigned q[4] = {0u, 0u, 0u, 0u};
      if (fwd ? (arow < s.nrows) : (acol >= 0)) {
#pragma unroll
        for (int i = 0; i < 4; ++i) {
          float x[2];
#pragma unroll
          for (int h2 = 0; h2 < 2; ++h2) {
            const int j = 2 * i + h2;
            const int k = 32 * kc + 16 * (j >> 2) + 4 * kg + (j & 3);
            x[h2] = 0.f;
            if (fwd) {
              const int row = s.row_off + arow, col = map_col(s.cm, k);
              if (col >= 0) x[h2] = params[s.off_v + (int64_t)row * s.cols + col] * (scale[s.scale_off + row] * s.mult);
            } else if (k < s.nrows) {
              const int row = s.row_off + k;
              x[h2] = params[s.off_v + (int64_t)row * s.cols + acol] * scale[s.scale_off + row] * s.mult;
            }
          }
          unsigned p0, p1, p2;
          split3_pair(x[0], x[1], p0, p1, p2);
          q[i] = sp == 0 ? p0 : (sp == 1 ? p1 : p2);
        }
      }
      out = __builtin_bit_cast(f32x4, u32x4{q[0], q[1], q[2], q[3]});
    } else if (s.type == SEG_ROWVEC_H) {
      const int rr = c / s.KC, nt = c % s.KC, kg = lane >> 4;
      const int row = s.row_off + rr;
      const float sc = scale[s.scale_off + row] * s.mult;
#pragma unroll
      for (int t = 0; t < 4; ++t) {
        const int col = map_col(s.cm, 16 * nt + 4 * kg + t);
        out[t] = col >= 0 ? params[s.off_v + (int64_t)row * s.cols + col] * sc : 0.f;
      }
    } else if (s.type == SEG_WBWD) {
      const int kt = c / s.KC, nc = c % s.KC;
      const int col = map_col(s.cm, 32 * kt + i32);
      if (col >= 0) {
#pragma unroll
        for (int t = 0; t < 4; ++t) {
          const int r = 8 * nc + 4 * hi + t;
          if (r < s.nrows) {
            const int row = s.row_off + r;
            out[t] = params[s.off_v + (int64_t)row * s.cols + col] * scale[s.scale_off + row] * s.mult;
          }
        }
      }
    } else if (s.type == SEG_ROWVEC) {
      const int rr = c / s.KC, kc = c % s.KC;
      const int row = s.row_off + rr;
      const float sc = scale[s.scale_off + row] * s.mult;
#pragma unroll
      for (int t = 0; t < 4; ++t) {
        const int col = map_col(s.cm, 8 * kc + 4 * hi + t);
        out[t] = col >= 0 ? params[s.off_v + (int64_t)row * s.cols + col] * sc : 0.f;
      }
    } else if (s.type == SEG_SCALAR) {
#pragma unroll
      for (int t = 0; t < 4; ++t) out[t] = (t < s.nrows) ? params[s.off_bias + s.row_off + t] : 0.f;
    }
  }
  reinterpret_cast<f32x4*>(chunks)[chunk * 64 + lane] = out;
}

void fill_tab(RowTab& tab, const NetPlan& np) {
  for (int l = 0; l < np.d.n_lin; ++l) {
    const int e = tab.n++;
    tab.rows[e] = np.d.out_dim[l]; tab.cols[e] = np.d.in_dim[l]; tab.scale_off[e] = np.scale_off[l];
    tab.off_v[e] = np.d.off_v[l]; tab.off_g[e] = np.d.off_g[l];
  }
}

}  // namespace

extern "C" int i2sdf_pack_weights(const i2sdf_plan* p, const float* params, float* packed, void* stream) {
  if (!p || !params || !packed) return I2SDF_EINVAL;
  if (!p->d_segs) return I2SDF_EHIP;
  hipStream_t st = (hipStream_t)stream;
  RowTab tab{};
  fill_tab(tab, p->sdf); fill_tab(tab, p->rgb);
  if (p->light.d.n_lin) fill_tab(tab, p->light);
  rowscale_kernel<<<cdiv(p->n_scale, 4), 256, 0, st>>>(tab, params, packed, p->n_scale);
  const int64_t threads = p->total_chunks * 64;
  pack_kernel<<<(unsigned)((threads + 255) / 256), 256, 0, st>>>(p->d_segs, p->n_segs, params, packed, packed + p->scale_floats,
                                                                 p->total_chunks);
  return i2sdf_hip_check(hipGetLastError(), "pack_weights launch");
}
