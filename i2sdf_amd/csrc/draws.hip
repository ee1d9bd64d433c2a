// All random draws of one training forward in ONE launch.
//
// The reference draws them with separate torch calls inside the sampler and the network forward (ray_sampler.py:60-66 stratified
// jitter `torch.rand(z_vals.shape)`, :176-177 inverse-CDF `torch.rand(.., N_samples)`, :223 `torch.randperm(n)[:N_samples_extra]`,
// :234 `torch.randint(n_z, (B,))`; model/network/__init__.py:177,184 `uniform_(-R, R)` eikonal points and `uniform_(-0.005,
// 0.005)` neighbour offsets).  As torch ops that is ~12 launches of 5-40 us on the critical path of a 7 ms step (the k-subset
// alone is a top-k over random keys).  Here: Philox4x32-10, counter = element index, key = (seed, stream id), one thread per
// four outputs; the k-subsets (one per possible sampler iteration, row length n_eval*(it+1)) are drawn by one thread each by
// rejection (k = 32 of n >= 128: exactly uniform, in random order, ~1.1 draws per element).
#include <hip/hip_runtime.h>
#include "../../include/i2sdf.h"

int i2sdf_hip_check(hipError_t e, const char* what);

namespace {

struct U4 { unsigned x, y, z, w; };
constexpr int MAX_EXTRA_ROWS = 16, MAX_EXTRA = 128;      // sampler iterations / extra columns the k-subset thread supports

__device__ __forceinline__ U4 philox4x32_10(U4 c, unsigned k0, unsigned k1) {
#pragma unroll
  for (int r = 0; r < 10; ++r) {
    const unsigned long long p0 = 0xD2511F53ull * c.x, p1 = 0xCD9E8D57ull * c.z;
    c = U4{(unsigned)(p1 >> 32) ^ c.y ^ k0, (unsigned)p1, (unsigned)(p0 >> 32) ^ c.w ^ k1, (unsigned)p0};
    k0 += 0x9E3779B9u; k1 += 0xBB67AE85u;
  }
  return c;
}
__device__ __forceinline__ float u01(unsigned x) { return (float)(x >> 8) * 5.9604644775390625e-8f; }   // [0, 1), 24 bits like torch.rand

struct DrawArgs {
  unsigned seed_lo, seed_hi;
  int64_t n_strat, n_cdf, n_eik, n_nbr, B;      // element counts of the float outputs (n_eik = n_nbr = 3B)
  int32_t n_z, n_eval, n_extra, max_iters;
  float R, nbr_half;
  float *strat_u, *cdf_u, *eik_pts, *nbr_off;
  int32_t *eik_idx, *extra_idx;
};

__device__ __forceinline__ void fill4(float* out, int64_t n, int64_t q, unsigned stream, const DrawArgs& a, float lo, float scale) {
  if (out == nullptr || 4 * q >= n) return;
  const U4 r = philox4x32_10(U4{(unsigned)q, (unsigned)(q >> 32), stream, 0u}, a.seed_lo, a.seed_hi);
  const unsigned v[4] = {r.x, r.y, r.z, r.w};
#pragma unroll
  for (int i = 0; i < 4; ++i)
    if (4 * q + i < n) out[4 * q + i] = fmaf(u01(v[i]), scale, lo);
}

__global__ __launch_bounds__(256) void draws_kernel(DrawArgs a) {
  const int64_t t = (int64_t)blockIdx.x * 256 + threadIdx.x;
  fill4(a.strat_u, a.n_strat, t, 1u, a, 0.f, 1.f);
  fill4(a.cdf_u, a.n_cdf, t, 2u, a, 0.f, 1.f);
  fill4(a.eik_pts, a.n_eik, t, 3u, a, -a.R, 2.f * a.R);
  fill4(a.nbr_off, a.n_nbr, t, 4u, a, -a.nbr_half, 2.f * a.nbr_half);
  if (a.eik_idx != nullptr && 4 * t < a.B) {
    const U4 r = philox4x32_10(U4{(unsigned)t, (unsigned)(t >> 32), 5u, 0u}, a.seed_lo, a.seed_hi);
    const unsigned v[4] = {r.x, r.y, r.z, r.w};
#pragma unroll
    for (int i = 0; i < 4; ++i)
      if (4 * t + i < a.B) a.eik_idx[4 * t + i] = (int32_t)(((unsigned long long)v[i] * (unsigned)a.n_z) >> 32);
  }
  // row r: n_extra distinct columns of [0, n_eval*(r+1)), random order, drawn by rejection.  The chain of a row is sequential, so a row
  // is given to one WAVE of block 0: every lane runs the same Philox sequence, lane j keeps entries j and j+64 of the row in registers,
  // and "is the candidate already in the row" is one ballot instead of a loop over the row (one thread per row with the row in LDS took
  // 60 us -- the longest single item in front of the sampler; the sequence of accepted columns is unchanged).
  if (a.extra_idx != nullptr && blockIdx.x == 0) {
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    for (int row = wv; row < a.max_iters; row += 4) {
      const unsigned n = (unsigned)a.n_eval * (unsigned)(row + 1);
      unsigned ctr = 0;
      U4 r{};
      int have = 0;
      int32_t mine0 = -1, mine1 = -1;
      for (int k = 0; k < a.n_extra; ++k) {
        for (;;) {
          if (have == 0) { r = philox4x32_10(U4{ctr++, (unsigned)row, 6u, 0u}, a.seed_lo, a.seed_hi); have = 4; }
          const unsigned x = have == 4 ? r.x : have == 3 ? r.y : have == 2 ? r.z : r.w;
          --have;
          const int32_t c = (int32_t)(((unsigned long long)x * n) >> 32);
          if (__ballot(mine0 == c || mine1 == c) == 0ull) {
            if ((k & 63) == lane) { if (k < 64) mine0 = c; else mine1 = c; }
            break;
          }
        }
      }
      if (lane < a.n_extra) a.extra_idx[row * a.n_extra + lane] = mine0;
      if (lane + 64 < a.n_extra) a.extra_idx[row * a.n_extra + lane + 64] = mine1;
    }
  }
}

}  // namespace

extern "C" int i2sdf_training_draws(uint64_t seed, int64_t B, int32_t n_eval, int32_t n_samples, int32_t n_extra, int32_t max_iters,
                                    int32_t n_z, float eik_radius, float nbr_half_width, float* strat_u, float* cdf_u,
                                    int32_t* extra_idx, int32_t* eik_idx, float* eik_pts, float* nbr_off, void* stream) {
  if (B < 0 || n_eval < 0 || n_samples < 0 || n_extra < 0 || max_iters < 0 || n_z < 0) return I2SDF_EINVAL;
  if (extra_idx != nullptr && n_extra > 0 && (n_eval <= 0 || n_extra > n_eval)) return I2SDF_EINVAL;   // k distinct of n_eval*(it+1) >= n_eval
  if (extra_idx != nullptr && n_extra > 0 && (max_iters > MAX_EXTRA_ROWS || n_extra > MAX_EXTRA)) return I2SDF_EINVAL;
  if (eik_idx != nullptr && n_z <= 0) return I2SDF_EINVAL;
  DrawArgs a{};
  a.seed_lo = (unsigned)seed; a.seed_hi = (unsigned)(seed >> 32);
  a.B = B; a.n_strat = B * n_eval; a.n_cdf = B * n_samples; a.n_eik = 3 * B; a.n_nbr = 3 * B;
  a.n_z = n_z; a.n_eval = n_eval; a.n_extra = n_extra; a.max_iters = (extra_idx != nullptr && n_extra > 0) ? max_iters : 0;
  a.R = eik_radius; a.nbr_half = nbr_half_width;
  a.strat_u = strat_u; a.cdf_u = cdf_u; a.eik_pts = eik_pts; a.nbr_off = nbr_off; a.eik_idx = eik_idx;
  a.extra_idx = a.max_iters > 0 ? extra_idx : nullptr;
  int64_t quads = 0;
  auto need = [&](const void* p, int64_t n) { if (p != nullptr && (n + 3) / 4 > quads) quads = (n + 3) / 4; };
  need(strat_u, a.n_strat); need(cdf_u, a.n_cdf); need(eik_pts, a.n_eik); need(nbr_off, a.n_nbr); need(eik_idx, B);
  if (a.max_iters > quads) quads = a.max_iters;
  if (quads == 0) return I2SDF_OK;
  draws_kernel<<<(unsigned)((quads + 255) / 256), 256, 0, (hipStream_t)stream>>>(a);
  return i2sdf_hip_check(hipGetLastError(), "draws_kernel");
}
