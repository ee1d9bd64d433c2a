// Error-bounded ray sampler: ErrorBoundSampler.get_z_vals / get_error_bound and UniformSampler.get_z_vals
// (model/network/ray_sampler.py:22-43,67-251; VolSDF Algorithm 1), without the SDF evaluations (those run in the
// MFMA kernel of mlp_fwd.hip between the steps below).
//
// One wave per ray; a row of n <= 640 samples lives in registers in a blocked layout (lane owns E consecutive
// samples), cumulative sums are a local serial scan + one cross-lane scan, searches/merges go through a few KB of LDS.
// The reference's `while not_converge` (a device->host sync per iteration, ray_sampler.py:83,151) becomes a flag in
// device memory: every step kernel of every possible iteration is enqueued up front and exits at once when the
// batch has converged, so the host never waits on the device.
//
//   per iteration it (row length n = N_eval * (it + 1)):
//     [sdf_forward on the newest samples]                                   (mlp_fwd.hip, ray mode)
//     sampler_beta_kernel     : merge sdf, d* bound, error bound at beta0, bisection -> beta; atomic OR of "beta > beta0"
//     sampler_resample_kernel : batch flag -> mode; density, transmittance, pdf/cdf, inverse-CDF samples, sorted merge; the last
//                               workgroup to finish raises the "done" flag (no separate launch)
//     (a caller-fixed iteration count needs no batch flag: sampler_iter_fixed_kernel does both steps in one launch)
//   sampler_final_kernel      : near/far/extra samples, sort, eikonal sample
#include "plan.h"

using namespace i2sdf;

int i2sdf_hip_check(hipError_t e, const char* what);

namespace {

constexpr int NMAX = 640;        // longest row (N_eval * max_total_iters)
constexpr int NNEW = 128;        // most new samples per iteration
constexpr int EMAX = NMAX / 64;

// state words
enum { ST_DONE = 0, ST_ITERS = 1, ST_FLAG0 = 2 /* +it */, ST_CNT0 = 16 /* +it: workgroups that finished step B of iteration it */, ST_WORDS = 32 };

// Wave-wide sum / max / inclusive scan by DPP (round 6).  The __shfl_* forms compile to ds_bpermute_b32 -- a trip through the LDS crossbar,
// ~100 cycles of latency each -- and the bisection of beta_step runs eighteen of them, strictly dependent, in every one of its eleven error-bound
// evaluations: at 1024 rays (one workgroup per CU) that latency chain WAS the kernel.  DPP moves data inside the VALU: row_shr:1/2/4/8 scan the
// four 16-lane rows, row_bcast:15 / :31 carry the row totals on (gfx9 has both), lane 63 then holds the total.  SAMPLER_DPP=0: the shuffle forms.
#ifndef SAMPLER_DPP
#define SAMPLER_DPP 1
#endif
template <int CTRL, int ROW_MASK>
__device__ __forceinline__ float dpp_mov(float old, float v) {      // lanes without a source lane (or outside ROW_MASK) get `old`
  return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(__builtin_bit_cast(int, old), __builtin_bit_cast(int, v), CTRL, ROW_MASK, 0xf, false));
}
constexpr int DPP_ROW_SHR1 = 0x111, DPP_ROW_SHR2 = 0x112, DPP_ROW_SHR4 = 0x114, DPP_ROW_SHR8 = 0x118, DPP_ROW_BCAST15 = 0x142, DPP_ROW_BCAST31 = 0x143;
__device__ __forceinline__ float wscan_incl(float v, int lane) {
#if SAMPLER_DPP
  (void)lane;
  v += dpp_mov<DPP_ROW_SHR1, 0xf>(0.f, v);
  v += dpp_mov<DPP_ROW_SHR2, 0xf>(0.f, v);
  v += dpp_mov<DPP_ROW_SHR4, 0xf>(0.f, v);
  v += dpp_mov<DPP_ROW_SHR8, 0xf>(0.f, v);
  v += dpp_mov<DPP_ROW_BCAST15, 0xa>(0.f, v);     // rows 1 and 3 += the total of the row below
  v += dpp_mov<DPP_ROW_BCAST31, 0xc>(0.f, v);     // rows 2 and 3 += the total of lanes 0..31
  return v;
#else
#pragma unroll
  for (int o = 1; o < 64; o <<= 1) {
    const float t = __shfl_up(v, o);
    if (lane >= o) v += t;
  }
  return v;
#endif
}
__device__ __forceinline__ float wsum(float v) {
#if SAMPLER_DPP
  return __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, wscan_incl(v, 0)), 63));
#else
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
  return v;
#endif
}
__device__ __forceinline__ float wmax(float v) {
#if SAMPLER_DPP
  const float lo = -3.4028234663852886e38f;
  v = fmaxf(v, dpp_mov<DPP_ROW_SHR1, 0xf>(lo, v));
  v = fmaxf(v, dpp_mov<DPP_ROW_SHR2, 0xf>(lo, v));
  v = fmaxf(v, dpp_mov<DPP_ROW_SHR4, 0xf>(lo, v));
  v = fmaxf(v, dpp_mov<DPP_ROW_SHR8, 0xf>(lo, v));
  v = fmaxf(v, dpp_mov<DPP_ROW_BCAST15, 0xa>(lo, v));
  v = fmaxf(v, dpp_mov<DPP_ROW_BCAST31, 0xc>(lo, v));
  return __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, v), 63));
#else
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o));
  return v;
#endif
}
// inclusive scan of a blocked row (lane owns E consecutive entries), in place
template <int E>
__device__ __forceinline__ void row_scan_incl(float (&v)[E], int lane) {
#pragma unroll
  for (int e = 1; e < E; ++e) v[e] += v[e - 1];
  const float tot = v[E - 1];
  const float pre = wscan_incl(tot, lane) - tot;
#pragma unroll
  for (int e = 0; e < E; ++e) v[e] += pre;
}

__device__ __forceinline__ float laplace_density(float s, float inv_beta) {
  const float e = expm1f(-fabsf(s) * inv_beta);
  const float sg = (s > 0.f) ? 1.f : ((s < 0.f) ? -1.f : 0.f);
  return inv_beta * (0.5f + 0.5f * sg * e);
}

// A ray's row: z, sdf and the interval quantities, blocked over the wave.
template <int E>
struct Row {
  float z[E], s[E];          // samples j = lane*E + e  (valid j < n)
  float a[E], ds[E];         // interval i = j: a = z[i+1]-z[i], d* bound; valid i < n-1
  int n;
};

// Theorem-1 bound per interval (ray_sampler.py:99-114)
template <int E>
__device__ __forceinline__ void row_intervals(Row<E>& r, int lane) {
  const float zn_next = __shfl_down(r.z[0], 1), sn_next = __shfl_down(r.s[0], 1);
#pragma unroll
  for (int e = 0; e < E; ++e) {
    const int j = lane * E + e;
    const float z1 = (e + 1 < E) ? r.z[(e + 1 < E) ? e + 1 : e] : zn_next;
    const float s1 = (e + 1 < E) ? r.s[(e + 1 < E) ? e + 1 : e] : sn_next;
    float a = 0.f, d = 0.f;
    if (j < r.n - 1) {
      a = z1 - r.z[e];
      const float b = fabsf(r.s[e]), c = fabsf(s1);
      const bool first = a * a + b * b <= c * c;
      const bool second = a * a + c * c <= b * b;
      const float sp = (a + b + c) / 2.0f;
      const float area = sp * (sp - a) * (sp - b) * (sp - c);
      const bool mask = !first && !second && (b + c - a > 0.f);
      float h = (2.0f * sqrtf(area)) / a;
      if (!(h == h)) h = 0.f;                                   // nan_to_num
      else if (h > 3.4028234663852886e38f) h = 3.4028234663852886e38f;
      else if (h < -3.4028234663852886e38f) h = -3.4028234663852886e38f;
      d = ((first && !second) ? b : 0.f) + (second ? c : 0.f) + (mask ? h : 0.f);
      const float sg0 = (r.s[e] > 0.f) ? 1.f : ((r.s[e] < 0.f) ? -1.f : 0.f);
      const float sg1 = (s1 > 0.f) ? 1.f : ((s1 < 0.f) ? -1.f : 0.f);
      if (!(sg0 * sg1 == 1.f)) d = 0.f;
    }
    r.a[e] = a; r.ds[e] = d;
  }
}

// get_error_bound (ray_sampler.py:243-251): max over intervals of (min(exp(cumsum err),1e6)-1) * exp(-integral)
template <int E>
__device__ __forceinline__ float row_error_bound(const Row<E>& r, float beta, int lane) {
  const float ib = 1.0f / beta;
  float fe[E], er[E];
#pragma unroll
  for (int e = 0; e < E; ++e) {
    const int j = lane * E + e;
    const bool ok = j < r.n - 1;
    fe[e] = ok ? r.a[e] * laplace_density(r.s[e], ib) : 0.f;
    er[e] = ok ? expf(-r.ds[e] / beta) * (r.a[e] * r.a[e]) / (4.0f * beta * beta) : 0.f;
  }
  float fi[E], ei[E];
#pragma unroll
  for (int e = 0; e < E; ++e) { fi[e] = fe[e]; ei[e] = er[e]; }
  row_scan_incl<E>(fi, lane);
  row_scan_incl<E>(ei, lane);
  float mx = -3.4e38f;
#pragma unroll
  for (int e = 0; e < E; ++e) {
    const int j = lane * E + e;
    if (j < r.n - 1) {
      const float integ = fi[e] - fe[e];                              // exclusive prefix
      const float bo = (fminf(expf(ei[e]), 1.0e6f) - 1.0f) * expf(-integ);
      mx = fmaxf(mx, bo);
    }
  }
  return wmax(mx);
}

struct SamplerArgs {
  int64_t B;
  int it, n, n_new;                // row length this iteration, new samples the previous step produced
  int N_eval, N_final, max_iters, beta_iters, force_iters;
  float eps, add_tiny, near, far;
  const float* beta_param; float beta_min;
  int* state;
  // rows (capacity NMAX per ray)
  float* z_cur; float* z_nxt;      // (B,NMAX) sorted samples: current / after this iteration's merge
  float* sdf_prev; float* sdf_cur; // (B,NMAX) sdf of the previous row / merged sdf of the current row
  const int* idx;                  // (B,NMAX) source index of row position (from the previous merge)
  int* idx_nxt;
  const float* sdf_new;            // (B,n_new) sdf at the newest samples
  float* samples;                  // (B,NNEW) out: new samples
  float* beta;                     // (B)
  const float* u_more; const float* u_final; int64_t ldu_final;   // u_final: (N_final) shared (ldu=0) or (B,N_final)
};

template <int E>
__device__ __forceinline__ void load_row(Row<E>& r, const float* z, const float* s, int n, int lane) {
  r.n = n;
#pragma unroll
  for (int e = 0; e < E; ++e) {
    const int j = lane * E + e;
    r.z[e] = (j < n) ? z[j] : 3.0e38f;
    r.s[e] = (j < n) ? s[j] : 0.f;
  }
}

// ---- step A: merged sdf row, d*, beta line search (ray_sampler.py:88-132) + batch flag (:151) ----------------
// fills r with the merged row of ray rc (and its intervals), stores the merged sdf row, returns the ray's new beta
template <int E>
__device__ __forceinline__ float beta_step(const SamplerArgs& a, Row<E>& r, int64_t rc, bool active, int lane) {
  const float beta0 = fabsf(a.beta_param[0]) + a.beta_min;
  const int n = a.n;
  // merge: sdf_cur[j] = idx[j] < n_old ? sdf_prev[idx[j]] : sdf_new[idx[j]-n_old]   (ray_sampler.py:90-95)
  float* sc = a.sdf_cur + rc * NMAX;
  const float* zc = a.z_cur + rc * NMAX;
  r.n = n;
  {
    const int n_old = n - a.n_new;
    const int* ix = a.idx + rc * NMAX;
    const float* sp = a.sdf_prev + rc * NMAX;
    const float* sn = a.sdf_new + rc * a.n_new;
#pragma unroll
    for (int e = 0; e < E; ++e) {
      const int j = lane * E + e;
      float sv = 0.f, zv = 3.0e38f;
      if (j < n) {
        zv = zc[j];
        if (a.it == 0) sv = sn[j];
        else { const int k = ix[j]; sv = (k < n_old) ? sp[k] : sn[k - n_old]; }
        if (active) sc[j] = sv;
      }
      r.z[e] = zv; r.s[e] = sv;
    }
  }
  row_intervals<E>(r, lane);
  float beta = a.beta[rc];
  const float err0 = row_error_bound<E>(r, beta0, lane);
  if (err0 <= a.eps) beta = beta0;
  float bmin = beta0, bmax = beta;
  for (int k = 0; k < a.beta_iters; ++k) {
    const float bmid = (bmin + bmax) / 2.0f;
    const float err = row_error_bound<E>(r, bmid, lane);
    if (err <= a.eps) bmax = bmid; else bmin = bmid;
  }
  return bmax;
}
template <int E>
__global__ __launch_bounds__(256) void sampler_beta_kernel(SamplerArgs a) {
  if (a.state[ST_DONE]) return;
  const int lane = threadIdx.x & 63;
  const int64_t ray = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (ray >= a.B) return;
  Row<E> r;
  const float beta = beta_step<E>(a, r, ray, true, lane);
  if (lane == 0) {
    a.beta[ray] = beta;
    if (beta > fabsf(a.beta_param[0]) + a.beta_min) atomicOr(&a.state[ST_FLAG0 + a.it], 1);
  }
}

// ---- step B: density/transmittance, pdf, inverse CDF, merge (ray_sampler.py:139-212) --------------------------
// the row r of ray rc (intervals computed) and its beta -> new samples, and for `more` the merged row of the next iteration
template <int E>
__device__ __forceinline__ void resample_step(const SamplerArgs& a, const Row<E>& r, float beta, bool more, int64_t ray, int64_t rc, bool active,
                                              float (&s_cdf)[4][NMAX + 1], float (&s_z)[4][NMAX], float (&s_new)[4][NNEW], int lane, int wv) {
  const int n = a.n;
  const float ib = 1.0f / beta;
  // free energy with the 1e10 tail, transmittance, weights (ray_sampler.py:139-147)
  float fe[E], cum[E];
#pragma unroll
  for (int e = 0; e < E; ++e) {
    const int j = lane * E + e;
    const float dist = (j < n - 1) ? r.a[e] : 1.0e10f;
    fe[e] = (j < n) ? dist * laplace_density(r.s[e], ib) : 0.f;
    cum[e] = (j < n - 1) ? fe[e] : 0.f;            // shifted free energy only ever uses fe[:-1]
  }
  row_scan_incl<E>(cum, lane);
  float pdf[E];
  if (more) {
    float er[E];
#pragma unroll
    for (int e = 0; e < E; ++e) {
      const int j = lane * E + e;
      er[e] = (j < n - 1) ? expf(-r.ds[e] / beta) * (r.a[e] * r.a[e]) / (4.0f * beta * beta) : 0.f;
    }
    row_scan_incl<E>(er, lane);
#pragma unroll
    for (int e = 0; e < E; ++e) {
      const int j = lane * E + e;
      const float T = expf(-(cum[e] - ((j < n - 1) ? fe[e] : 0.f)));
      pdf[e] = (j < n - 1) ? (fminf(expf(er[e]), 1.0e6f) - 1.0f) * T + a.add_tiny : 0.f;
    }
  } else {
#pragma unroll
    for (int e = 0; e < E; ++e) {
      const int j = lane * E + e;
      const float T = expf(-(cum[e] - ((j < n - 1) ? fe[e] : 0.f)));
      const float w = (1.0f - expf(-fe[e])) * T;
      pdf[e] = (j < n - 1) ? w + 1e-5f : 0.f;
    }
  }
  float tot = 0.f;
#pragma unroll
  for (int e = 0; e < E; ++e) tot += pdf[e];
  tot = wsum(tot);
  float cd[E];
#pragma unroll
  for (int e = 0; e < E; ++e) { pdf[e] = pdf[e] / tot; cd[e] = pdf[e]; }
  row_scan_incl<E>(cd, lane);
  // cdf = [0, cumsum(pdf)] : cdf[j] = exclusive prefix, j = 0..n-1
#pragma unroll
  for (int e = 0; e < E; ++e) {
    const int j = lane * E + e;
    if (j < n) { s_cdf[wv][j] = cd[e] - pdf[e]; s_z[wv][j] = r.z[e]; }
  }
  __syncthreads();
  // inverse CDF (ray_sampler.py:186-207)
  const int N = more ? a.N_eval : a.N_final;
  for (int k = lane; k < N; k += 64) {
    float u;
    if (more) u = a.u_more[k];
    else u = a.u_final[rc * a.ldu_final + k];
    int lo = 0, hi = n;                                   // first index with cdf > u  (searchsorted right=True)
    while (lo < hi) {
      const int mid = (lo + hi) >> 1;
      if (s_cdf[wv][mid] <= u) lo = mid + 1; else hi = mid;
    }
    const int below = (lo - 1 > 0) ? lo - 1 : 0;
    const int above = (lo < n - 1) ? lo : n - 1;
    const float c0 = s_cdf[wv][below], c1 = s_cdf[wv][above];
    const float b0 = s_z[wv][below], b1 = s_z[wv][above];
    float denom = c1 - c0;
    if (denom < 1e-5f) denom = 1.0f;
    const float t = (u - c0) / denom;
    const float smp = b0 + t * (b1 - b0);
    s_new[wv][k] = smp;
    if (active) a.samples[ray * NNEW + k] = smp;
  }
  __syncthreads();
  if (!more) return;
  // sorted merge of z (n) and the N new samples -> z_nxt (n+N), idx_nxt (ray_sampler.py:211-212); stable: old first
  if (active) {
    float* zn = a.z_nxt + ray * NMAX;
    int* ixn = a.idx_nxt + ray * NMAX;
    for (int j = lane; j < n; j += 64) {
      const float v = s_z[wv][j];
      int lo = 0, hi = N;                                 // # new samples strictly below v
      while (lo < hi) { const int mid = (lo + hi) >> 1; if (s_new[wv][mid] < v) lo = mid + 1; else hi = mid; }
      zn[j + lo] = v; ixn[j + lo] = j;
    }
    for (int k = lane; k < N; k += 64) {
      const float v = s_new[wv][k];
      int lo = 0, hi = n;                                 // # old samples <= v
      while (lo < hi) { const int mid = (lo + hi) >> 1; if (s_z[wv][mid] <= v) lo = mid + 1; else hi = mid; }
      zn[k + lo] = v; ixn[k + lo] = n + k;
    }
  }
}

// The last workgroup to finish step B of the iteration that produced the final samples raises ST_DONE: every workgroup of this
// launch has finished by then, and the launches behind it (the remaining iterations, enqueued up front) return at once.
__device__ __forceinline__ void iteration_done(const SamplerArgs& a, bool more) {
  __syncthreads();
  if (threadIdx.x == 0) {
    __threadfence();
    const int c = atomicAdd(&a.state[ST_CNT0 + a.it], 1);
    if (c == (int)gridDim.x - 1 && !more) { a.state[ST_ITERS] = a.it + 1; __threadfence(); a.state[ST_DONE] = 1; }
  }
}

template <int E>
__global__ __launch_bounds__(256) void sampler_resample_kernel(SamplerArgs a) {
  __shared__ float s_cdf[4][NMAX + 1];
  __shared__ float s_z[4][NMAX];
  __shared__ float s_new[4][NNEW];
  if (a.state[ST_DONE]) return;
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
  const int64_t ray = (int64_t)blockIdx.x * 4 + wv;
  const bool active = ray < a.B;
  const int64_t rc = active ? ray : a.B - 1;
  bool not_conv = a.state[ST_FLAG0 + a.it] != 0;
  if (a.force_iters > 0) not_conv = (a.it + 1) < a.force_iters;
  const bool more = not_conv && (a.it + 1 < a.max_iters);
  Row<E> r;
  load_row<E>(r, a.z_cur + rc * NMAX, a.sdf_cur + rc * NMAX, a.n, lane);
  row_intervals<E>(r, lane);
  resample_step<E>(a, r, a.beta[rc], more, ray, rc, active, s_cdf, s_z, s_new, lane, wv);
  iteration_done(a, more);
}

// Steps A and B in one launch when the iteration count is fixed by the caller (force_iters > 0): the batch-global test
// `beta.max() > beta0` (ray_sampler.py:151), which needs every ray's step A before any ray's step B, is not consulted then, and
// the row stays in registers between the steps.
template <int E>
__global__ __launch_bounds__(256) void sampler_iter_fixed_kernel(SamplerArgs a) {
  __shared__ float s_cdf[4][NMAX + 1];
  __shared__ float s_z[4][NMAX];
  __shared__ float s_new[4][NNEW];
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
  const int64_t ray = (int64_t)blockIdx.x * 4 + wv;
  const bool active = ray < a.B;
  const int64_t rc = active ? ray : a.B - 1;
  const bool more = ((a.it + 1) < a.force_iters) && (a.it + 1 < a.max_iters);
  Row<E> r;
  const float beta = beta_step<E>(a, r, rc, active, lane);
  if (active && lane == 0) a.beta[ray] = beta;
  resample_step<E>(a, r, beta, more, ray, rc, active, s_cdf, s_z, s_new, lane, wv);
  iteration_done(a, more);
}

// ---- initial uniform / stratified samples + Lemma-2 beta (ray_sampler.py:22-43, 75-77) ------------------------
__global__ __launch_bounds__(256) void sampler_init_kernel(int64_t B, int N, const float* __restrict__ t_lin, const float* __restrict__ strat_u,
                                                            float near, float far, float eps, float* __restrict__ z_cur,
                                                            float* __restrict__ samples, float* __restrict__ beta, int* __restrict__ state) {
  const int lane = threadIdx.x & 63;
  const int64_t ray = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (blockIdx.x == 0 && threadIdx.x < ST_WORDS) state[threadIdx.x] = 0;
  if (ray >= B) return;
  float acc = 0.f, carry = 0.f;
  for (int base = 0; base < N; base += 64) {
    const int j = base + lane;
    float z = 0.f;
    if (j < N) {
      const float t = t_lin[j];
      z = near * (1.0f - t) + far * t;
      if (strat_u) {
        const float tm = t_lin[j > 0 ? j - 1 : 0], tp = t_lin[j < N - 1 ? j + 1 : N - 1];
        const float zm = near * (1.0f - tm) + far * tm, zq = near * (1.0f - tp) + far * tp;
        const float lower = (j > 0) ? 0.5f * (z + zm) : z;
        const float upper = (j < N - 1) ? 0.5f * (zq + z) : z;
        z = lower + (upper - lower) * strat_u[ray * N + j];
      }
      z_cur[ray * NMAX + j] = z;
      samples[ray * NNEW + j] = z;
    }
    float zp = __shfl_up(z, 1);
    if (lane == 0) zp = carry;
    if (j < N && j > 0) { const float d = z - zp; acc = fmaf(d, d, acc); }
    carry = __shfl(z, 63);
  }
  acc = wsum(acc);
  if (lane == 0) beta[ray] = sqrtf((1.0f / (4.0f * logf(eps + 1.0f))) * acc);
}

// ---- epilogue: z = sort([samples(N_final), near, far, z_row[extra_idx]]) ; eikonal sample (ray_sampler.py:215-234) -----
__global__ __launch_bounds__(256) void sampler_final_kernel(int64_t B, const int* __restrict__ state, int N_eval, const float* z_a,
                                                             const float* z_b, const float* __restrict__ samples, int N_final, int N_extra,
                                                             const int* __restrict__ extra_idx, const int* __restrict__ extra_tab,
                                                             float near, float far, const int* __restrict__ eik_idx, float* __restrict__ z_out,
                                                             int64_t ldz, float* __restrict__ z_eik, int* __restrict__ iters_out) {
  __shared__ float buf[4][256];
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
  const int64_t ray = (int64_t)blockIdx.x * 4 + wv;
  const bool active = ray < B;
  const int64_t rc = active ? ray : B - 1;
  const int iters = state[ST_ITERS];
  if (iters_out != nullptr && blockIdx.x == 0 && threadIdx.x == 0) iters_out[0] = iters;      // the caller's copy of the iteration count
  const int n_row = N_eval * iters;
  const float* zr = ((iters & 1) ? z_a : z_b) + rc * NMAX;       // row used by the last executed iteration
  const int total = N_final + 2 + N_extra;
  for (int k = lane; k < 256; k += 64) {
    float v = 3.0e38f;
    if (k < N_final) v = samples[rc * NNEW + k];
    else if (k == N_final) v = near;
    else if (k == N_final + 1) v = far;
    else if (k < total) {
      const int e = k - N_final - 2;
      int id;
      if (extra_idx) id = extra_idx[(iters - 1) * N_extra + e];      // train: randperm(n)[:N_extra] per possible row length, shared by all rays
      else id = extra_tab[(iters - 1) * N_extra + e];               // eval: linspace(0, n-1, N_extra).long(), tabulated per row length
      id = id < 0 ? 0 : (id > n_row - 1 ? n_row - 1 : id);
      v = zr[id];
    }
    buf[wv][k] = v;
  }
  __syncthreads();
  // bitonic sort of 256 keys, 4 per lane through LDS (tiny; one wave per ray)
  for (int k = 2; k <= 256; k <<= 1) {
    for (int j = k >> 1; j > 0; j >>= 1) {
      for (int i = lane; i < 256; i += 64) {
        const int l = i ^ j;
        if (l > i) {
          const float x = buf[wv][i], y = buf[wv][l];
          const bool up = (i & k) == 0;
          if ((x > y) == up) { buf[wv][i] = y; buf[wv][l] = x; }
        }
      }
      __syncthreads();
    }
  }
  if (active) {
    for (int k = lane; k < total; k += 64) z_out[ray * ldz + k] = buf[wv][k];
    if (lane == 0 && z_eik) {
      int id = eik_idx ? eik_idx[ray] : 0;
      id = id < 0 ? 0 : (id > total - 1 ? total - 1 : id);
      z_eik[ray] = buf[wv][id];
    }
  }
}

// ---- get_error_bound as its own entry (ray_sampler.py:243-251) with the Theorem-1 d* (:99-114): the same device functions
// the Algorithm-1 kernels above use, on caller-provided rows
template <int E>
__global__ __launch_bounds__(256) void error_bound_kernel(int64_t B, int n, const float* __restrict__ z, const float* __restrict__ sdf,
                                                           const float* __restrict__ beta, int64_t ldbeta, const float* __restrict__ d_star_in,
                                                           float* __restrict__ d_star_out, float* __restrict__ bound) {
  const int lane = threadIdx.x & 63;
  const int64_t ray = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (ray >= B) return;
  Row<E> r;
  load_row<E>(r, z + ray * n, sdf + ray * n, n, lane);
  row_intervals<E>(r, lane);
#pragma unroll
  for (int e = 0; e < E; ++e) {
    const int j = lane * E + e;
    if (j < n - 1) {
      if (d_star_in) r.ds[e] = d_star_in[ray * (n - 1) + j];
      if (d_star_out) d_star_out[ray * (n - 1) + j] = r.ds[e];
    }
  }
  if (bound) {
    const float eb = row_error_bound<E>(r, beta[ray * ldbeta], lane);
    if (lane == 0) bound[ray] = eb;
  }
}

template <int E>
int launch_iter(const SamplerArgs& a, hipStream_t st, const i2sdf_exchange* ex) {
  const unsigned grid = (unsigned)((a.B + 3) / 4);
  if (a.force_iters > 0 && !ex) {          // fixed iteration count: one launch per iteration
    sampler_iter_fixed_kernel<E><<<grid, 256, 0, st>>>(a);
    return I2SDF_OK;
  }
  sampler_beta_kernel<E><<<grid, 256, 0, st>>>(a);
  // 1-GPU-equivalent data parallelism: `beta.max() > beta0` (ray_sampler.py:151) over the rays of ALL ranks
  if (ex) {
    const int rc = ex->allreduce(ex->ctx, a.state + ST_FLAG0 + a.it, 1, I2SDF_XCHG_I32, I2SDF_XCHG_MAX, (void*)st);
    if (rc) return rc;
  }
  sampler_resample_kernel<E><<<grid, 256, 0, st>>>(a);
  return I2SDF_OK;
}

}  // namespace

extern "C" int64_t i2sdf_sampler_workspace_floats(int64_t B) {
  // z rows x2, sdf rows x2, idx rows x2 (as ints), samples, sdf_new, beta, state
  return B * (int64_t)(NMAX * 6 + NNEW * 2 + 1) + ST_WORDS + 64;
}

extern "C" int i2sdf_error_bound(const float* z, const float* sdf, int64_t B, int32_t n, const float* beta, int64_t ldbeta,
                                 const float* d_star_in, float* d_star_out, float* bound, void* stream) {
  if (B == 0) return I2SDF_OK;
  if (!z || !sdf || B < 0 || n < 2 || n > NMAX || (bound && !beta) || (!bound && !d_star_out) || (ldbeta != 0 && ldbeta != 1)) return I2SDF_EINVAL;
  hipStream_t st = (hipStream_t)stream;
  const unsigned grid = (unsigned)((B + 3) / 4);
#define I2SDF_EB(E_) error_bound_kernel<E_><<<grid, 256, 0, st>>>(B, n, z, sdf, beta, ldbeta, d_star_in, d_star_out, bound)
  switch (cdiv(n, 64)) {
    case 1: I2SDF_EB(1); break;
    case 2: I2SDF_EB(2); break;
    case 3: I2SDF_EB(3); break;
    case 4: I2SDF_EB(4); break;
    case 5: I2SDF_EB(5); break;
    case 6: I2SDF_EB(6); break;
    case 7: I2SDF_EB(7); break;
    case 8: I2SDF_EB(8); break;
    case 9: I2SDF_EB(9); break;
    default: I2SDF_EB(10); break;
  }
#undef I2SDF_EB
  return i2sdf_hip_check(hipGetLastError(), "error_bound launch");
}

int i2sdf_sdf_forward_rays_flagged(const i2sdf_plan* p, const float* packed, const float* cam, const float* dirs, const float* z, int64_t ldz,
                                   int32_t n_per_ray, int64_t B, float* sdf_out, const int* skip_flag, void* stream);

extern "C" int i2sdf_sample_rays(const i2sdf_plan* p, const float* packed, const float* params, const i2sdf_sampler_cfg* sc,
                                 const float* cam, const float* dirs, int64_t B, int32_t training, const float* t_lin, const float* u_more,
                                 const float* u_final, int64_t ldu_final, const int32_t* extra_tab, const float* strat_u,
                                 const int32_t* extra_idx, const int32_t* eik_idx, int32_t force_iters, float* workspace, float* z_out,
                                 int64_t ldz, float* z_eik, int32_t* iters_out, void* stream) {
  if (B == 0) return I2SDF_OK;
  if (!p || !packed || !params || !sc || !cam || !dirs || !t_lin || !u_more || !u_final || !workspace || !z_out || B < 0) return I2SDF_EINVAL;
  if (sc->N_samples_eval > NNEW || sc->N_samples > NNEW || sc->N_samples_eval * sc->max_total_iters > NMAX || sc->max_total_iters > 12)
    return I2SDF_EINVAL;
  if (sc->N_samples + 2 + sc->N_samples_extra > 256 || ldz < sc->N_samples + 2 + sc->N_samples_extra) return I2SDF_EINVAL;
  if (training && (!strat_u || (sc->N_samples_extra > 0 && !extra_idx))) return I2SDF_EINVAL;
  if (!training && sc->N_samples_extra > 0 && !extra_tab) return I2SDF_EINVAL;
  hipStream_t st = (hipStream_t)stream;
  float* w = workspace;
  float* zA = w; w += B * NMAX;
  float* zB = w; w += B * NMAX;
  float* sA = w; w += B * NMAX;
  float* sB = w; w += B * NMAX;
  int* iA = (int*)w; w += B * NMAX;
  int* iB = (int*)w; w += B * NMAX;
  float* samples = w; w += B * NNEW;
  float* sdf_new = w; w += B * NNEW;
  float* beta = w; w += B;
  int* state = (int*)w;
  const float near = sc->near, far = 2.0f * p->desc.scene_bounding_sphere;
  const unsigned grid = (unsigned)((B + 3) / 4);
  // the batch-global OR over ranks exists for TRAINING batches only: eval calls (validation on one rank, render_image chunks dealt
  // unevenly to the ranks, bubble sweeps) are rank-local and must not enter a collective that the other ranks never issue
  const i2sdf_exchange* gcomm = (training && p->exchange.allreduce && (p->dp_flags & I2SDF_DP_GLOBAL_SAMPLER) && force_iters <= 0) ? &p->exchange : nullptr;
  sampler_init_kernel<<<grid, 256, 0, st>>>(B, sc->N_samples_eval, t_lin, training ? strat_u : nullptr, near, far, sc->eps, zA, samples, beta, state);
  // With a fixed iteration count the host knows where the loop ends; otherwise every iteration is enqueued and the ones after
  // convergence return immediately (device flag, no host synchronisation).
  const int n_it = (force_iters > 0 && force_iters < sc->max_total_iters) ? force_iters : sc->max_total_iters;
  for (int it = 0; it < n_it; ++it) {
    const int n_new = sc->N_samples_eval;
    int rc = i2sdf_sdf_forward_rays_flagged(p, packed, cam, dirs, samples, NNEW, n_new, B, sdf_new, state + ST_DONE, stream);
    if (rc) return rc;
    SamplerArgs a{};
    a.B = B; a.it = it; a.n = sc->N_samples_eval * (it + 1); a.n_new = n_new;
    a.N_eval = sc->N_samples_eval; a.N_final = sc->N_samples; a.max_iters = sc->max_total_iters; a.beta_iters = sc->beta_iters;
    a.force_iters = force_iters;
    a.eps = sc->eps; a.add_tiny = sc->add_tiny; a.near = near; a.far = far;
    a.beta_param = params + p->desc.off_beta; a.beta_min = p->desc.beta_min;
    a.state = state;
    const bool even = (it & 1) == 0;
    a.z_cur = even ? zA : zB; a.z_nxt = even ? zB : zA;
    a.sdf_cur = even ? sA : sB; a.sdf_prev = even ? sB : sA;
    a.idx = even ? iB : iA; a.idx_nxt = even ? iA : iB;       // idx written by iteration it-1 lives in its idx_nxt
    a.sdf_new = sdf_new; a.samples = samples; a.beta = beta;
    a.u_more = u_more; a.u_final = u_final; a.ldu_final = ldu_final;
    const int E = cdiv(a.n, 64);
    int xrc = I2SDF_OK;
    switch (E) {
      case 1: xrc = launch_iter<1>(a, st, gcomm); break;
      case 2: xrc = launch_iter<2>(a, st, gcomm); break;
      case 3: xrc = launch_iter<3>(a, st, gcomm); break;
      case 4: xrc = launch_iter<4>(a, st, gcomm); break;
      case 5: xrc = launch_iter<5>(a, st, gcomm); break;
      case 6: xrc = launch_iter<6>(a, st, gcomm); break;
      case 7: xrc = launch_iter<7>(a, st, gcomm); break;
      case 8: xrc = launch_iter<8>(a, st, gcomm); break;
      case 9: xrc = launch_iter<9>(a, st, gcomm); break;
      default: xrc = launch_iter<10>(a, st, gcomm); break;
    }
    if (xrc) return xrc;
  }
  sampler_final_kernel<<<grid, 256, 0, st>>>(B, state, sc->N_samples_eval, zA, zB, samples, sc->N_samples, sc->N_samples_extra,
                                             training ? (const int*)extra_idx : nullptr, (const int*)extra_tab, near, far, (const int*)eik_idx, z_out, ldz,
                                             z_eik, iters_out);
  return i2sdf_hip_check(hipGetLastError(), "sample_rays launch");
}
