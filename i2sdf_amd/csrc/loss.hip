// I2SDFLoss forward + gradient w.r.t. the render outputs in two tiny launches:
// model/network/__init__.py:289-406 with its quirks (angular term = the same L1 normal term, :368-369).
// Replaces ~100 element-wise torch kernels (forward and autograd backward) per training step.
#include <algorithm>
#include "plan.h"

using namespace i2sdf;

int i2sdf_hip_check(hipError_t e, const char* what);

namespace {

enum { S_RGB = 0, S_EIK, S_SMOOTH, S_MASK, S_DEPTH, S_DEPTH_CNT, S_NORMAL, S_NORMAL_CNT, S_BUBBLE, S_LIGHT, S_N };
// denominators of the means: local values, or (data parallel, i2sdf_loss_cfg.exchange) their mean over the ranks
enum { C_B = 0, C_NPC, C_DEPTH, C_NORMAL, C_N };
constexpr int LOSS_BLOCKS = 64;

struct LossArgs {
  i2sdf_loss_cfg c;
  int64_t B, n_pc;
  const float *rgb, *depth, *wsum, *normal, *grad_theta, *diff_norm, *surface, *lmask;
  const float *gt_rgb, *gt_depth, *gt_normal, *gt_mask, *gt_lmask;
  const uint8_t *depth_mask, *normal_mask;
  float* partial;      // (LOSS_BLOCKS, S_N)
  float* sums;         // (S_N)   only written / read on the data-parallel path (reduced = 1)
  float* cnt;          // (C_N)   likewise: the denominators the exchange hook averages over the ranks
  int nb;              // workgroups of the reduction launch = rows of `partial`
  int reduced;         // 1: sums / cnt are in memory (loss_reduce_kernel + exchange ran); 0: every workgroup of the gradient launch adds the
                       //    block partials up itself, in block order -- no counter, no state in the scratch, nothing to initialise
  float* losses;       // (10): loss, rgb, eikonal, smooth, mask, depth, normal, angular, bubble, light_mask
  float* loss_value;   // (1) | NULL: the total once more, as a tensor of its own
  float *g_rgb, *g_depth, *g_wsum, *g_normal, *g_grad_theta, *g_diff_norm, *g_surface, *g_lmask;
};

__device__ __forceinline__ float bce(float p_raw, float y, float& dp) {
  const float p = fminf(fmaxf(p_raw, 1e-3f), 1.0f - 1e-3f);
  const bool inside = p_raw >= 1e-3f && p_raw <= 1.0f - 1e-3f;
  const float l = -(y * fmaxf(logf(p), -100.f) + (1.0f - y) * fmaxf(logf(1.0f - p), -100.f));   // F.binary_cross_entropy clamps log at -100
  dp = inside ? (-(y / p) + (1.0f - y) / (1.0f - p)) : 0.f;
  return l;
}

__global__ __launch_bounds__(256) void loss_partial_kernel(LossArgs a) {
  float s[S_N];
#pragma unroll
  for (int i = 0; i < S_N; ++i) s[i] = 0.f;
  const int64_t stride = (int64_t)gridDim.x * 256;
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < a.B; i += stride) {
#pragma unroll
    for (int k = 0; k < 3; ++k) s[S_RGB] += fabsf(a.rgb[i * 3 + k] - a.gt_rgb[i * 3 + k]);
    if (a.diff_norm) s[S_SMOOTH] += a.diff_norm[i];
    if (a.gt_mask) { float d; s[S_MASK] += bce(a.wsum[i], a.gt_mask[i], d); }
    if (a.gt_depth) {
      const float m = a.depth_mask[i] ? 1.f : 0.f, d = a.depth[i] - a.gt_depth[i];
      s[S_DEPTH] += m * d * d; s[S_DEPTH_CNT] += m;
    }
    if (a.gt_normal && a.normal) {
      const float m = a.normal_mask[i] ? 1.f : 0.f;
      const float dot = a.normal[i * 3] * a.gt_normal[i * 3] + a.normal[i * 3 + 1] * a.gt_normal[i * 3 + 1] + a.normal[i * 3 + 2] * a.gt_normal[i * 3 + 2];
      s[S_NORMAL] += m * fabsf(1.0f - dot); s[S_NORMAL_CNT] += m;
    }
    if (a.lmask && a.gt_lmask) { float d; s[S_LIGHT] += bce(a.lmask[i], a.gt_lmask[i], d); }
  }
  if (a.grad_theta)
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < 2 * a.B; i += stride) {
      const float x = a.grad_theta[i * 3], y = a.grad_theta[i * 3 + 1], z = a.grad_theta[i * 3 + 2];
      const float d = sqrtf(x * x + y * y + z * z) - 1.0f;
      s[S_EIK] += d * d;
    }
  if (a.surface)
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < a.n_pc; i += stride) s[S_BUBBLE] += fabsf(a.surface[i]);
  __shared__ float sm[4][S_N];
#pragma unroll
  for (int k = 0; k < S_N; ++k) {
    float v = s[k];
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
    if ((threadIdx.x & 63) == 0) sm[threadIdx.x >> 6][k] = v;
  }
  __syncthreads();
  if (threadIdx.x < S_N) a.partial[blockIdx.x * S_N + threadIdx.x] = sm[0][threadIdx.x] + sm[1][threadIdx.x] + sm[2][threadIdx.x] + sm[3][threadIdx.x];
}

// The block partials added up in BLOCK ORDER (so the sums do not depend on any schedule) + the local denominators.  Round 4 had the
// last-arriving workgroup of the reduction launch do this behind an atomic arrival counter kept in the caller's scratch: a scratch that
// was not zeroed once, an aborted launch or two streams sharing one scratch left the counter off and the sums silently stale (ADVICE
// r4).  Now nothing persists between calls: every workgroup of the gradient launch runs this itself (<= 64 x 10 floats from L2), or --
// data parallel, where the denominators go through the exchange hook between the launches -- one workgroup of loss_reduce_kernel.
__device__ __forceinline__ void loss_reduce(const LossArgs& a, float (&tot)[S_N], float (&cnt)[C_N]) {
  if (threadIdx.x < S_N) {
    float v = 0.f;
    for (int b = 0; b < a.nb; ++b) v += a.partial[b * S_N + threadIdx.x];
    tot[threadIdx.x] = v;
    if (threadIdx.x == S_DEPTH_CNT) cnt[C_DEPTH] = v;
    if (threadIdx.x == S_NORMAL_CNT) cnt[C_NORMAL] = v;
  }
  if (threadIdx.x == 0) { cnt[C_B] = (float)a.B; cnt[C_NPC] = (float)a.n_pc; }
}

__global__ __launch_bounds__(64) void loss_reduce_kernel(LossArgs a) {
  __shared__ float tot[S_N], cnt[C_N];
  loss_reduce(a, tot, cnt);
  __syncthreads();
  if (threadIdx.x < S_N) a.sums[threadIdx.x] = tot[threadIdx.x];
  if (threadIdx.x < C_N) a.cnt[threadIdx.x] = cnt[threadIdx.x];
}

// the reported values (workgroup 0 of the gradient launch)
__device__ __forceinline__ void loss_finalize(const LossArgs& a, const float (&tot)[S_N], const float (&cnt)[C_N]) {
  if (threadIdx.x == 0) {
    const float B = cnt[C_B];
    const float rgb = tot[S_RGB] / (3.0f * B);
    const float eik = a.grad_theta ? tot[S_EIK] / (2.0f * B) : 0.f;
    const float smooth = (a.diff_norm && a.c.smooth_on && a.c.smooth_w > 0.f) ? tot[S_SMOOTH] / B : 0.f;
    const float mask = (a.gt_mask && a.c.mask_w > 0.f) ? tot[S_MASK] / B : 0.f;
    const float depth = (a.gt_depth && a.c.depth_w > 0.f) ? tot[S_DEPTH] / cnt[C_DEPTH] : 0.f;
    const float nl1 = (a.gt_normal && a.normal) ? tot[S_NORMAL] / cnt[C_NORMAL] : 0.f;
    const float normal = a.c.normal_w > 0.f ? nl1 : 0.f, angular = a.c.angular_w > 0.f ? nl1 : 0.f;
    const float bubble = (a.surface && a.c.bubble_w > 0.f) ? tot[S_BUBBLE] / cnt[C_NPC] : 0.f;
    const float light = (a.lmask && a.gt_lmask && a.c.light_w > 0.f) ? tot[S_LIGHT] / B : 0.f;
    a.losses[0] = rgb + a.c.eikonal_w * eik + a.c.smooth_w * smooth + a.c.mask_w * mask + a.c.depth_w * depth + a.c.normal_w * normal +
                  a.c.angular_w * angular + a.c.bubble_w * bubble + a.c.light_w * light;
    a.losses[1] = rgb; a.losses[2] = eik; a.losses[3] = smooth; a.losses[4] = mask; a.losses[5] = depth;
    a.losses[6] = normal; a.losses[7] = angular; a.losses[8] = bubble; a.losses[9] = light;
    if (a.loss_value) a.loss_value[0] = a.losses[0];
  }
}

__global__ __launch_bounds__(256) void loss_grad_kernel(LossArgs a) {
  __shared__ float tot[S_N], cnt[C_N];
  if (a.reduced) {
    if (threadIdx.x < S_N) tot[threadIdx.x] = a.sums[threadIdx.x];
    if (threadIdx.x < C_N) cnt[threadIdx.x] = a.cnt[threadIdx.x];
  } else {
    loss_reduce(a, tot, cnt);
  }
  __syncthreads();
  if (blockIdx.x == 0) loss_finalize(a, tot, cnt);
  const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
  const float B = cnt[C_B];
  if (i < a.B) {
#pragma unroll
    for (int k = 0; k < 3; ++k) {
      const float d = a.rgb[i * 3 + k] - a.gt_rgb[i * 3 + k];
      a.g_rgb[i * 3 + k] = (d > 0.f ? 1.f : (d < 0.f ? -1.f : 0.f)) / (3.0f * B);
    }
    float gd = 0.f;
    if (a.gt_depth && a.c.depth_w > 0.f && a.depth_mask[i]) gd = a.c.depth_w * 2.0f * (a.depth[i] - a.gt_depth[i]) / cnt[C_DEPTH];
    a.g_depth[i] = gd;
    float gw = 0.f;
    if (a.gt_mask && a.c.mask_w > 0.f) { float d; (void)bce(a.wsum[i], a.gt_mask[i], d); gw = a.c.mask_w * d / B; }
    a.g_wsum[i] = gw;
    if (a.g_normal) {
      float g0 = 0.f, g1 = 0.f, g2 = 0.f;
      if (a.gt_normal && a.normal && a.normal_mask[i]) {
        const float w = ((a.c.normal_w > 0.f ? a.c.normal_w : 0.f) + (a.c.angular_w > 0.f ? a.c.angular_w : 0.f)) / cnt[C_NORMAL];
        const float n0 = a.gt_normal[i * 3], n1 = a.gt_normal[i * 3 + 1], n2 = a.gt_normal[i * 3 + 2];
        const float u = 1.0f - (a.normal[i * 3] * n0 + a.normal[i * 3 + 1] * n1 + a.normal[i * 3 + 2] * n2);
        const float sg = u > 0.f ? -1.f : (u < 0.f ? 1.f : 0.f);          // d|1-dot| / d dot
        g0 = w * sg * n0; g1 = w * sg * n1; g2 = w * sg * n2;
      }
      a.g_normal[i * 3] = g0; a.g_normal[i * 3 + 1] = g1; a.g_normal[i * 3 + 2] = g2;
    }
    if (a.g_diff_norm) a.g_diff_norm[i] = (a.c.smooth_on && a.c.smooth_w > 0.f) ? a.c.smooth_w / B : 0.f;
    if (a.g_lmask) {
      float gl = 0.f;
      if (a.lmask && a.gt_lmask && a.c.light_w > 0.f) { float d; (void)bce(a.lmask[i], a.gt_lmask[i], d); gl = a.c.light_w * d / B; }
      a.g_lmask[i] = gl;
    }
  }
  if (a.g_grad_theta && i < 2 * a.B) {
    const float x = a.grad_theta[i * 3], y = a.grad_theta[i * 3 + 1], z = a.grad_theta[i * 3 + 2];
    const float nrm = sqrtf(x * x + y * y + z * z);
    const float f = nrm > 0.f ? a.c.eikonal_w * 2.0f * (nrm - 1.0f) / (nrm * 2.0f * B) : 0.f;
    a.g_grad_theta[i * 3] = f * x; a.g_grad_theta[i * 3 + 1] = f * y; a.g_grad_theta[i * 3 + 2] = f * z;
  }
  if (a.g_surface && i < a.n_pc) {
    const float sv = a.surface[i];
    a.g_surface[i] = a.c.bubble_w > 0.f ? a.c.bubble_w * (sv > 0.f ? 1.f : (sv < 0.f ? -1.f : 0.f)) / cnt[C_NPC] : 0.f;
  }
}

// ---------------------------------------------------------------------------------------------------------------
// Eikonal / smoothness outputs of the network (model/network/__init__.py:188-193): from the gradients of the 3B extra points
// [B uniform | B near-surface | B neighbours]   grad_theta = rows [0, 2B),
//   diff_norm[i] = || normalize(g[B+i]) - normalize(g[2B+i]) ||_2,   normalize(v) = v / max(||v||, 1e-6)  (F.normalize eps)
// and the backward of both in one launch.  In torch these are ~8 forward and ~25 autograd-backward kernels on (2B,3) tensors.
// Conventions of the torch backward formulas are kept: d||x||/dx = 0 at x = 0; no gradient through ||v|| where ||v|| < eps.
// ---------------------------------------------------------------------------------------------------------------
constexpr float NRM_EPS = 1e-6f;
// No FMA contraction from here on: n1 - n2 must subtract the ROUNDED unit vectors, so that identical normals give exactly 0
// (as they do in torch); contracted, the difference is the rounding error of n2 and the gradient an O(1) noise vector.
#pragma clang fp contract(off)

__device__ __forceinline__ void unit3(const float* __restrict__ g, float (&v)[3], float (&n)[3], float& r) {
  v[0] = g[0]; v[1] = g[1]; v[2] = g[2];
  r = sqrtf(v[0] * v[0] + v[1] * v[1] + v[2] * v[2]);
  const float inv = 1.0f / fmaxf(r, NRM_EPS);
  n[0] = v[0] * inv; n[1] = v[1] * inv; n[2] = v[2] * inv;
}

__global__ __launch_bounds__(256) void eik_out_fwd_kernel(const float* __restrict__ g, int64_t B, float* __restrict__ theta,
                                                           float* __restrict__ diff) {
  const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (i >= B) return;
  float v1[3], n1[3], r1, v2[3], n2[3], r2;
  unit3(g + (B + i) * 3, v1, n1, r1);
  unit3(g + (2 * B + i) * 3, v2, n2, r2);
  const float dx = n1[0] - n2[0], dy = n1[1] - n2[1], dz = n1[2] - n2[2];
  diff[i] = sqrtf(dx * dx + dy * dy + dz * dz);
  if (theta) {
#pragma unroll
    for (int c = 0; c < 3; ++c) { theta[i * 3 + c] = g[i * 3 + c]; theta[(B + i) * 3 + c] = v1[c]; }
  }
}

__device__ __forceinline__ void unit3_bwd(const float (&n)[3], float r, const float (&gn)[3], float (&gv)[3]) {
  if (r >= NRM_EPS) {       // clamp_min passes the gradient of ||v||: gv = (gn - n (gn.n)) / r
    const float dot = gn[0] * n[0] + gn[1] * n[1] + gn[2] * n[2];
    const float inv = 1.0f / r;
#pragma unroll
    for (int c = 0; c < 3; ++c) gv[c] = (gn[c] - n[c] * dot) * inv;
  } else {
#pragma unroll
    for (int c = 0; c < 3; ++c) gv[c] = gn[c] * (1.0f / NRM_EPS);
  }
}

__global__ __launch_bounds__(256) void eik_out_bwd_kernel(const float* __restrict__ g, const float* __restrict__ theta_bar,
                                                           const float* __restrict__ diff_bar, int64_t B, float* __restrict__ g_bar) {
  const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (i >= B) return;
  float o0[3] = {0.f, 0.f, 0.f}, o1[3] = {0.f, 0.f, 0.f}, o2[3] = {0.f, 0.f, 0.f};
  if (theta_bar) {
#pragma unroll
    for (int c = 0; c < 3; ++c) { o0[c] = theta_bar[i * 3 + c]; o1[c] = theta_bar[(B + i) * 3 + c]; }
  }
  if (diff_bar) {
    float v1[3], n1[3], r1, v2[3], n2[3], r2;
    unit3(g + (B + i) * 3, v1, n1, r1);
    unit3(g + (2 * B + i) * 3, v2, n2, r2);
    const float d[3] = {n1[0] - n2[0], n1[1] - n2[1], n1[2] - n2[2]};
    const float nd = sqrtf(d[0] * d[0] + d[1] * d[1] + d[2] * d[2]);
    const float f = nd > 0.f ? diff_bar[i] / nd : 0.f;
    const float gn1[3] = {f * d[0], f * d[1], f * d[2]}, gn2[3] = {-f * d[0], -f * d[1], -f * d[2]};
    float a[3], b[3];
    unit3_bwd(n1, r1, gn1, a);
    unit3_bwd(n2, r2, gn2, b);
#pragma unroll
    for (int c = 0; c < 3; ++c) { o1[c] += a[c]; o2[c] += b[c]; }
  }
#pragma unroll
  for (int c = 0; c < 3; ++c) { g_bar[i * 3 + c] = o0[c]; g_bar[(B + i) * 3 + c] = o1[c]; g_bar[(2 * B + i) * 3 + c] = o2[c]; }
}

}  // namespace

extern "C" int64_t i2sdf_loss_scratch_floats(void) { return LOSS_BLOCKS * S_N + S_N + C_N + 4; }

extern "C" int i2sdf_loss_forward_backward(const i2sdf_loss_cfg* cfg, int64_t B, int64_t n_pc, const float* rgb, const float* depth,
                                           const float* wsum, const float* normal, const float* grad_theta, const float* diff_norm,
                                           const float* surface, const float* lmask, const float* gt_rgb, const float* gt_depth,
                                           const uint8_t* depth_mask, const float* gt_normal, const uint8_t* normal_mask, const float* gt_mask,
                                           const float* gt_lmask, float* scratch, float* losses, float* loss_value, float* g_rgb, float* g_depth, float* g_wsum,
                                           float* g_normal, float* g_grad_theta, float* g_diff_norm, float* g_surface, float* g_lmask,
                                           void* stream) {
  if (!cfg || B <= 0 || !rgb || !depth || !wsum || !gt_rgb || !scratch || !losses || !g_rgb || !g_depth || !g_wsum) return I2SDF_EINVAL;
  if ((gt_depth && !depth_mask) || (gt_normal && !normal_mask) || (normal && !g_normal && gt_normal)) return I2SDF_EINVAL;
  LossArgs a{};
  a.c = *cfg; a.B = B; a.n_pc = surface ? n_pc : 0;
  a.rgb = rgb; a.depth = depth; a.wsum = wsum; a.normal = normal; a.grad_theta = grad_theta; a.diff_norm = diff_norm; a.surface = surface;
  a.lmask = lmask; a.gt_rgb = gt_rgb; a.gt_depth = gt_depth; a.gt_normal = gt_normal; a.gt_mask = gt_mask; a.gt_lmask = gt_lmask;
  a.depth_mask = depth_mask; a.normal_mask = normal_mask;
  a.partial = scratch; a.sums = scratch + LOSS_BLOCKS * S_N; a.cnt = a.sums + S_N; a.losses = losses;
  a.loss_value = loss_value;
  a.g_rgb = g_rgb; a.g_depth = g_depth; a.g_wsum = g_wsum; a.g_normal = g_normal; a.g_grad_theta = g_grad_theta; a.g_diff_norm = g_diff_norm;
  a.g_surface = g_surface; a.g_lmask = g_lmask;
  hipStream_t st = (hipStream_t)stream;
  const int64_t work = std::max<int64_t>(2 * B, a.n_pc);
  const int nb = (int)std::min<int64_t>(LOSS_BLOCKS, (work + 255) / 256);
  a.nb = nb; a.reduced = cfg->exchange ? 1 : 0;
  loss_partial_kernel<<<nb, 256, 0, st>>>(a);
  if (cfg->exchange) {       // data parallel: denominators -> their mean over the ranks (global count / world)
    if (!cfg->exchange->allreduce) return I2SDF_EINVAL;
    loss_reduce_kernel<<<1, 64, 0, st>>>(a);
    const int rc = cfg->exchange->allreduce(cfg->exchange->ctx, a.cnt, C_N, I2SDF_XCHG_F32, I2SDF_XCHG_AVG, stream);
    if (rc) return rc;
  }
  loss_grad_kernel<<<(unsigned)((work + 255) / 256), 256, 0, st>>>(a);
  return i2sdf_hip_check(hipGetLastError(), "loss_forward_backward launch");
}

// ---- glue of a training step as single launches (include/i2sdf.h: i2sdf_extra_points, i2sdf_backward_seeds) ----------------------
namespace {
__global__ __launch_bounds__(256) void extra_points_kernel(const float* __restrict__ cam, const float* __restrict__ dirs,
                                                           const float* __restrict__ z, const float* __restrict__ eik,
                                                           const float* __restrict__ off, int64_t B, float* __restrict__ out) {
#pragma clang fp contract(off)                                     // the reference's two roundings (cam + z * dirs): no fused multiply-add
  const int64_t e = (int64_t)blockIdx.x * 256 + threadIdx.x;       // element of a (B,3) block
  if (e >= 3 * B) return;
  const int64_t i = e / 3;
  const float prod = z[i] * dirs[e];
  const float near = cam[e] + prod;
  out[e] = eik[e];
  out[3 * B + e] = near;
  out[6 * B + e] = near + off[e];
}

__global__ __launch_bounds__(256) void backward_seeds_kernel(float* __restrict__ beta_grad, int64_t n_beta, float* __restrict__ sbar,
                                                             float* __restrict__ nbar, int64_t Mm, int64_t Ms,
                                                             const float* __restrict__ g_eik, int64_t n_eik,
                                                             const float* __restrict__ g_surf, int64_t n_pc, int64_t n_main) {
  int64_t e = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (e < n_beta) { beta_grad[e] = 0.f; return; }
  e -= n_beta;
  const int64_t X = Ms - Mm;                                       // extra rows
  if (e < X) {                                                     // sdf_bar rows [Mm, Ms)
    const int64_t k = e - n_eik;
    sbar[Mm + e] = (g_surf && k >= 0 && k < n_pc) ? g_surf[k] : 0.f;
    return;
  }
  e -= X;
  if (e < 3 * X) {                                                 // grad_bar rows [Mm, Ms)
    nbar[3 * Mm + e] = (g_eik && e < 3 * n_eik) ? g_eik[e] : 0.f;
    return;
  }
  e -= 3 * X;
  if (e < n_main) nbar[e] = 0.f;                                   // grad_bar rows [0, Mm) when nobody else writes them
}
}  // namespace

extern "C" int i2sdf_extra_points(const float* cam, const float* dirs, const float* z_eik, const float* eik_pts, const float* nbr_off,
                                  int64_t B, float* out, void* stream) {
  if (B == 0) return I2SDF_OK;
  if (!cam || !dirs || !z_eik || !eik_pts || !nbr_off || !out || B < 0) return I2SDF_EINVAL;
  extra_points_kernel<<<(unsigned)((3 * B + 255) / 256), 256, 0, (hipStream_t)stream>>>(cam, dirs, z_eik, eik_pts, nbr_off, B, out);
  return i2sdf_hip_check(hipGetLastError(), "extra_points launch");
}

extern "C" int i2sdf_backward_seeds(float* beta_grad, int64_t n_beta, float* sdf_bar, float* grad_bar, int64_t M_main, int64_t M_sdf,
                                    const float* g_eik, int64_t n_eik, const float* g_surf, int64_t n_pc, int32_t zero_main_grad,
                                    void* stream) {
  if (n_beta < 0 || M_main < 0 || M_sdf < M_main || n_eik < 0 || n_pc < 0 || n_eik + n_pc > M_sdf - M_main) return I2SDF_EINVAL;
  if ((n_beta > 0 && !beta_grad) || (M_sdf > 0 && (!sdf_bar || !grad_bar))) return I2SDF_EINVAL;
  const int64_t n_main = zero_main_grad ? 3 * M_main : 0;
  const int64_t total = n_beta + 4 * (M_sdf - M_main) + n_main;
  if (total == 0) return I2SDF_OK;
  backward_seeds_kernel<<<(unsigned)((total + 255) / 256), 256, 0, (hipStream_t)stream>>>(beta_grad, n_beta, sdf_bar, grad_bar, M_main, M_sdf,
                                                                                           g_eik, n_eik, g_surf, n_pc, n_main);
  return i2sdf_hip_check(hipGetLastError(), "backward_seeds launch");
}

extern "C" int i2sdf_eikonal_outputs_forward(const float* grad_all, int64_t B, float* grad_theta, float* diff_norm, void* stream) {
  if (B == 0) return I2SDF_OK;
  if (!grad_all || !diff_norm || B < 0) return I2SDF_EINVAL;
  eik_out_fwd_kernel<<<(unsigned)((B + 255) / 256), 256, 0, (hipStream_t)stream>>>(grad_all, B, grad_theta, diff_norm);
  return i2sdf_hip_check(hipGetLastError(), "eikonal_outputs_forward launch");
}

extern "C" int i2sdf_eikonal_outputs_backward(const float* grad_all, const float* grad_theta_bar, const float* diff_norm_bar, int64_t B,
                                              float* grad_all_bar, void* stream) {
  if (B == 0) return I2SDF_OK;
  if (!grad_all || !grad_all_bar || B < 0) return I2SDF_EINVAL;
  eik_out_bwd_kernel<<<(unsigned)((B + 255) / 256), 256, 0, (hipStream_t)stream>>>(grad_all, grad_theta_bar, diff_norm_bar, B, grad_all_bar);
  return i2sdf_hip_check(hipGetLastError(), "eikonal_outputs_backward launch");
}
