// Device pieces of I2SDFLoss (model/network/__init__.py:289-406) shared by the stand-alone loss launches (loss.hip) and the fused
// loss + render-backward launch (render.hip: i2sdf_render_loss_backward): per-ray loss terms, per-ray gradient seeds, the eikonal /
// smoothness outputs' backward.  One definition, so both paths compute the same numbers.
#pragma once
#include "plan.h"

namespace i2sdf {

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

// the terms of ray i, added to the running sums s
__device__ __forceinline__ void loss_ray_terms(const LossArgs& a, int64_t i, float (&s)[S_N]) {
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
__device__ __forceinline__ float loss_eik_term(const float* __restrict__ g) {      // (||g|| - 1)^2 of one row of grad_theta
  const float x = g[0], y = g[1], z = g[2];
  const float d = sqrtf(x * x + y * y + z * z) - 1.0f;
  return d * d;
}

// d loss / d (render outputs of ray i), for an upstream gradient of 1; cnt = the denominators
struct LossSeeds { float rgb[3], depth, wsum, normal[3], lmask, diff_norm; };
__device__ __forceinline__ void loss_ray_grads(const LossArgs& a, int64_t i, const float* cnt, LossSeeds& o) {
  const float B = cnt[C_B];
#pragma unroll
  for (int k = 0; k < 3; ++k) {
    const float d = a.rgb[i * 3 + k] - a.gt_rgb[i * 3 + k];
    o.rgb[k] = (d > 0.f ? 1.f : (d < 0.f ? -1.f : 0.f)) / (3.0f * B);
  }
  o.depth = 0.f;
  if (a.gt_depth && a.c.depth_w > 0.f && a.depth_mask[i]) o.depth = a.c.depth_w * 2.0f * (a.depth[i] - a.gt_depth[i]) / cnt[C_DEPTH];
  o.wsum = 0.f;
  if (a.gt_mask && a.c.mask_w > 0.f) { float d; (void)bce(a.wsum[i], a.gt_mask[i], d); o.wsum = a.c.mask_w * d / B; }
  o.normal[0] = o.normal[1] = o.normal[2] = 0.f;
  if (a.gt_normal && a.normal && a.normal_mask[i]) {
    const float w = ((a.c.normal_w > 0.f ? a.c.normal_w : 0.f) + (a.c.angular_w > 0.f ? a.c.angular_w : 0.f)) / cnt[C_NORMAL];
    const float n0 = a.gt_normal[i * 3], n1 = a.gt_normal[i * 3 + 1], n2 = a.gt_normal[i * 3 + 2];
    const float u = 1.0f - (a.normal[i * 3] * n0 + a.normal[i * 3 + 1] * n1 + a.normal[i * 3 + 2] * n2);
    const float sg = u > 0.f ? -1.f : (u < 0.f ? 1.f : 0.f);          // d|1-dot| / d dot
    o.normal[0] = w * sg * n0; o.normal[1] = w * sg * n1; o.normal[2] = w * sg * n2;
  }
  o.diff_norm = (a.c.smooth_on && a.c.smooth_w > 0.f) ? a.c.smooth_w / B : 0.f;
  o.lmask = 0.f;
  if (a.lmask && a.gt_lmask && a.c.light_w > 0.f) { float d; (void)bce(a.lmask[i], a.gt_lmask[i], d); o.lmask = a.c.light_w * d / B; }
}
// d loss / d (row of grad_theta): eikonal term, mean over 2B rows
__device__ __forceinline__ void loss_eik_grad(const LossArgs& a, const float* __restrict__ g, float B, float (&o)[3]) {
  const float x = g[0], y = g[1], z = g[2];
  const float nrm = sqrtf(x * x + y * y + z * z);
  const float f = nrm > 0.f ? a.c.eikonal_w * 2.0f * (nrm - 1.0f) / (nrm * 2.0f * B) : 0.f;
  o[0] = f * x; o[1] = f * y; o[2] = f * z;
}
__device__ __forceinline__ float loss_surface_grad(const LossArgs& a, float sv, float n_pc) {
  return a.c.bubble_w > 0.f ? a.c.bubble_w * (sv > 0.f ? 1.f : (sv < 0.f ? -1.f : 0.f)) / n_pc : 0.f;
}

// the reported values from the totals and the denominators (one thread)
__device__ __forceinline__ void loss_values(const LossArgs& a, const float* tot, const float* cnt) {
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

// ---- eikonal / smoothness outputs (model/network/__init__.py:188-193), see loss.hip -------------------------------------------------
constexpr float NRM_EPS = 1e-6f;
// No FMA contraction in these: n1 - n2 must subtract the ROUNDED unit vectors, so that identical normals give exactly 0 (as they do in
// torch); contracted, the difference is the rounding error of n2 and the gradient an O(1) noise vector.
__device__ __forceinline__ void unit3(const float* __restrict__ g, float (&v)[3], float (&n)[3], float& r) {
#pragma clang fp contract(off)
  v[0] = g[0]; v[1] = g[1]; v[2] = g[2];
  r = sqrtf(v[0] * v[0] + v[1] * v[1] + v[2] * v[2]);
  const float inv = 1.0f / fmaxf(r, NRM_EPS);
  n[0] = v[0] * inv; n[1] = v[1] * inv; n[2] = v[2] * inv;
}
__device__ __forceinline__ void unit3_bwd(const float (&n)[3], float r, const float (&gn)[3], float (&gv)[3]) {
#pragma clang fp contract(off)
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
// forward of point index i: theta rows i, B + i (the raw gradients), diff_norm[i] = || normalize(g[B+i]) - normalize(g[2B+i]) ||
__device__ __forceinline__ void eik_out_fwd_point(const float* __restrict__ g, int64_t B, int64_t i, float* __restrict__ theta, float* __restrict__ diff) {
#pragma clang fp contract(off)
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
// backward of (grad_theta rows i, B+i; diff_norm[i]) w.r.t. the gradients g of the extra points i, B+i, 2B+i: o0 / o1 / o2
// th0 / th1: d loss / d grad_theta rows i and B+i (zeros if there is none); db: d loss / d diff_norm[i] (has_diff: is there one)
__device__ __forceinline__ void eik_out_bwd_point(const float* __restrict__ g, int64_t B, int64_t i, const float (&th0)[3], const float (&th1)[3],
                                                  bool has_diff, float db, float (&o0)[3], float (&o1)[3], float (&o2)[3]) {
#pragma clang fp contract(off)
#pragma unroll
  for (int c = 0; c < 3; ++c) { o0[c] = th0[c]; o1[c] = th1[c]; o2[c] = 0.f; }
  if (has_diff) {
    float v1[3], n1[3], r1, v2[3], n2[3], r2;
    unit3(g + (B + i) * 3, v1, n1, r1);
    unit3(g + (2 * B + i) * 3, v2, n2, r2);
    const float d[3] = {n1[0] - n2[0], n1[1] - n2[1], n1[2] - n2[2]};
    const float nd = sqrtf(d[0] * d[0] + d[1] * d[1] + d[2] * d[2]);
    const float f = nd > 0.f ? db / nd : 0.f;
    const float gn1[3] = {f * d[0], f * d[1], f * d[2]}, gn2[3] = {-f * d[0], -f * d[1], -f * d[2]};
    float a_[3], b_[3];
    unit3_bwd(n1, r1, gn1, a_);
    unit3_bwd(n2, r2, gn2, b_);
#pragma unroll
    for (int c = 0; c < 3; ++c) { o1[c] += a_[c]; o2[c] += b_[c]; }
  }
}

}  // namespace i2sdf
