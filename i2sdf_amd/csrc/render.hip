// Per-ray kernels: ray generation, Laplace density + log-space alpha compositing (forward and backward).
// One wave (64 lanes) per ray, the ray's samples live in registers, prefix sums by cross-lane shuffles.
// HBM-bound (a few KB per ray); see DESIGN.md for the byte counts.
#include <algorithm>
#include "loss_dev.h"

using namespace i2sdf;

int i2sdf_hip_check(hipError_t e, const char* what);

namespace {

constexpr int MAX_SEG = 4;      // up to 256 samples per ray

// wave-wide inclusive scan / sum by DPP (row_shr:1/2/4/8 inside the 16-lane rows, row_bcast:15 / :31 across them) instead of six dependent
// ds_bpermute_b32 round trips through the LDS crossbar each (round 6; sampler.hip has the same pair with its measurements)
template <int CTRL, int ROW_MASK>
__device__ __forceinline__ float dpp_mov0(float v) {      // lanes without a source lane (or outside ROW_MASK) get 0
  return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), CTRL, ROW_MASK, 0xf, false));
}
__device__ __forceinline__ float wave_incl_scan(float v, int lane) {
  (void)lane;
  v += dpp_mov0<0x111, 0xf>(v);
  v += dpp_mov0<0x112, 0xf>(v);
  v += dpp_mov0<0x114, 0xf>(v);
  v += dpp_mov0<0x118, 0xf>(v);
  v += dpp_mov0<0x142, 0xa>(v);      // row_bcast:15 -> rows 1 and 3
  v += dpp_mov0<0x143, 0xc>(v);      // row_bcast:31 -> rows 2 and 3
  return v;
}
__device__ __forceinline__ float wave_sum(float v) {
  return __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, wave_incl_scan(v, 0)), 63));
}

// ---------------------------------------------------------------------------------------------
// R1/R2: utils/rend_util.py:92-147 (get_camera_params + lift, pose-matrix branch) and
// model/network/__init__.py:88-93 (repeat cam_loc, norm, F.normalize eps 1e-12).
// ---------------------------------------------------------------------------------------------
// un-normalised world direction of pixel (x, y) -> unit dir, norm.  p = pose rows (3x4), k = intrinsics (4x4 row-major)
__device__ __forceinline__ void pixel_ray(float x, float y, const float* k, const float (&p)[12], float (&d)[3], float& nrm) {
  const float fx = k[0], sk = k[1], cx = k[2], fy = k[5], cy = k[6];
  // x_lift = (x - cx + cy*sk/fy - sk*y/fy) / fx * z ; z = 1   (rend_util.py:143-144), same association order
  const float xl = __fdiv_rn(__fsub_rn(__fadd_rn(__fsub_rn(x, cx), __fdiv_rn(__fmul_rn(cy, sk), fy)), __fdiv_rn(__fmul_rn(sk, y), fy)), fx);
  const float yl = __fdiv_rn(__fsub_rn(y, cy), fy);
#pragma unroll
  for (int r = 0; r < 3; ++r) {
    // world = pose @ [xl, yl, 1, 1]; dir = world - pose[:3,3]
    const float w = fmaf(p[r * 4 + 0], xl, fmaf(p[r * 4 + 1], yl, p[r * 4 + 2] + p[r * 4 + 3]));
    d[r] = w - p[r * 4 + 3];
  }
  nrm = sqrtf(d[0] * d[0] + d[1] * d[1] + d[2] * d[2]);
  const float inv = 1.0f / fmaxf(nrm, 1e-12f);
  d[0] *= inv; d[1] *= inv; d[2] *= inv;
}

// pose rows from either a 4x4 cam->world matrix or [qr qi qj qk tx ty tz] (rend_util.py:93-98, quat_to_rot :150-167)
__device__ __forceinline__ void load_pose(const float* pose, int64_t b, bool quat, float (&p)[12]) {
  if (!quat) {
#pragma unroll
    for (int i = 0; i < 12; ++i) p[i] = pose[b * 16 + i];
    return;
  }
  const float* q = pose + b * 7;
  const float n = fmaxf(sqrtf(q[0] * q[0] + q[1] * q[1] + q[2] * q[2] + q[3] * q[3]), 1e-12f);
  const float qr = q[0] / n, qi = q[1] / n, qj = q[2] / n, qk = q[3] / n;
  p[0] = 1.f - 2.f * (qj * qj + qk * qk); p[1] = 2.f * (qj * qi - qk * qr);       p[2] = 2.f * (qi * qk + qr * qj);        p[3] = q[4];
  p[4] = 2.f * (qj * qi + qk * qr);       p[5] = 1.f - 2.f * (qi * qi + qk * qk); p[6] = 2.f * (qj * qk - qi * qr);        p[7] = q[5];
  p[8] = 2.f * (qk * qi - qj * qr);       p[9] = 2.f * (qj * qk + qi * qr);       p[10] = 1.f - 2.f * (qi * qi + qj * qj); p[11] = q[6];
}

__global__ __launch_bounds__(256) void raygen_kernel(const float* __restrict__ uv, const float* __restrict__ pose, bool quat,
                                                      const float* __restrict__ K, int64_t N, int P, float* __restrict__ cam,
                                                      float* __restrict__ dirs, float* __restrict__ dnorm) {
  const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (i >= N) return;
  const int64_t b = i / P;
  float p[12], d[3], nrm;
  load_pose(pose, b, quat, p);
  pixel_ray(uv[i * 2 + 0], uv[i * 2 + 1], K + b * 16, p, d, nrm);
  dnorm[i] = nrm;
#pragma unroll
  for (int r = 0; r < 3; ++r) { cam[i * 3 + r] = p[r * 4 + 3]; dirs[i * 3 + r] = d[r]; }
}

// N2: ray batch straight from HBM-resident camera / image tables.  One thread per ray: the global pixel index selects
// the image (K, pose: 128 B shared by all rays of an image, cache resident) and the pixel; ground truth is gathered in the same pass.
__global__ __launch_bounds__(256) void ray_batch_kernel(i2sdf_ray_tables t, const int64_t* __restrict__ tidx, int64_t B,
                                                         i2sdf_ray_batch_out o) {
  const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (i >= B) return;
  const int64_t hw = (int64_t)t.height * t.width;
  int64_t g = tidx[i];
  if (g < 0 || g >= hw * t.n_images) {            // never read outside the tables: clamp, and count (i2sdf_ray_batch_out.n_bad)
    if (o.n_bad) atomicAdd(o.n_bad, 1);
    g = g < 0 ? 0 : hw * t.n_images - 1;
  }
  const int64_t img = g / hw, pix = g - img * hw;
  const float x = (float)(pix % t.width), y = (float)(pix / t.width);     // dataset/train_dataset.py:67-70 (flipped mgrid: uv = (col, row))
  float p[12], d[3], nrm;
  load_pose(t.pose, img, t.pose_is_quat != 0, p);
  pixel_ray(x, y, t.intrinsics + img * 16, p, d, nrm);
  if (o.image_idx) o.image_idx[i] = img;
  if (o.uv) { o.uv[i * 2 + 0] = x; o.uv[i * 2 + 1] = y; }
  o.dnorm[i] = nrm;
#pragma unroll
  for (int r = 0; r < 3; ++r) { o.cam_loc[i * 3 + r] = p[r * 4 + 3]; o.dirs[i * 3 + r] = d[r]; }
  if (t.rgb && o.rgb) {
#pragma unroll
    for (int r = 0; r < 3; ++r) o.rgb[i * 3 + r] = t.rgb[g * 3 + r];
  }
  if (t.depth && o.depth) o.depth[i] = t.depth[g];
  if (t.normal && o.normal) {
#pragma unroll
    for (int r = 0; r < 3; ++r) o.normal[i * 3 + r] = t.normal[g * 3 + r];
  }
  if (t.mask && o.mask) o.mask[i] = t.mask[g];
  if (t.light_mask && o.light_mask) o.light_mask[i] = t.light_mask[g];
  if (t.depth_mask && o.depth_mask) o.depth_mask[i] = t.depth_mask[g];
  if (t.normal_mask && o.normal_mask) o.normal_mask[i] = t.normal_mask[g];
}

// R3: utils/rend_util.py:211-227.  Rays that miss the sphere are counted instead of exit()
__global__ __launch_bounds__(256) void sphere_isect_kernel(const float* __restrict__ cam, const float* __restrict__ dirs, int64_t N,
                                                            float r, float* __restrict__ out, int32_t* __restrict__ n_miss) {
  const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (i >= N) return;
  const float ox = cam[i * 3], oy = cam[i * 3 + 1], oz = cam[i * 3 + 2];
  const float dot = dirs[i * 3] * ox + dirs[i * 3 + 1] * oy + dirs[i * 3 + 2] * oz;
  const float nn = sqrtf(ox * ox + oy * oy + oz * oz);
  const float under = dot * dot - (nn * nn - r * r);
  if (under <= 0.f) {
    atomicAdd(n_miss, 1);
    out[i * 2] = 0.f; out[i * 2 + 1] = 0.f;
    return;
  }
  const float sq = sqrtf(under);
  out[i * 2 + 0] = fmaxf(-sq - dot, 0.f);
  out[i * 2 + 1] = fmaxf(sq - dot, 0.f);
}

// Laplace density (model/network/density.py:21-26): (1/beta)(0.5 + 0.5 sign(s) expm1(-|s|/beta))
__device__ __forceinline__ float laplace_density(float s, float inv_beta) {
  const float e = expm1f(-fabsf(s) * inv_beta);
  const float sg = (s > 0.f) ? 1.f : ((s < 0.f) ? -1.f : 0.f);
  return inv_beta * (0.5f + 0.5f * sg * e);
}

struct CompArgs {
  const float* beta_param; float beta_min;
  const float* z; int64_t ldz;            // (B, n+1): samples then z_max
  const float* sdf;                       // (B*n)
  const float* rgb;                       // (B*n,3)
  const float* grad;                      // (B*n,3) or null
  const float* lmask;                     // (B*n) or null
  const float* dnorm;                     // (B)
  int64_t B; int n;
  int normal_detached;                    // informational (forward is identical)
  float* o_rgb; float* o_depth; float* o_wsum; float* o_normal; float* o_lmask;
  float* w_save;                          // (B,n) or null
  float* nsum_save;                       // (B,3) or null
  // backward
  const float* g_rgb; const float* g_depth; const float* g_wsum; const float* g_normal; const float* g_lmask;
  float* sdf_bar; float* rgb_bar; float* grad_bar; float* lmask_bar; float* beta_bar_partial;   // (B)
  // composite forward + the eikonal / smoothness outputs of the extra points in one launch (i2sdf_composite_forward_eik): eik_g (3B,3) =
  // d sdf / d x of the extra points -> eik_theta (2B,3), eik_diff (B); the wave of ray i does point index i
  const float* eik_g = nullptr; float* eik_theta = nullptr; float* eik_diff = nullptr;
  float* grad_bar_all = nullptr;          // fused loss + backward only: the whole (M_sdf,3) d loss / d grad tensor (grad_bar = its ray-sample rows when there is a normal term, else NULL)
};

// model/network/__init__.py:223-240 (volume_rendering) + :120-125,:169,:204-219 (composites)
__global__ __launch_bounds__(256) void composite_fwd_kernel(CompArgs a) {
  const int lane = threadIdx.x & 63;
  const int64_t ray = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (ray >= a.B) return;
  const float beta = fabsf(a.beta_param[0]) + a.beta_min;
  const float inv_beta = 1.0f / beta;
  const float* zr = a.z + ray * a.ldz;
  const int n = a.n;
  float carry = 0.f;
  float acc_rgb[3] = {0, 0, 0}, acc_d = 0.f, acc_w = 0.f, acc_n[3] = {0, 0, 0}, acc_l = 0.f;
  for (int base = 0; base < n; base += 64) {
    const int j = base + lane;
    const bool ok = j < n;
    const int64_t m = ray * n + (ok ? j : n - 1);
    float zj = 0.f, E = 0.f;
    if (ok) {
      zj = zr[j];
      const float delta = zr[j + 1] - zj;      // last: z_max - z_last (z_max is stored as column n)
      E = delta * laplace_density(a.sdf[m], inv_beta);
    }
    const float incl = wave_incl_scan(E, lane);
    const float excl = carry + incl - E;
    const float T = expf(-excl);
    const float w = ok ? (1.0f - expf(-E)) * T : 0.f;
    carry += __shfl(incl, 63);
    if (ok && a.w_save) a.w_save[ray * n + j] = w;
    if (ok) {
      acc_rgb[0] = fmaf(w, a.rgb[m * 3 + 0], acc_rgb[0]);
      acc_rgb[1] = fmaf(w, a.rgb[m * 3 + 1], acc_rgb[1]);
      acc_rgb[2] = fmaf(w, a.rgb[m * 3 + 2], acc_rgb[2]);
      acc_d = fmaf(w, zj, acc_d);
      acc_w += w;
      if (a.grad) {
        const float gx = a.grad[m * 3 + 0], gy = a.grad[m * 3 + 1], gz = a.grad[m * 3 + 2];
        const float inv = 1.0f / fmaxf(sqrtf(gx * gx + gy * gy + gz * gz), 1e-12f);
        acc_n[0] = fmaf(w, gx * inv, acc_n[0]); acc_n[1] = fmaf(w, gy * inv, acc_n[1]); acc_n[2] = fmaf(w, gz * inv, acc_n[2]);
      }
      if (a.lmask) acc_l = fmaf(w, a.lmask[m], acc_l);
    }
  }
#pragma unroll
  for (int i = 0; i < 3; ++i) { acc_rgb[i] = wave_sum(acc_rgb[i]); acc_n[i] = wave_sum(acc_n[i]); }
  acc_d = wave_sum(acc_d); acc_w = wave_sum(acc_w); acc_l = wave_sum(acc_l);
  if (lane == 0) {
    a.o_rgb[ray * 3 + 0] = acc_rgb[0]; a.o_rgb[ray * 3 + 1] = acc_rgb[1]; a.o_rgb[ray * 3 + 2] = acc_rgb[2];
    a.o_depth[ray] = acc_d / fmaxf(a.dnorm[ray], 1e-6f);
    a.o_wsum[ray] = acc_w;
    if (a.o_normal) {
      const float inv = 1.0f / fmaxf(sqrtf(acc_n[0] * acc_n[0] + acc_n[1] * acc_n[1] + acc_n[2] * acc_n[2]), 1e-12f);
      a.o_normal[ray * 3 + 0] = acc_n[0] * inv; a.o_normal[ray * 3 + 1] = acc_n[1] * inv; a.o_normal[ray * 3 + 2] = acc_n[2] * inv;
      if (a.nsum_save) { a.nsum_save[ray * 3 + 0] = acc_n[0]; a.nsum_save[ray * 3 + 1] = acc_n[1]; a.nsum_save[ray * 3 + 2] = acc_n[2]; }
    }
    if (a.o_lmask) a.o_lmask[ray] = acc_l;
    if (a.eik_g) eik_out_fwd_point(a.eik_g, a.B, ray, a.eik_theta, a.eik_diff);
  }
}

// SURVEY appendix A.5.  The normal and light-mask composites use w.detach() in training
// (model/network/__init__.py:169,207), so they contribute to grad_bar / lmask_bar only.
// One wave = one ray.  gr / gd_raw / gw / gn / gl: d loss / d (rgb, depth, weight_sum, normal_values, light_mask) of this ray; returns the ray's
// share of d loss / d beta (summed over the wave).  Shared by composite_bwd_kernel and the fused loss + render-backward kernel below.
__device__ __forceinline__ float composite_bwd_ray(const CompArgs& a, int64_t ray, int lane, const float (&gr)[3], float gd_raw, float gw,
                                                   const float (&gn)[3], bool has_gn, float gl) {
  const float beta = fabsf(a.beta_param[0]) + a.beta_min;
  const float inv_beta = 1.0f / beta;
  const float* zr = a.z + ray * a.ldz;
  const int n = a.n;
  const int nseg = (n + 63) / 64;
  const float gr0 = gr[0], gr1 = gr[1], gr2 = gr[2];
  const float gd = gd_raw / fmaxf(a.dnorm[ray], 1e-6f);
  // d normal_values / d nsum
  float gN[3] = {0, 0, 0};
  if (has_gn && a.grad_bar) {
    const float N0 = a.nsum_save[ray * 3 + 0], N1 = a.nsum_save[ray * 3 + 1], N2 = a.nsum_save[ray * 3 + 2];
    const float nn = sqrtf(N0 * N0 + N1 * N1 + N2 * N2);
    if (nn > 1e-12f) {
      const float inv = 1.0f / nn;
      const float o0 = N0 * inv, o1 = N1 * inv, o2 = N2 * inv;
      const float g0 = gn[0], g1 = gn[1], g2 = gn[2];
      const float dot = o0 * g0 + o1 * g1 + o2 * g2;
      gN[0] = (g0 - o0 * dot) * inv; gN[1] = (g1 - o1 * dot) * inv; gN[2] = (g2 - o2 * dot) * inv;
    } else {
      const float inv = 1e12f;
      gN[0] = gn[0] * inv; gN[1] = gn[1] * inv; gN[2] = gn[2] * inv;
    }
  }
  // pass 1 (forward order): E, T, w, wbar per lane-segment, kept in registers
  float E[MAX_SEG], T[MAX_SEG], w[MAX_SEG], wb[MAX_SEG], sd[MAX_SEG], dl[MAX_SEG];
  float carry = 0.f;
#pragma unroll
  for (int sgi = 0; sgi < MAX_SEG; ++sgi) {
    E[sgi] = T[sgi] = w[sgi] = wb[sgi] = sd[sgi] = dl[sgi] = 0.f;
    if (sgi < nseg) {
      const int j = sgi * 64 + lane;
      const bool ok = j < n;
      const int64_t m = ray * n + (ok ? j : n - 1);
      float zj = 0.f;
      if (ok) {
        zj = zr[j];
        dl[sgi] = zr[j + 1] - zj;
        sd[sgi] = a.sdf[m];
        E[sgi] = dl[sgi] * laplace_density(sd[sgi], inv_beta);
      }
      const float incl = wave_incl_scan(E[sgi], lane);
      T[sgi] = expf(-(carry + incl - E[sgi]));
      w[sgi] = ok ? (1.0f - expf(-E[sgi])) * T[sgi] : 0.f;
      carry += __shfl(incl, 63);
      if (ok) {
        const float c0 = a.rgb[m * 3 + 0], c1 = a.rgb[m * 3 + 1], c2 = a.rgb[m * 3 + 2];
        wb[sgi] = gr0 * c0 + gr1 * c1 + gr2 * c2 + gd * zj + gw;
        a.rgb_bar[m * 3 + 0] = w[sgi] * gr0; a.rgb_bar[m * 3 + 1] = w[sgi] * gr1; a.rgb_bar[m * 3 + 2] = w[sgi] * gr2;
        if (a.grad_bar) {
          // d/d g of  w * g/||g||  contracted with gN
          const float gx = a.grad[m * 3 + 0], gy = a.grad[m * 3 + 1], gz = a.grad[m * 3 + 2];
          const float nn = fmaxf(sqrtf(gx * gx + gy * gy + gz * gz), 1e-12f);
          const float inv = 1.0f / nn;
          const float h0 = gx * inv, h1 = gy * inv, h2 = gz * inv;
          const float u0 = w[sgi] * gN[0], u1 = w[sgi] * gN[1], u2 = w[sgi] * gN[2];
          const float dot = h0 * u0 + h1 * u1 + h2 * u2;
          a.grad_bar[m * 3 + 0] = (u0 - h0 * dot) * inv; a.grad_bar[m * 3 + 1] = (u1 - h1 * dot) * inv; a.grad_bar[m * 3 + 2] = (u2 - h2 * dot) * inv;
        }
        if (a.lmask_bar) a.lmask_bar[m] = w[sgi] * gl;
      }
    }
  }
  // pass 2 (reverse order): suffix sums of wbar*w
  float tail = 0.f, bb = 0.f;
#pragma unroll
  for (int sgi = MAX_SEG - 1; sgi >= 0; --sgi) {
    if (sgi < nseg) {
      const int j = sgi * 64 + lane;
      const bool ok = j < n;
      const float ww = wb[sgi] * w[sgi];
      const float incl = wave_incl_scan(ww, lane);
      const float tot = __shfl(incl, 63);
      const float suffix = tail + (tot - incl);          // sum over i > j
      tail += tot;
      if (ok) {
        const float Ebar = wb[sgi] * (T[sgi] - w[sgi]) - suffix;
        const float sbar_sigma = dl[sgi] * Ebar;
        const float s = sd[sgi];
        const float ex = expf(-fabsf(s) * inv_beta);
        a.sdf_bar[ray * n + j] = sbar_sigma * (-ex * 0.5f * inv_beta * inv_beta);
        const float sigma = laplace_density(s, inv_beta);
        bb += sbar_sigma * (-sigma * inv_beta + s * ex * 0.5f * inv_beta * inv_beta * inv_beta);
      }
    }
  }
  return wave_sum(bb);
}

__global__ __launch_bounds__(256) void composite_bwd_kernel(CompArgs a) {
  const int lane = threadIdx.x & 63;
  const int64_t ray = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (ray >= a.B) return;
  const float gr[3] = {a.g_rgb[ray * 3 + 0], a.g_rgb[ray * 3 + 1], a.g_rgb[ray * 3 + 2]};
  const bool has_gn = a.g_normal != nullptr;
  const float gn[3] = {has_gn ? a.g_normal[ray * 3 + 0] : 0.f, has_gn ? a.g_normal[ray * 3 + 1] : 0.f, has_gn ? a.g_normal[ray * 3 + 2] : 0.f};
  const float gl = (a.g_lmask && a.lmask_bar) ? a.g_lmask[ray] : 0.f;
  const float bb = composite_bwd_ray(a, ray, lane, gr, a.g_depth ? a.g_depth[ray] : 0.f, a.g_wsum ? a.g_wsum[ray] : 0.f, gn, has_gn, gl);
  if (lane == 0) a.beta_bar_partial[ray] = bb;
}

// ---------------------------------------------------------------------------------------------------------------
// Round 6: I2SDFLoss + everything between it and the radiance backward in ONE launch (+ a one-workgroup finish).  Through round 5 the
// step ran composite_fwd -> eik_out_fwd -> loss_partial -> loss_grad -> [torch: ones seed, foreach-mul] -> eik_out_bwd -> backward_seeds
// -> composite_bwd -> beta_reduce between the radiance forward and backward: ten launches of a few microseconds each (~67 us with their
// gaps).  A workgroup = 4 rays, one wave each, as in the compositing kernels.  It
//   1. counts the masked rays itself (the denominators of the masked means are sums of 0 / 1 over the ground truth's masks: B bytes per
//      mask, exact in any order) -- so no reduction stands between the loss terms and their gradients;
//   2. per ray: the gradient seeds of the render outputs (loss_dev.h: loss_ray_grads, for an upstream gradient of 1), written out for
//      autograd, and straight into composite_bwd_ray -> sdf_bar / rgb_bar / grad_bar / lmask_bar rows of the ray's samples;
//   3. per ray index i: the eikonal / smoothness gradients of the extra points i, B + i, 2B + i (eik_out_bwd_point) -> grad_bar rows,
//      zero sdf_bar rows; the bubble point cloud's rows by a grid-stride loop;
//   4. its partial sums of the ten loss terms and of d loss / d beta -> partial[workgroup]; render_loss_finish_kernel adds them up in
//      workgroup order (deterministic) -> the reported values, the total, d loss / d beta_param.
// The module scales sdf_bar / rgb_bar / grad_bar / lmask_bar / the beta gradient by the upstream gradient when autograd delivers it
// (i2sdf_scale_seeds: one launch).
// ---------------------------------------------------------------------------------------------------------------
struct RenderLossArgs {
  LossArgs l;            // render outputs, ground truth, loss configuration, output gradient seeds (g_*), losses / loss_value
  CompArgs c;            // compositing inputs and the per-sample gradient outputs
  const float* grad_all; // (3B,3) d sdf / d x of the extra points (rows M_main .. of the forward's gradient tensor) or NULL
  int64_t M_main, M_sdf; // rows of sdf_bar / grad_bar: [ray samples | 3B extra points | n_pc bubble points | ...]
  int64_t n_eik;         // 3B or 0
  float* part;           // (workgroups, S_N + 1): loss sums, beta partial
  float* beta_grad;      // (1)
};

__global__ __launch_bounds__(256) void render_loss_bwd_kernel(RenderLossArgs a) {
  __shared__ float sm[4][S_N + 1];
  __shared__ float s_cnt[C_N];
  __shared__ float s_red[2][4];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int64_t B = a.l.B;
  // 1. denominators
  {
    float cd = 0.f, cn = 0.f;
    for (int64_t i = tid; i < B; i += 256) {
      if (a.l.gt_depth) cd += a.l.depth_mask[i] ? 1.f : 0.f;
      if (a.l.gt_normal && a.l.normal) cn += a.l.normal_mask[i] ? 1.f : 0.f;
    }
    cd = wave_sum(cd); cn = wave_sum(cn);
    if (lane == 0) { s_red[0][wave] = cd; s_red[1][wave] = cn; }
    __syncthreads();
    if (tid == 0) {
      s_cnt[C_B] = (float)B; s_cnt[C_NPC] = (float)a.l.n_pc;
      s_cnt[C_DEPTH] = (s_red[0][0] + s_red[0][1]) + (s_red[0][2] + s_red[0][3]);
      s_cnt[C_NORMAL] = (s_red[1][0] + s_red[1][1]) + (s_red[1][2] + s_red[1][3]);
    }
    __syncthreads();
  }
  float s[S_N];
#pragma unroll
  for (int k = 0; k < S_N; ++k) s[k] = 0.f;
  float bb = 0.f;
  const int64_t ray = (int64_t)blockIdx.x * 4 + wave;
  if (ray < B) {
    LossSeeds g;
    loss_ray_grads(a.l, ray, s_cnt, g);
    if (lane == 0) {
      loss_ray_terms(a.l, ray, s);
#pragma unroll
      for (int k = 0; k < 3; ++k) a.l.g_rgb[ray * 3 + k] = g.rgb[k];
      a.l.g_depth[ray] = g.depth;
      a.l.g_wsum[ray] = g.wsum;
      if (a.l.g_normal) { a.l.g_normal[ray * 3] = g.normal[0]; a.l.g_normal[ray * 3 + 1] = g.normal[1]; a.l.g_normal[ray * 3 + 2] = g.normal[2]; }
      if (a.l.g_diff_norm) a.l.g_diff_norm[ray] = g.diff_norm;
      if (a.l.g_lmask) a.l.g_lmask[ray] = g.lmask;
    }
    // 2. compositing backward of this ray with those seeds
    const bool has_gn = a.l.g_normal != nullptr && a.c.grad_bar != nullptr && a.c.nsum_save != nullptr;
    bb = composite_bwd_ray(a.c, ray, lane, g.rgb, g.depth, g.wsum, g.normal, has_gn, a.c.lmask_bar ? g.lmask : 0.f);
    // 3. the extra points of ray index i = ray: rows M_main + {i, B + i, 2B + i}
    if (a.n_eik > 0 && lane == 0) {
      float th0[3] = {0.f, 0.f, 0.f}, th1[3] = {0.f, 0.f, 0.f}, o0[3], o1[3], o2[3];
      const float* gth = a.l.grad_theta;          // (2B,3) = rows [0, 2B) of grad_all (theta row B + i = the near-surface point's gradient)
      if (gth) {
        s[S_EIK] += loss_eik_term(gth + ray * 3) + loss_eik_term(gth + (B + ray) * 3);
        loss_eik_grad(a.l, gth + ray * 3, (float)B, th0);
        loss_eik_grad(a.l, gth + (B + ray) * 3, (float)B, th1);
        if (a.l.g_grad_theta) {
#pragma unroll
          for (int c = 0; c < 3; ++c) { a.l.g_grad_theta[ray * 3 + c] = th0[c]; a.l.g_grad_theta[(B + ray) * 3 + c] = th1[c]; }
        }
      }
      const bool has_diff = a.l.diff_norm != nullptr && g.diff_norm != 0.f;
      eik_out_bwd_point(a.grad_all, B, ray, th0, th1, has_diff, g.diff_norm, o0, o1, o2);
      float* nb = a.c.grad_bar_all + a.M_main * 3;
#pragma unroll
      for (int c = 0; c < 3; ++c) { nb[ray * 3 + c] = o0[c]; nb[(B + ray) * 3 + c] = o1[c]; nb[(2 * B + ray) * 3 + c] = o2[c]; }
      float* sb = a.c.sdf_bar + a.M_main;
      sb[ray] = 0.f; sb[B + ray] = 0.f; sb[2 * B + ray] = 0.f;
    }
    // rays without a normal term: the ray samples' grad_bar rows are zeros (the sweeps read them)
    if (a.c.grad_bar == nullptr && a.c.grad_bar_all != nullptr) {
      for (int j = lane; j < 3 * a.c.n; j += 64) a.c.grad_bar_all[ray * 3 * a.c.n + j] = 0.f;
    }
  }
  // bubble point cloud + any rows behind it: sdf_bar = d loss / d sdf, grad_bar = 0
  {
    const int64_t X = a.M_sdf - a.M_main - a.n_eik;
    const int64_t stride = (int64_t)gridDim.x * 256;
    for (int64_t k = (int64_t)blockIdx.x * 256 + tid; k < X; k += stride) {
      float gs = 0.f;
      if (a.l.surface && k < a.l.n_pc) {
        const float sv = a.l.surface[k];
        s[S_BUBBLE] += fabsf(sv);
        gs = loss_surface_grad(a.l, sv, s_cnt[C_NPC]);
        if (a.l.g_surface) a.l.g_surface[k] = gs;
      }
      const int64_t row = a.M_main + a.n_eik + k;
      a.c.sdf_bar[row] = gs;
      a.c.grad_bar_all[row * 3] = 0.f; a.c.grad_bar_all[row * 3 + 1] = 0.f; a.c.grad_bar_all[row * 3 + 2] = 0.f;
    }
  }
  // 4. partial sums of this workgroup
#pragma unroll
  for (int k = 0; k < S_N; ++k) {
    const float v = wave_sum(s[k]);
    if (lane == 0) sm[wave][k] = v;
  }
  if (lane == 0) sm[wave][S_N] = bb;
  __syncthreads();
  if (tid <= S_N) a.part[(int64_t)blockIdx.x * (S_N + 1) + tid] = (sm[0][tid] + sm[1][tid]) + (sm[2][tid] + sm[3][tid]);
}

// one workgroup: the workgroups' partial sums in workgroup order -> reported values, total, d loss / d beta_param (for an upstream gradient of 1)
__global__ __launch_bounds__(256) void render_loss_finish_kernel(RenderLossArgs a, int nwg) {
  __shared__ float tot[S_N + 1], cnt[C_N];
  const int tid = threadIdx.x;
  // column c by the 16 threads 16 c .. 16 c + 15: thread j sums workgroups j, j + 16, ... in order, then a fixed tree over the 16
  const int c = tid >> 4, j = tid & 15;
  float v = 0.f;
  if (c <= S_N)
    for (int b = j; b < nwg; b += 16) v += a.part[(int64_t)b * (S_N + 1) + c];
#pragma unroll
  for (int o = 8; o > 0; o >>= 1) v += __shfl_xor(v, o);
  if (c <= S_N && j == 0) tot[c] = v;
  __syncthreads();
  if (tid == 0) {
    cnt[C_B] = (float)a.l.B; cnt[C_NPC] = (float)a.l.n_pc; cnt[C_DEPTH] = tot[S_DEPTH_CNT]; cnt[C_NORMAL] = tot[S_NORMAL_CNT];
    loss_values(a.l, tot, cnt);
    const float b = a.c.beta_param[0];
    a.beta_grad[0] = (b > 0.f ? 1.f : (b < 0.f ? -1.f : 0.f)) * tot[S_N];
  }
}

// seeds computed for an upstream gradient of 1 -> times the gradient autograd delivered (a device scalar); beta_out[0] = beta_in[0] * g
__global__ __launch_bounds__(256) void scale_seeds_kernel(const float* __restrict__ g, float* __restrict__ x0, int64_t n0, float* __restrict__ x1, int64_t n1,
                                                          float* __restrict__ x2, int64_t n2, float* __restrict__ x3, int64_t n3,
                                                          const float* __restrict__ beta_in, float* __restrict__ beta_out) {
  const float s = g[0];
  const int64_t stride = (int64_t)gridDim.x * 256;
  const int64_t t = (int64_t)blockIdx.x * 256 + threadIdx.x;
  for (int64_t i = t; i < n0; i += stride) x0[i] *= s;
  for (int64_t i = t; i < n1; i += stride) x1[i] *= s;
  for (int64_t i = t; i < n2; i += stride) x2[i] *= s;
  for (int64_t i = t; i < n3; i += stride) x3[i] *= s;
  if (t == 0 && beta_out) beta_out[0] = beta_in[0] * s;
}

// deterministic sum of per-ray partials; writes sign(beta_param) * sum  (d|b|/db) ACCUMULATING into out[0]
__global__ __launch_bounds__(1024) void beta_reduce_kernel(const float* __restrict__ part, int64_t B, const float* __restrict__ beta_param,
                                                            float* __restrict__ out) {
  __shared__ float sm[16];
  float s = 0.f;
  for (int64_t i = threadIdx.x; i < B; i += 1024) s += part[i];
  s = wave_sum(s);
  if ((threadIdx.x & 63) == 0) sm[threadIdx.x >> 6] = s;
  __syncthreads();
  if (threadIdx.x == 0) {
    float t = 0.f;
    for (int i = 0; i < 16; ++i) t += sm[i];
    const float b = beta_param[0];
    out[0] += (b > 0.f ? 1.f : (b < 0.f ? -1.f : 0.f)) * t;
  }
}

}  // namespace

extern "C" int i2sdf_ray_setup(const float* uv, const float* pose, const float* intrinsics, int64_t batch, int32_t pixels, float* cam_loc,
                               float* dirs, float* dnorm, void* stream) {
  return i2sdf_ray_setup_ex(uv, pose, 0, intrinsics, batch, pixels, cam_loc, dirs, dnorm, stream);
}

extern "C" int i2sdf_ray_setup_ex(const float* uv, const float* pose, int32_t pose_is_quat, const float* intrinsics, int64_t batch,
                                  int32_t pixels, float* cam_loc, float* dirs, float* dnorm, void* stream) {
  if (batch < 0 || pixels < 0) return I2SDF_EINVAL;
  const int64_t N = batch * pixels;
  if (N == 0) return I2SDF_OK;
  if (!uv || !pose || !intrinsics || !cam_loc || !dirs || !dnorm) return I2SDF_EINVAL;
  raygen_kernel<<<(unsigned)((N + 255) / 256), 256, 0, (hipStream_t)stream>>>(uv, pose, pose_is_quat != 0, intrinsics, N, pixels, cam_loc,
                                                                              dirs, dnorm);
  return i2sdf_hip_check(hipGetLastError(), "ray_setup launch");
}

extern "C" int i2sdf_ray_batch(const i2sdf_ray_tables* t, const int64_t* tidx, int64_t n_rays, const i2sdf_ray_batch_out* out, void* stream) {
  if (!t || !out || n_rays < 0) return I2SDF_EINVAL;
  if (n_rays == 0) return I2SDF_OK;
  if (!tidx || !t->intrinsics || !t->pose || t->n_images <= 0 || t->height <= 0 || t->width <= 0) return I2SDF_EINVAL;
  if (!out->cam_loc || !out->dirs || !out->dnorm) return I2SDF_EINVAL;
  ray_batch_kernel<<<(unsigned)((n_rays + 255) / 256), 256, 0, (hipStream_t)stream>>>(*t, tidx, n_rays, *out);
  return i2sdf_hip_check(hipGetLastError(), "ray_batch launch");
}

extern "C" int i2sdf_sphere_intersections(const float* cam_loc, const float* dirs, int64_t n_rays, float radius, float* t_near_far,
                                          int32_t* n_miss, void* stream) {
  if (n_rays < 0) return I2SDF_EINVAL;
  if (n_rays == 0) return I2SDF_OK;
  if (!cam_loc || !dirs || !t_near_far || !n_miss) return I2SDF_EINVAL;
  sphere_isect_kernel<<<(unsigned)((n_rays + 255) / 256), 256, 0, (hipStream_t)stream>>>(cam_loc, dirs, n_rays, radius, t_near_far, n_miss);
  return i2sdf_hip_check(hipGetLastError(), "sphere_intersections launch");
}

extern "C" int i2sdf_composite_forward(const float* beta_param, float beta_min, const float* z, int64_t ldz, const float* sdf,
                                       const float* rgb, const float* grad, const float* lmask, const float* dnorm, int64_t B, int32_t n,
                                       float* o_rgb, float* o_depth, float* o_wsum, float* o_normal, float* o_lmask, float* w_save,
                                       float* nsum_save, void* stream) {
  if (B == 0) return I2SDF_OK;
  if (!beta_param || !z || !sdf || !rgb || !dnorm || !o_rgb || !o_depth || !o_wsum || B < 0 || n <= 0 || n > 64 * MAX_SEG) return I2SDF_EINVAL;
  if (o_normal && !grad) return I2SDF_EINVAL;
  if (o_lmask && !lmask) return I2SDF_EINVAL;
  CompArgs a{};
  a.beta_param = beta_param; a.beta_min = beta_min; a.z = z; a.ldz = ldz; a.sdf = sdf; a.rgb = rgb; a.grad = o_normal ? grad : nullptr;
  a.lmask = o_lmask ? lmask : nullptr; a.dnorm = dnorm; a.B = B; a.n = n;
  a.o_rgb = o_rgb; a.o_depth = o_depth; a.o_wsum = o_wsum; a.o_normal = o_normal; a.o_lmask = o_lmask; a.w_save = w_save; a.nsum_save = nsum_save;
  composite_fwd_kernel<<<(unsigned)((B + 3) / 4), 256, 0, (hipStream_t)stream>>>(a);
  return i2sdf_hip_check(hipGetLastError(), "composite_forward launch");
}

extern "C" int i2sdf_composite_backward(const float* beta_param, float beta_min, const float* z, int64_t ldz, const float* sdf,
                                        const float* rgb, const float* grad, const float* dnorm, const float* nsum_save, int64_t B, int32_t n,
                                        const float* g_rgb, const float* g_depth, const float* g_wsum, const float* g_normal,
                                        const float* g_lmask, float* sdf_bar, float* rgb_bar, float* grad_bar, float* lmask_bar,
                                        float* beta_partial, float* beta_grad_accum, void* stream) {
  if (B == 0) return I2SDF_OK;
  if (!beta_param || !z || !sdf || !rgb || !dnorm || !g_rgb || !sdf_bar || !rgb_bar || !beta_partial || B < 0 || n <= 0 || n > 64 * MAX_SEG)
    return I2SDF_EINVAL;
  if (grad_bar && (!grad || !nsum_save)) return I2SDF_EINVAL;
  CompArgs a{};
  a.beta_param = beta_param; a.beta_min = beta_min; a.z = z; a.ldz = ldz; a.sdf = sdf; a.rgb = rgb; a.grad = grad; a.dnorm = dnorm;
  a.nsum_save = const_cast<float*>(nsum_save); a.B = B; a.n = n;
  a.g_rgb = g_rgb; a.g_depth = g_depth; a.g_wsum = g_wsum; a.g_normal = g_normal; a.g_lmask = g_lmask;
  a.sdf_bar = sdf_bar; a.rgb_bar = rgb_bar; a.grad_bar = grad_bar; a.lmask_bar = lmask_bar; a.beta_bar_partial = beta_partial;
  hipStream_t st = (hipStream_t)stream;
  composite_bwd_kernel<<<(unsigned)((B + 3) / 4), 256, 0, st>>>(a);
  if (beta_grad_accum) beta_reduce_kernel<<<1, 1024, 0, st>>>(beta_partial, B, beta_param, beta_grad_accum);
  return i2sdf_hip_check(hipGetLastError(), "composite_backward launch");
}

extern "C" int64_t i2sdf_render_loss_scratch_floats(int64_t B) { return B < 0 ? 0 : ((B + 3) / 4) * (S_N + 1) + 4; }

// I2SDFLoss (model/network/__init__.py:289-406) on the outputs of a training render AND the backward of everything between those outputs
// and the per-sample gradients the MLP backward kernels start from (compositing backward :223-240,:120-125,:169,:204-219, the eikonal /
// smoothness outputs' backward :188-193, the seeds of the extra points) -- for an upstream gradient of 1; i2sdf_scale_seeds applies the one
// autograd delivers.  Two launches (the second a single workgroup).
extern "C" int i2sdf_render_loss_backward(const i2sdf_loss_cfg* cfg, int64_t B, int32_t n, int64_t n_pc, int64_t M_main, int64_t M_sdf, int64_t n_eik,
                                          const float* beta_param, float beta_min, const float* z, int64_t ldz, const float* sdf, const float* rgb_pts,
                                          const float* grad_pts, const float* dnorm, const float* nsum_save,
                                          const float* rgb, const float* depth, const float* wsum, const float* normal, const float* grad_theta,
                                          const float* diff_norm, const float* surface, const float* lmask,
                                          const float* gt_rgb, const float* gt_depth, const uint8_t* depth_mask, const float* gt_normal,
                                          const uint8_t* normal_mask, const float* gt_mask, const float* gt_lmask,
                                          float* scratch, float* losses, float* loss_value,
                                          float* g_rgb, float* g_depth, float* g_wsum, float* g_normal, float* g_grad_theta, float* g_diff_norm,
                                          float* g_surface, float* g_lmask,
                                          float* sdf_bar, float* rgb_bar, float* grad_bar, int32_t normal_term, float* lmask_bar, float* beta_grad,
                                          void* stream) {
  if (!cfg || cfg->exchange || B <= 0 || n <= 0 || n > 64 * MAX_SEG || M_main != B * n || M_sdf < M_main || n_pc < 0) return I2SDF_EINVAL;
  if (n_eik != 0 && n_eik != 3 * B) return I2SDF_EINVAL;
  if (M_sdf - M_main < n_eik + (surface ? n_pc : 0)) return I2SDF_EINVAL;
  if (!beta_param || !z || !sdf || !rgb_pts || !dnorm || !rgb || !depth || !wsum || !gt_rgb || !scratch || !losses || !g_rgb || !g_depth || !g_wsum ||
      !sdf_bar || !rgb_bar || !grad_bar || !beta_grad)
    return I2SDF_EINVAL;
  if ((gt_depth && !depth_mask) || (gt_normal && !normal_mask) || (normal && gt_normal && !g_normal)) return I2SDF_EINVAL;
  if (normal_term && (!grad_pts || !nsum_save || !normal)) return I2SDF_EINVAL;
  if (n_eik && (!grad_pts || !grad_theta)) return I2SDF_EINVAL;
  if (lmask_bar && !lmask) return I2SDF_EINVAL;
  RenderLossArgs a{};
  a.l.c = *cfg; a.l.B = B; a.l.n_pc = surface ? n_pc : 0;
  a.l.rgb = rgb; a.l.depth = depth; a.l.wsum = wsum; a.l.normal = normal; a.l.grad_theta = n_eik ? grad_theta : nullptr; a.l.diff_norm = n_eik ? diff_norm : nullptr;
  a.l.surface = surface; a.l.lmask = lmask; a.l.gt_rgb = gt_rgb; a.l.gt_depth = gt_depth; a.l.gt_normal = gt_normal; a.l.gt_mask = gt_mask;
  a.l.gt_lmask = gt_lmask; a.l.depth_mask = depth_mask; a.l.normal_mask = normal_mask;
  a.l.losses = losses; a.l.loss_value = loss_value;
  a.l.g_rgb = g_rgb; a.l.g_depth = g_depth; a.l.g_wsum = g_wsum; a.l.g_normal = g_normal; a.l.g_grad_theta = g_grad_theta; a.l.g_diff_norm = g_diff_norm;
  a.l.g_surface = g_surface; a.l.g_lmask = g_lmask;
  a.c.beta_param = beta_param; a.c.beta_min = beta_min; a.c.z = z; a.c.ldz = ldz; a.c.sdf = sdf; a.c.rgb = rgb_pts; a.c.grad = grad_pts; a.c.dnorm = dnorm;
  a.c.nsum_save = const_cast<float*>(nsum_save); a.c.B = B; a.c.n = n;
  a.c.sdf_bar = sdf_bar; a.c.rgb_bar = rgb_bar; a.c.grad_bar = normal_term ? grad_bar : nullptr; a.c.grad_bar_all = grad_bar; a.c.lmask_bar = lmask_bar;
  a.grad_all = grad_pts ? grad_pts + 3 * M_main : nullptr; a.M_main = M_main; a.M_sdf = M_sdf; a.n_eik = n_eik;
  a.part = scratch; a.beta_grad = beta_grad;
  hipStream_t st = (hipStream_t)stream;
  const unsigned nwg = (unsigned)((B + 3) / 4);
  render_loss_bwd_kernel<<<nwg, 256, 0, st>>>(a);
  render_loss_finish_kernel<<<1, 256, 0, st>>>(a, (int)nwg);
  return i2sdf_hip_check(hipGetLastError(), "render_loss_backward launch");
}

extern "C" int i2sdf_scale_seeds(const float* g, float* sdf_bar, int64_t n_sdf, float* grad_bar, int64_t n_grad, float* rgb_bar, int64_t n_rgb,
                                 float* lmask_bar, int64_t n_lmask, const float* beta_in, float* beta_out, void* stream) {
  if (!g || n_sdf < 0 || n_grad < 0 || n_rgb < 0 || n_lmask < 0 || (n_sdf && !sdf_bar) || (n_grad && !grad_bar) || (n_rgb && !rgb_bar) ||
      (n_lmask && !lmask_bar) || (beta_out && !beta_in))
    return I2SDF_EINVAL;
  const int64_t most = std::max(std::max(n_sdf, n_grad), std::max(n_rgb, n_lmask));
  const unsigned grid = (unsigned)std::min<int64_t>(std::max<int64_t>((most + 255) / 256, 1), 2048);
  scale_seeds_kernel<<<grid, 256, 0, (hipStream_t)stream>>>(g, sdf_bar, n_sdf, grad_bar, n_grad, rgb_bar, n_rgb, lmask_bar, n_lmask, beta_in, beta_out);
  return i2sdf_hip_check(hipGetLastError(), "scale_seeds launch");
}

// i2sdf_composite_forward + i2sdf_eikonal_outputs_forward in ONE launch (round 6: the training forward's two per-ray launches between the
// radiance net and the loss): grad_all (3B,3) = d sdf / d x of the extra points [uniform | near | neighbour] -> grad_theta (2B,3), diff_norm (B)
extern "C" int i2sdf_composite_forward_eik(const float* beta_param, float beta_min, const float* z, int64_t ldz, const float* sdf,
                                           const float* rgb, const float* grad, const float* lmask, const float* dnorm, int64_t B, int32_t n,
                                           float* o_rgb, float* o_depth, float* o_wsum, float* o_normal, float* o_lmask, float* w_save,
                                           float* nsum_save, const float* grad_all, float* grad_theta, float* diff_norm, void* stream) {
  if (B == 0) return I2SDF_OK;
  if (!beta_param || !z || !sdf || !rgb || !dnorm || !o_rgb || !o_depth || !o_wsum || B < 0 || n <= 0 || n > 64 * MAX_SEG) return I2SDF_EINVAL;
  if (o_normal && !grad) return I2SDF_EINVAL;
  if (o_lmask && !lmask) return I2SDF_EINVAL;
  if (!grad_all || !grad_theta || !diff_norm) return I2SDF_EINVAL;
  CompArgs a{};
  a.beta_param = beta_param; a.beta_min = beta_min; a.z = z; a.ldz = ldz; a.sdf = sdf; a.rgb = rgb; a.grad = o_normal ? grad : nullptr;
  a.lmask = o_lmask ? lmask : nullptr; a.dnorm = dnorm; a.B = B; a.n = n;
  a.o_rgb = o_rgb; a.o_depth = o_depth; a.o_wsum = o_wsum; a.o_normal = o_normal; a.o_lmask = o_lmask; a.w_save = w_save; a.nsum_save = nsum_save;
  a.eik_g = grad_all; a.eik_theta = grad_theta; a.eik_diff = diff_norm;
  composite_fwd_kernel<<<(unsigned)((B + 3) / 4), 256, 0, (hipStream_t)stream>>>(a);
  return i2sdf_hip_check(hipGetLastError(), "composite_forward_eik launch");
}
