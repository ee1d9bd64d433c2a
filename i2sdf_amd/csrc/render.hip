// Per-ray kernels: ray generation, Laplace density + log-space alpha compositing (forward and backward).
// One wave (64 lanes) per ray, the ray's samples live in registers, prefix sums by cross-lane shuffles.
// HBM-bound (a few KB per ray); see DESIGN.md for the byte counts.
#include "plan.h"

using namespace i2sdf;

int i2sdf_hip_check(hipError_t e, const char* what);

namespace {

constexpr int MAX_SEG = 4;      // up to 256 samples per ray

__device__ __forceinline__ float wave_incl_scan(float v, int lane) {
#pragma unroll
  for (int o = 1; o < 64; o <<= 1) {
    const float t = __shfl_up(v, o);
    if (lane >= o) v += t;
  }
  return v;
}
__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
  return v;
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
  }
}

// SURVEY appendix A.5.  The normal and light-mask composites use w.detach() in training
// (model/network/__init__.py:169,207), so they contribute to grad_bar / lmask_bar only.
__global__ __launch_bounds__(256) void composite_bwd_kernel(CompArgs a) {
  const int lane = threadIdx.x & 63;
  const int64_t ray = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (ray >= a.B) return;
  const float beta = fabsf(a.beta_param[0]) + a.beta_min;
  const float inv_beta = 1.0f / beta;
  const float* zr = a.z + ray * a.ldz;
  const int n = a.n;
  const int nseg = (n + 63) / 64;
  const float gr0 = a.g_rgb[ray * 3 + 0], gr1 = a.g_rgb[ray * 3 + 1], gr2 = a.g_rgb[ray * 3 + 2];
  const float gd = a.g_depth ? a.g_depth[ray] / fmaxf(a.dnorm[ray], 1e-6f) : 0.f;
  const float gw = a.g_wsum ? a.g_wsum[ray] : 0.f;
  // d normal_values / d nsum
  float gN[3] = {0, 0, 0};
  if (a.g_normal && a.grad_bar) {
    const float N0 = a.nsum_save[ray * 3 + 0], N1 = a.nsum_save[ray * 3 + 1], N2 = a.nsum_save[ray * 3 + 2];
    const float nn = sqrtf(N0 * N0 + N1 * N1 + N2 * N2);
    if (nn > 1e-12f) {
      const float inv = 1.0f / nn;
      const float o0 = N0 * inv, o1 = N1 * inv, o2 = N2 * inv;
      const float g0 = a.g_normal[ray * 3 + 0], g1 = a.g_normal[ray * 3 + 1], g2 = a.g_normal[ray * 3 + 2];
      const float dot = o0 * g0 + o1 * g1 + o2 * g2;
      gN[0] = (g0 - o0 * dot) * inv; gN[1] = (g1 - o1 * dot) * inv; gN[2] = (g2 - o2 * dot) * inv;
    } else {
      const float inv = 1e12f;
      gN[0] = a.g_normal[ray * 3 + 0] * inv; gN[1] = a.g_normal[ray * 3 + 1] * inv; gN[2] = a.g_normal[ray * 3 + 2] * inv;
    }
  }
  const float gl = (a.g_lmask && a.lmask_bar) ? a.g_lmask[ray] : 0.f;
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
  bb = wave_sum(bb);
  if (lane == 0) a.beta_bar_partial[ray] = bb;
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
