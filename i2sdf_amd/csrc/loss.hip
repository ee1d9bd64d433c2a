// I2SDFLoss forward + gradient w.r.t. the render outputs in two tiny launches:
// model/network/__init__.py:289-406 with its quirks (angular term = the same L1 normal term, :368-369).
// Replaces ~100 element-wise torch kernels (forward and autograd backward) per training step.
#include <algorithm>
#include "loss_dev.h"

using namespace i2sdf;

int i2sdf_hip_check(hipError_t e, const char* what);

namespace {

__global__ __launch_bounds__(256) void loss_partial_kernel(LossArgs a) {
  float s[S_N];
#pragma unroll
  for (int i = 0; i < S_N; ++i) s[i] = 0.f;
  const int64_t stride = (int64_t)gridDim.x * 256;
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < a.B; i += stride) loss_ray_terms(a, i, s);
  if (a.grad_theta)
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < 2 * a.B; i += stride) s[S_EIK] += loss_eik_term(a.grad_theta + i * 3);
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

// the reported values (workgroup 0 of the gradient launch): loss_dev.h: loss_values
__global__ __launch_bounds__(256) void loss_grad_kernel(LossArgs a) {
  __shared__ float tot[S_N], cnt[C_N];
  if (a.reduced) {
    if (threadIdx.x < S_N) tot[threadIdx.x] = a.sums[threadIdx.x];
    if (threadIdx.x < C_N) cnt[threadIdx.x] = a.cnt[threadIdx.x];
  } else {
    loss_reduce(a, tot, cnt);
  }
  __syncthreads();
  if (blockIdx.x == 0 && threadIdx.x == 0) loss_values(a, tot, cnt);
  const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
  const float B = cnt[C_B];
  if (i < a.B) {
    LossSeeds g;
    loss_ray_grads(a, i, cnt, g);
#pragma unroll
    for (int k = 0; k < 3; ++k) a.g_rgb[i * 3 + k] = g.rgb[k];
    a.g_depth[i] = g.depth;
    a.g_wsum[i] = g.wsum;
    if (a.g_normal) { a.g_normal[i * 3] = g.normal[0]; a.g_normal[i * 3 + 1] = g.normal[1]; a.g_normal[i * 3 + 2] = g.normal[2]; }
    if (a.g_diff_norm) a.g_diff_norm[i] = g.diff_norm;
    if (a.g_lmask) a.g_lmask[i] = g.lmask;
  }
  if (a.g_grad_theta && i < 2 * a.B) {
    float o[3];
    loss_eik_grad(a, a.grad_theta + i * 3, B, o);
    a.g_grad_theta[i * 3] = o[0]; a.g_grad_theta[i * 3 + 1] = o[1]; a.g_grad_theta[i * 3 + 2] = o[2];
  }
  if (a.g_surface && i < a.n_pc) a.g_surface[i] = loss_surface_grad(a, a.surface[i], cnt[C_NPC]);
}

// ---------------------------------------------------------------------------------------------------------------
// Eikonal / smoothness outputs of the network (model/network/__init__.py:188-193): from the gradients of the 3B extra points
// [B uniform | B near-surface | B neighbours]   grad_theta = rows [0, 2B),
//   diff_norm[i] = || normalize(g[B+i]) - normalize(g[2B+i]) ||_2,   normalize(v) = v / max(||v||, 1e-6)  (F.normalize eps)
// and the backward of both in one launch.  In torch these are ~8 forward and ~25 autograd-backward kernels on (2B,3) tensors.
// Conventions of the torch backward formulas are kept: d||x||/dx = 0 at x = 0; no gradient through ||v|| where ||v|| < eps.
// ---------------------------------------------------------------------------------------------------------------
// (unit3 / unit3_bwd / eik_out_bwd_point with FMA contraction off: loss_dev.h)
#pragma clang fp contract(off)

__global__ __launch_bounds__(256) void eik_out_fwd_kernel(const float* __restrict__ g, int64_t B, float* __restrict__ theta,
                                                           float* __restrict__ diff) {
  const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (i >= B) return;
  eik_out_fwd_point(g, B, i, theta, diff);
}

__global__ __launch_bounds__(256) void eik_out_bwd_kernel(const float* __restrict__ g, const float* __restrict__ theta_bar,
                                                           const float* __restrict__ diff_bar, int64_t B, float* __restrict__ g_bar) {
  const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (i >= B) return;
  float th0[3] = {0.f, 0.f, 0.f}, th1[3] = {0.f, 0.f, 0.f}, o0[3], o1[3], o2[3];
  if (theta_bar) {
#pragma unroll
    for (int c = 0; c < 3; ++c) { th0[c] = theta_bar[i * 3 + c]; th1[c] = theta_bar[(B + i) * 3 + c]; }
  }
  eik_out_bwd_point(g, B, i, th0, th1, diff_bar != nullptr, diff_bar ? diff_bar[i] : 0.f, o0, o1, o2);
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
