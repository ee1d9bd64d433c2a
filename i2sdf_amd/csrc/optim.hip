// Adam over a flat fp32 parameter buffer in ONE launch (the reference's optimizer is torch.optim.Adam(params, lr, eps=1e-15),
// model/trainer/recon.py:201-203; torch's foreach implementation is ~60 sub-5-us launches per step on the 47 parameter tensors).
// Same update rule and operation order as torch.optim.Adam (non-amsgrad, maximize=False):
//   g   = grad (+ weight_decay * p)
//   m   = m + (1 - beta1) * (g - m)                       (Tensor.lerp_)
//   v   = beta2 * v + (1 - beta2) * g * g                 (mul_ + addcmul_)
//   den = sqrt(v) / sqrt(1 - beta2^t) + eps
//   p   = p - (lr / (1 - beta1^t)) * (m / den)            (addcdiv_)
// Bias corrections are computed on the host in double, as torch does.  HBM-bound: 16 B read + 12 B written per parameter.
#include <hip/hip_runtime.h>
#include "../../include/i2sdf.h"

int i2sdf_hip_check(hipError_t e, const char* what);

namespace {

typedef float f32x4 __attribute__((ext_vector_type(4)));

__device__ __forceinline__ void adam1(float& p, float g, float& m, float& v, float beta1, float beta2, float eps, float wd,
                                      float step_size, float bc2_sqrt, float gscale) {
  g *= gscale;
  if (wd != 0.f) g = fmaf(wd, p, g);
  m = m + (1.0f - beta1) * (g - m);
  v = beta2 * v + (1.0f - beta2) * g * g;
  const float den = sqrtf(v) / bc2_sqrt + eps;
  p = p - step_size * (m / den);
}

__global__ __launch_bounds__(256) void adam_kernel(float* __restrict__ p, const float* __restrict__ g, float* __restrict__ m,
                                                    float* __restrict__ v, int64_t n, float beta1, float beta2, float eps, float wd,
                                                    float step_size, float bc2_sqrt, float gscale, int vec_ok) {
  const int64_t i = ((int64_t)blockIdx.x * 256 + threadIdx.x) * 4;
  if (i >= n) return;
  if (vec_ok && i + 4 <= n) {
    f32x4 pp = *reinterpret_cast<f32x4*>(p + i), mm = *reinterpret_cast<f32x4*>(m + i), vv = *reinterpret_cast<f32x4*>(v + i);
    const f32x4 gg = *reinterpret_cast<const f32x4*>(g + i);
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      float pk = pp[k], mk = mm[k], vk = vv[k];
      adam1(pk, gg[k], mk, vk, beta1, beta2, eps, wd, step_size, bc2_sqrt, gscale);
      pp[k] = pk; mm[k] = mk; vv[k] = vk;
    }
    *reinterpret_cast<f32x4*>(p + i) = pp; *reinterpret_cast<f32x4*>(m + i) = mm; *reinterpret_cast<f32x4*>(v + i) = vv;
  } else {
    for (int64_t j = i; j < n && j < i + 4; ++j) adam1(p[j], g[j], m[j], v[j], beta1, beta2, eps, wd, step_size, bc2_sqrt, gscale);
  }
}

}  // namespace

extern "C" int i2sdf_adam_step(float* params, const float* grads, float* exp_avg, float* exp_avg_sq, int64_t n, float lr, float beta1,
                               float beta2, float eps, float weight_decay, int64_t step, float grad_scale, void* stream) {
  if (n == 0) return I2SDF_OK;
  if (!params || !grads || !exp_avg || !exp_avg_sq || n < 0 || step < 1) return I2SDF_EINVAL;
  const double bc1 = 1.0 - pow((double)beta1, (double)step), bc2 = 1.0 - pow((double)beta2, (double)step);
  const float step_size = (float)((double)lr / bc1), bc2_sqrt = (float)sqrt(bc2);
  const int vec_ok = (((uintptr_t)params | (uintptr_t)grads | (uintptr_t)exp_avg | (uintptr_t)exp_avg_sq) & 15) == 0;
  const unsigned grid = (unsigned)((n + 1023) / 1024);
  adam_kernel<<<grid, 256, 0, (hipStream_t)stream>>>(params, grads, exp_avg, exp_avg_sq, n, beta1, beta2, eps, weight_decay, step_size, bc2_sqrt,
                                                     grad_scale, vec_ok);
  return i2sdf_hip_check(hipGetLastError(), "adam_step launch");
}
