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

struct AdamC { float omb1, beta2, omb2, eps, wd, step_size, bc2_sqrt, gscale; };

__device__ __forceinline__ void adam1(float& p, float g, float& m, float& v, const AdamC& c) {
  const float eps = c.eps, wd = c.wd, step_size = c.step_size, bc2_sqrt = c.bc2_sqrt;
  g *= c.gscale;
  if (wd != 0.f) g = fmaf(wd, p, g);
  m = m + c.omb1 * (g - m);
  v = c.beta2 * v + c.omb2 * g * g;
  const float den = sqrtf(v) / bc2_sqrt + eps;
  p = p - step_size * (m / den);
}

__global__ __launch_bounds__(256) void adam_kernel(float* __restrict__ p, const float* __restrict__ g, float* __restrict__ m,
                                                    float* __restrict__ v, int64_t n, AdamC c, int vec_ok) {
  const int64_t i = ((int64_t)blockIdx.x * 256 + threadIdx.x) * 4;
  if (i >= n) return;
  if (vec_ok && i + 4 <= n) {
    f32x4 pp = *reinterpret_cast<f32x4*>(p + i), mm = *reinterpret_cast<f32x4*>(m + i), vv = *reinterpret_cast<f32x4*>(v + i);
    const f32x4 gg = *reinterpret_cast<const f32x4*>(g + i);
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      float pk = pp[k], mk = mm[k], vk = vv[k];
      adam1(pk, gg[k], mk, vk, c);
      pp[k] = pk; mm[k] = mk; vv[k] = vk;
    }
    *reinterpret_cast<f32x4*>(p + i) = pp; *reinterpret_cast<f32x4*>(m + i) = mm; *reinterpret_cast<f32x4*>(v + i) = vv;
  } else {
    for (int64_t j = i; j < n && j < i + 4; ++j) adam1(p[j], g[j], m[j], v[j], c);
  }
}

}  // namespace

extern "C" int i2sdf_adam_step(float* params, const float* grads, float* exp_avg, float* exp_avg_sq, int64_t n, double lr, double beta1,
                               double beta2, double eps, double weight_decay, int64_t step, double grad_scale, void* stream) {
  if (n == 0) return I2SDF_OK;
  if (!params || !grads || !exp_avg || !exp_avg_sq || n < 0 || step < 1) return I2SDF_EINVAL;
  // hyper-parameters arrive as doubles and every derived constant is formed in double, then rounded once -- as torch does with its
  // Python floats (1 - beta2 formed in fp32 would be off by 5e-5 relative)
  const double bc1 = 1.0 - pow(beta1, (double)step), bc2 = 1.0 - pow(beta2, (double)step);
  const AdamC c{(float)(1.0 - beta1), (float)beta2, (float)(1.0 - beta2), (float)eps, (float)weight_decay, (float)(lr / bc1), (float)sqrt(bc2),
                (float)grad_scale};
  const int vec_ok = (((uintptr_t)params | (uintptr_t)grads | (uintptr_t)exp_avg | (uintptr_t)exp_avg_sq) & 15) == 0;
  const unsigned grid = (unsigned)((n + 1023) / 1024);
  adam_kernel<<<grid, 256, 0, (hipStream_t)stream>>>(params, grads, exp_avg, exp_avg_sq, n, c, vec_ok);
  return i2sdf_hip_check(hipGetLastError(), "adam_step launch");
}
