// Full-image inference as ONE call: the caller's chunk loop -- utils.split_input -> model(chunk) ->
// utils.merge_output (utils/__init__.py:35-84) inside VolumeRenderSystem.test_step / plotting (model/eval/recon.py:161-182) --
// enqueued from C++ on one stream.  Per chunk: ray set-up -> error-bounded sampler (device-side loop, per-chunk convergence
// flag exactly as the reference, whose sampler sees one chunk at a time) -> SDF MLP + d sdf/dx -> radiance MLP [-> light head]
// -> density/compositing, whose outputs go STRAIGHT into rows [lo, hi) of the (H*W, C) result tensors the plot writers read
// (no per-chunk dict, no torch.cat).  One chunk-sized workspace is reused by every chunk (stream order makes that safe); no
// allocation, no host synchronisation.
#include <hip/hip_runtime.h>
#include "plan.h"

using namespace i2sdf;

namespace {

struct Carve {
  float* p; int64_t used = 0;
  explicit Carve(float* base) : p(base) {}
  float* take(int64_t n) { n = (n + 63) / 64 * 64; float* r = p ? p + used : nullptr; used += n; return r; }   // 256-B aligned pieces
};

struct Pieces {
  float *cam, *dirs, *dnorm, *z, *samp, *sdf, *feat, *grad, *hs, *rgb, *lm;
  int64_t total;
};

Pieces carve(const i2sdf_plan* p, const i2sdf_sampler_cfg* sc, int64_t chunk, float* base) {
  const int n_z = sc->N_samples + sc->N_samples_extra + 2, n = n_z - 1;
  const int64_t M = chunk * n, Mp = (M + 127) / 128 * 128;
  const int H = p->sdf.d.hidden, L = p->sdf.d.n_lin;
  Carve c(base);
  Pieces q{};
  q.cam = c.take(chunk * 3); q.dirs = c.take(chunk * 3); q.dnorm = c.take(chunk);
  q.z = c.take(chunk * n_z);
  q.samp = c.take(i2sdf_sampler_workspace_floats(chunk));
  q.sdf = c.take(M); q.feat = c.take(Mp * p->F); q.grad = c.take(M * 3);
  q.hs = c.take((int64_t)(L - 1) * Mp * H);
  q.rgb = c.take(M * 3);
  q.lm = c.take(p->light.d.n_lin ? M : 0);
  q.total = c.used;
  return q;
}

}  // namespace

extern "C" int64_t i2sdf_render_image_workspace_floats(const i2sdf_plan* p, const i2sdf_sampler_cfg* sc, int64_t chunk) {
  if (!p || !sc || chunk <= 0) return 0;
  return carve(p, sc, chunk, nullptr).total;
}

extern "C" int i2sdf_render_image(const i2sdf_plan* p, const float* packed, const float* params, const i2sdf_sampler_cfg* sc,
                                  const float* uv, const float* pose, int32_t pose_is_quat, const float* intrinsics, int64_t P,
                                  int64_t chunk, const float* t_lin, const float* u_more, const float* u_final,
                                  const int32_t* extra_tab, float* workspace, float* o_rgb, float* o_depth, float* o_wsum,
                                  float* o_normal, float* o_lmask, float* o_z, int32_t* o_iters, void* stream) {
  if (P == 0) return I2SDF_OK;
  if (!p || !packed || !params || !sc || !uv || !pose || !intrinsics || !workspace || !o_rgb || !o_depth || !o_wsum || P < 0 || chunk <= 0)
    return I2SDF_EINVAL;
  const bool light = p->light.d.n_lin > 0;
  if (light != (o_lmask != nullptr) && light) return I2SDF_EINVAL;        // a light head needs its output
  const int n_z = sc->N_samples + sc->N_samples_extra + 2, n = n_z - 1;
  const Pieces q = carve(p, sc, chunk < P ? chunk : P, workspace);
  const float* beta_param = params + p->desc.off_beta;
  int ci = 0;
  for (int64_t lo = 0; lo < P; lo += chunk, ++ci) {
    const int64_t B = (P - lo < chunk) ? P - lo : chunk;
    const int64_t M = B * n, Mp = (M + 127) / 128 * 128;
    int rc = i2sdf_ray_setup_ex(uv + lo * 2, pose, pose_is_quat, intrinsics, 1, (int32_t)B, q.cam, q.dirs, q.dnorm, stream);
    if (rc) return rc;
    rc = i2sdf_sample_rays(p, packed, params, sc, q.cam, q.dirs, B, 0, t_lin, u_more, u_final, 0, extra_tab, nullptr, nullptr, nullptr, 0,
                           q.samp, q.z, n_z, nullptr, o_iters ? o_iters + ci : nullptr, stream);
    if (rc) return rc;
    if (o_z) {
      hipError_t e = hipMemcpyAsync(o_z + lo * n_z, q.z, sizeof(float) * B * n_z, hipMemcpyDeviceToDevice, (hipStream_t)stream);
      if (e != hipSuccess) return I2SDF_EHIP;
    }
    rc = i2sdf_sdf_forward_grad(p, packed, nullptr, q.cam, q.dirs, q.z, n_z, n, M, M, Mp, q.sdf, q.feat, q.grad, q.hs, nullptr, nullptr, stream);
    if (rc) return rc;
    rc = i2sdf_rgb_forward(p, packed, q.dirs, n, q.feat, M, Mp, q.rgb, nullptr, nullptr, stream);
    if (rc) return rc;
    if (light) {
      rc = i2sdf_light_forward(p, packed, q.feat, M, Mp, q.lm, nullptr, stream);
      if (rc) return rc;
    }
    rc = i2sdf_composite_forward(beta_param, p->desc.beta_min, q.z, n_z, q.sdf, q.rgb, o_normal ? q.grad : nullptr, light ? q.lm : nullptr,
                                 q.dnorm, B, n, o_rgb + lo * 3, o_depth + lo, o_wsum + lo, o_normal ? o_normal + lo * 3 : nullptr,
                                 light ? o_lmask + lo : nullptr, nullptr, nullptr, stream);
    if (rc) return rc;
  }
  return I2SDF_OK;
}
