// the two bulk re-uses of the SDF forward outside the training step.
//
//  i2sdf_sdf_grid   -- the SDF volume marching cubes is run on (model/eval/recon.py:46-51 coarse 100^3 grid, :75-90 the
//                      PCA-aligned fine grid, up to 512 along the shortest axis).  The reference materialises every grid
//                      point on the host (np.meshgrid -> (n,3) tensor, utils/plots.py:440-489; 1.6 GB at 512^3), ships it
//                      through a 32-worker DataLoader in 2 M-point batches and copies every batch of values back
//                      (model/eval/recon.py:96-103).  Here the three axis vectors are the whole input: points are generated
//                      in HBM one chunk ahead of the SDF kernel that consumes them (12 B written + 12 B read per point next to
//                      0.9 MFLOP of MLP), the values land in one device buffer, in the reference's flat order or directly in
//                      the (x, y, z) volume order measure.marching_cubes wants.
//  i2sdf_pdf_update -- VolumeRenderSystem.update_pdf (model/trainer/recon.py:142-152) fused with the error it is fed
//                      (:195-199, :248-252): |clamp(rgb) - clamp(gt)| mean over channels, or |depth - gt|; clamp to pdf_max,
//                      prune below pdf_prune, scatter through the pixel->point links.  One launch instead of ~10.
#include <hip/hip_runtime.h>
#include "../../include/i2sdf.h"

int i2sdf_hip_check(hipError_t e, const char* what);

namespace {

struct GridXf { float r[9]; float t[3]; int on; };

__global__ __launch_bounds__(256) void grid_points_kernel(const float* __restrict__ x, const float* __restrict__ y,
                                                           const float* __restrict__ z, int nx, int ny, int nz, int order, GridXf xf,
                                                           int64_t start, int64_t count, float* __restrict__ pts) {
  const int64_t t = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (t >= count) return;
  const int64_t g = start + t;
  const int k = (int)(g % nz);
  const int64_t q = g / nz;
  int i, j;                                    // i: index into y, j: index into x
  if (order == I2SDF_GRID_ORDER_MESHGRID) { j = (int)(q % nx); i = (int)(q / nx); }      // np.meshgrid(x, y, z).ravel(): (y, x, z)
  else                                    { i = (int)(q % ny); j = (int)(q / ny); }      // volume order (x, y, z)
  float px = x[j], py = y[i], pz = z[k];
  if (xf.on) {
    const float ox = fmaf(xf.r[2], pz, fmaf(xf.r[1], py, xf.r[0] * px)) + xf.t[0];
    const float oy = fmaf(xf.r[5], pz, fmaf(xf.r[4], py, xf.r[3] * px)) + xf.t[1];
    const float oz = fmaf(xf.r[8], pz, fmaf(xf.r[7], py, xf.r[6] * px)) + xf.t[2];
    px = ox; py = oy; pz = oz;
  }
  pts[3 * t + 0] = px; pts[3 * t + 1] = py; pts[3 * t + 2] = pz;
}

__global__ __launch_bounds__(256) void pdf_update_kernel(const float* __restrict__ pred, const float* __restrict__ target, int channels,
                                                          const int64_t* __restrict__ idx, int64_t idx0, int64_t n,
                                                          const int64_t* __restrict__ links, int64_t n_links, float pdf_max, int has_max,
                                                          float pdf_prune, float* __restrict__ pdf, int64_t n_pdf,
                                                          int32_t* __restrict__ n_bad) {
  const int64_t t = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (t >= n) return;
  const int64_t pix = idx ? idx[t] : idx0 + t;
  if (pix < 0 || pix >= n_links) { if (n_bad) atomicAdd(n_bad, 1); return; }
  const int64_t link = links[pix];
  if (link == -1) return;
  if (link < 0 || link >= n_pdf) { if (n_bad) atomicAdd(n_bad, 1); return; }
  float v;
  if (channels == 1) {
    v = fabsf(pred[t] - target[t]);
  } else {
    float s = 0.f;
    for (int c = 0; c < channels; ++c) {
      const float a = fminf(fmaxf(pred[t * channels + c], 0.f), 1.f), b = fminf(fmaxf(target[t * channels + c], 0.f), 1.f);
      s += fabsf(a - b);
    }
    v = s / (float)channels;
  }
  if (has_max) v = fminf(v, pdf_max);          // value.clamp(max=pdf_max)  (:146-147)
  if (v < pdf_prune) v = 0.f;                   // PDF pruning              (:148)
  pdf[link] = v;                                // links are unique per pixel (dataset/train_dataset.py:124-127)
}

}  // namespace

extern "C" int64_t i2sdf_sdf_grid_workspace_floats(int64_t chunk_points) { return chunk_points > 0 ? 3 * chunk_points : 0; }

extern "C" int i2sdf_sdf_grid(const i2sdf_plan* plan, const float* packed, const float* x, const float* y, const float* z, int32_t nx,
                              int32_t ny, int32_t nz, int32_t order, const float* rot, const float* trans, int64_t first, int64_t count,
                              float* sdf_out, float* workspace, int64_t chunk_points, void* stream) {
  if (!plan || !packed || nx < 0 || ny < 0 || nz < 0 || first < 0 || count < 0) return I2SDF_EINVAL;
  if (order != I2SDF_GRID_ORDER_MESHGRID && order != I2SDF_GRID_ORDER_VOLUME) return I2SDF_EINVAL;
  const int64_t total = (int64_t)nx * ny * nz;
  if (first + count > total) return I2SDF_EINVAL;
  if (count == 0) return I2SDF_OK;
  if (!x || !y || !z || !sdf_out || !workspace || chunk_points <= 0) return I2SDF_EINVAL;
  GridXf xf{};
  xf.on = (rot || trans) ? 1 : 0;
  for (int a = 0; a < 9; ++a) xf.r[a] = rot ? rot[a] : ((a % 4 == 0) ? 1.f : 0.f);
  for (int a = 0; a < 3; ++a) xf.t[a] = trans ? trans[a] : 0.f;
  hipStream_t st = (hipStream_t)stream;
  for (int64_t lo = 0; lo < count; lo += chunk_points) {
    const int64_t n = count - lo < chunk_points ? count - lo : chunk_points;
    grid_points_kernel<<<(unsigned)((n + 255) / 256), 256, 0, st>>>(x, y, z, nx, ny, nz, order, xf, first + lo, n, workspace);
    if (int rc = i2sdf_hip_check(hipGetLastError(), "grid_points_kernel")) return rc;
    // same stream: the next chunk's generator runs after this chunk's SDF kernel has consumed the workspace
    if (int rc = i2sdf_sdf_forward(plan, packed, workspace, n, sdf_out + lo, nullptr, 0, stream)) return rc;
  }
  return I2SDF_OK;
}

extern "C" int i2sdf_pdf_update(const float* pred, const float* target, int32_t channels, const int64_t* pixel_idx, int64_t first_pixel,
                                int64_t n, const int64_t* pointlinks, int64_t n_links, double pdf_max, double pdf_prune, float* pdf,
                                int64_t n_pdf, int32_t* n_bad, void* stream) {
  if (n == 0) return I2SDF_OK;
  if (!pred || !target || !pointlinks || !pdf || n < 0 || n_links <= 0 || n_pdf <= 0 || channels < 1 || channels > 4) return I2SDF_EINVAL;
  if (!pixel_idx && first_pixel < 0) return I2SDF_EINVAL;
  const int has_max = pdf_max == pdf_max && pdf_max < 3.0e38;      // NaN / +inf = "pdf_max is None"
  pdf_update_kernel<<<(unsigned)((n + 255) / 256), 256, 0, (hipStream_t)stream>>>(pred, target, channels, pixel_idx, first_pixel, n, pointlinks,
                                                                                   n_links, (float)pdf_max, has_max, (float)pdf_prune, pdf, n_pdf,
                                                                                   n_bad);
  return i2sdf_hip_check(hipGetLastError(), "pdf_update_kernel");
}
