/* Plain C caller of the C ABI (include/i2sdf.h): no Python, no torch -- the library's boundary is pointers and sizes.
 *
 *   sdf_volume <net_desc.bin> <params.bin> <axes.bin> <out.bin>
 *
 * net_desc.bin : the bytes of an i2sdf_net_desc (what a host binding fills from the reference's yaml `model:` node)
 * params.bin   : n_params floats, the reference's state_dict order (a checkpoint's `model.*` tensors, flattened)
 * axes.bin     : int32 nx, ny, nz, then nx + ny + nz floats (utils/plots.py get_grid_uniform / get_grid axes as float32)
 * out.bin      : nx*ny*nz floats, the (x, y, z) volume measure.marching_cubes is run on (model/eval/recon.py:53-54,94)
 *
 * Build (examples/build.sh): gcc -D__HIP_PLATFORM_AMD__ -I/opt/rocm/include -Iinclude examples/sdf_volume.c
 *                            -L/opt/rocm/lib -lamdhip64 -Li2sdf_amd/lib -li2sdf_hip -Wl,-rpath,...
 * tests/test_gpu_c_example.py runs it against the Python module's sdf_volume (bit-identical). */
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <hip/hip_runtime_api.h>
#include "i2sdf.h"

#define CHECK_HIP(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); return 2; } } while (0)
#define CHECK_LIB(x) do { int r_ = (x); if (r_ != I2SDF_OK) { fprintf(stderr, "%s: %s (%s)\n", #x, i2sdf_strerror(r_), i2sdf_last_hip_error()); return 3; } } while (0)

static void* slurp(const char* path, size_t* n) {
  FILE* f = fopen(path, "rb");
  if (!f) { perror(path); return NULL; }
  fseek(f, 0, SEEK_END);
  long sz = ftell(f);
  fseek(f, 0, SEEK_SET);
  void* buf = malloc((size_t)sz);
  if (fread(buf, 1, (size_t)sz, f) != (size_t)sz) { fclose(f); free(buf); return NULL; }
  fclose(f);
  *n = (size_t)sz;
  return buf;
}

int main(int argc, char** argv) {
  if (argc != 5) { fprintf(stderr, "usage: %s net_desc.bin params.bin axes.bin out.bin\n", argv[0]); return 1; }
  size_t n_desc = 0, n_par = 0, n_ax = 0;
  i2sdf_net_desc* desc = (i2sdf_net_desc*)slurp(argv[1], &n_desc);
  float* params_h = (float*)slurp(argv[2], &n_par);
  char* axes_h = (char*)slurp(argv[3], &n_ax);
  if (!desc || !params_h || !axes_h || n_desc != sizeof(i2sdf_net_desc) || n_par != (size_t)desc->n_params * sizeof(float)) {
    fprintf(stderr, "bad input files (desc %zu of %zu bytes, params %zu bytes for %lld floats)\n", n_desc, sizeof(i2sdf_net_desc), n_par,
            desc ? (long long)desc->n_params : -1LL);
    return 1;
  }
  int32_t dims[3];
  memcpy(dims, axes_h, sizeof dims);
  const int32_t nx = dims[0], ny = dims[1], nz = dims[2];
  const int64_t total = (int64_t)nx * ny * nz;
  if (n_ax != sizeof dims + (size_t)(nx + ny + nz) * sizeof(float)) { fprintf(stderr, "bad axes file\n"); return 1; }

  i2sdf_plan* plan = NULL;
  CHECK_LIB(i2sdf_plan_create(desc, &plan));
  CHECK_LIB(i2sdf_plan_set_option(plan, I2SDF_OPT_SDF_FWD_BF16X3, 1));      /* fp32-accurate on the bf16 matrix pipe */

  float *params_d, *packed_d, *axes_d, *out_d, *ws_d;
  const int64_t chunk = 1 << 20;
  CHECK_HIP(hipMalloc((void**)&params_d, n_par));
  CHECK_HIP(hipMalloc((void**)&packed_d, (size_t)i2sdf_plan_pack_floats(plan) * sizeof(float)));
  CHECK_HIP(hipMalloc((void**)&axes_d, (size_t)(nx + ny + nz) * sizeof(float)));
  CHECK_HIP(hipMalloc((void**)&out_d, (size_t)total * sizeof(float)));
  CHECK_HIP(hipMalloc((void**)&ws_d, (size_t)i2sdf_sdf_grid_workspace_floats(chunk) * sizeof(float)));
  CHECK_HIP(hipMemcpy(params_d, params_h, n_par, hipMemcpyHostToDevice));
  CHECK_HIP(hipMemcpy(axes_d, axes_h + sizeof dims, (size_t)(nx + ny + nz) * sizeof(float), hipMemcpyHostToDevice));
  CHECK_HIP(hipMemset(packed_d, 0, (size_t)i2sdf_plan_pack_floats(plan) * sizeof(float)));

  hipStream_t stream;
  CHECK_HIP(hipStreamCreate(&stream));
  CHECK_LIB(i2sdf_pack_weights(plan, params_d, packed_d, stream));          /* weight-norm + stream packing, after every update */
  CHECK_LIB(i2sdf_sdf_grid(plan, packed_d, axes_d, axes_d + nx, axes_d + nx + ny, nx, ny, nz, I2SDF_GRID_ORDER_VOLUME, NULL, NULL, 0, total,
                           out_d, ws_d, chunk, stream));
  CHECK_HIP(hipStreamSynchronize(stream));

  float* out_h = (float*)malloc((size_t)total * sizeof(float));
  CHECK_HIP(hipMemcpy(out_h, out_d, (size_t)total * sizeof(float), hipMemcpyDeviceToHost));
  float lo = out_h[0], hi = out_h[0];
  for (int64_t i = 1; i < total; ++i) { if (out_h[i] < lo) lo = out_h[i]; if (out_h[i] > hi) hi = out_h[i]; }
  FILE* f = fopen(argv[4], "wb");
  if (!f || fwrite(out_h, sizeof(float), (size_t)total, f) != (size_t)total) { perror(argv[4]); return 1; }
  fclose(f);
  printf("i2sdf %d: %d x %d x %d volume, sdf in [%g, %g]\n", i2sdf_version(), nx, ny, nz, lo, hi);
  i2sdf_plan_destroy(plan);
  hipFree(params_d); hipFree(packed_d); hipFree(axes_d); hipFree(out_d); hipFree(ws_d);
  free(out_h); free(desc); free(params_h); free(axes_h);
  return 0;
}
