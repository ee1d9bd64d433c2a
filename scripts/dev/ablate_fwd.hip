// Dev-only ablation of the fused SDF forward kernel: which part of the non-MFMA time matters?
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -I include -I i2sdf_amd/csrc [-DI2SDF_ABL_*] scripts/dev/ablate_fwd.hip -o abl && ./abl
#include <stdio.h>
#include <stdlib.h>
#include <vector>
#include "epi.h"
using namespace i2sdf;

struct IdEpi { __device__ __forceinline__ void prefetch(int) {} __device__ __forceinline__ void elem(int, f32x16&, int) {} __device__ __forceinline__ void apply(int, f32x16&) {} };

template <int VAR>
__global__ __launch_bounds__(256) void k(const float* __restrict__ stream, int n_stages, int L, int skip, const float* __restrict__ pts, int64_t M,
                                         float* __restrict__ out) {
  constexpr int H = 256, NT = 8, KC = 32, PEC = 5;
  extern __shared__ __attribute__((aligned(16))) float lds[];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, hi = lane >> 5;
  const int64_t m = ((int64_t)blockIdx.x * 4 + wave) * 32 + (lane & 31);
  const float px = pts[m * 3], py = pts[m * 3 + 1], pz = pts[m * 3 + 2];
  float pe[PEC * 4];
  {
    float full[PEC * 8];
    if (VAR & 1) {
#pragma unroll
      for (int i = 0; i < PEC * 8; ++i) full[i] = px * (float)i + py + pz;
    } else pe_full<6>(px, py, pz, full);
    to_b_layout<PEC>(full, pe, hi);
  }
  WStream ws;
  ws.begin(stream, lds, n_stages, tid);
  f32x16 acc[NT];
  float h[NT * 16];
  SoftplusEpi sp{nullptr, hi, true};
  IdEpi id;
  if (VAR & 2) dense_op_epi<NT, PEC, NT * 4, 0, 0, IdEpi>(ws, pe, acc, id, tid); else dense_op_epi<NT, PEC, NT * 4, 0, 0, SoftplusEpi>(ws, pe, acc, sp, tid);
  commit_tiles<NT>(acc, h);
  for (int l = 1; l < L - 1; ++l) {
    if (l == skip) {
      float u[(KC + PEC) * 4];
#pragma unroll
      for (int i = 0; i < KC * 4; ++i) u[i] = h[i] * RS2;
#pragma unroll
      for (int i = 0; i < PEC * 4; ++i) u[KC * 4 + i] = pe[i] * RS2;
      if (VAR & 2) dense_op_epi<NT, KC + PEC, NT * 4, 0, 0, IdEpi>(ws, u, acc, id, tid); else dense_op_epi<NT, KC + PEC, NT * 4, 0, 0, SoftplusEpi>(ws, u, acc, sp, tid);
    } else {
      if (VAR & 2) dense_op_epi<NT, KC, NT * 4, 0, 0, IdEpi>(ws, h, acc, id, tid); else dense_op_epi<NT, KC, NT * 4, 0, 0, SoftplusEpi>(ws, h, acc, sp, tid);
    }
    commit_tiles<NT>(acc, h);
  }
  float s[1];
  rowvec_op<1, KC>(ws, h, s, tid);
  if (hi == 0) out[m] = s[0];
}

int main() {
  const int L = 9, skip = 4;
  const int64_t M = 131072;
  const int ns = sdf_fwd_stages(256, 256, 5, L, true, false);
  const size_t stream_floats = (size_t)(ns + 2) * STAGE_FLOATS;
  std::vector<float> hs(stream_floats), hp(M * 3);
  srand(1);
  for (auto& v : hs) v = (rand() / (float)RAND_MAX - 0.5f) * 0.1f;
  for (auto& v : hp) v = (rand() / (float)RAND_MAX - 0.5f) * 4.f;
  float *ds, *dp, *dout;
  hipMalloc(&ds, stream_floats * 4); hipMalloc(&dp, M * 12); hipMalloc(&dout, M * 4);
  hipMemcpy(ds, hs.data(), stream_floats * 4, hipMemcpyHostToDevice); hipMemcpy(dp, hp.data(), M * 12, hipMemcpyHostToDevice);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  auto run = [&](auto kern, const char* name) {
    for (int i = 0; i < 3; ++i) kern<<<M / 128, 256, LDS_BYTES>>>(ds, ns, L, skip, dp, M, dout);
    hipEventRecord(e0);
    for (int i = 0; i < 10; ++i) kern<<<M / 128, 256, LDS_BYTES>>>(ds, ns, L, skip, dp, M, dout);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1); ms /= 10;
    const double macs = 39.0 * 256 + 256.0 * 256 * 2 + 217.0 * 256 + 256.0 * 256 * 4 + 256;
    printf("%-28s %.3f ms  %.1f TFLOP/s\n", name, ms, 2 * macs * M / ms / 1e9);
  };
  run(k<0>, "full");
  run(k<1>, "no sincos");
  run(k<2>, "identity epilogue");
  run(k<3>, "no sincos + identity epi");
  return 0;
}
