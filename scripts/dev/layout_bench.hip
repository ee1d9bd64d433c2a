// Microbenchmark for DESIGN.md "what comes next": the saved-tensor access pattern of the K-outer bf16x3 ops in the current
// point-major layout [M][256] versus a blocked layout [M/32][16 k-chunks][32 points][16 floats].
// Every lane reads (and writes) the 8 floats of its point that belong to one 16-wide k-chunk, as x3_load8 / x3_store8 do.
//   hipcc --offload-arch=gfx950 -O3 layout_bench.hip -o layout_bench && ./layout_bench
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
typedef float f32x4 __attribute__((ext_vector_type(4)));
constexpr int H = 256, L = 8, KC = 16;

template <bool BLOCKED>
__global__ __launch_bounds__(256) void k(const float* __restrict__ a, const float* __restrict__ b, float* __restrict__ c, float* __restrict__ d, int64_t Mp) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, hi = lane >> 5, p = lane & 31;
  const int64_t tile = (int64_t)blockIdx.x * 4 + wave;
  const int64_t m = tile * 32 + p;
  float acc = 0.f;
  for (int l = 0; l < L; ++l) {
    const int64_t lo = (int64_t)l * Mp * H;
#pragma unroll 4
    for (int kc = 0; kc < KC; ++kc) {
      int64_t o0, o1;
      if (BLOCKED) { o0 = lo + ((tile * KC + kc) * 32 + p) * 16 + 8 * hi; o1 = o0 + 4; }       // 32 B contiguous per lane, 2 KB per wave
      else { o0 = lo + m * H + 16 * kc + 4 * hi; o1 = o0 + 8; }
      const f32x4 x0 = *reinterpret_cast<const f32x4*>(a + o0), x1 = *reinterpret_cast<const f32x4*>(a + o1);
      const f32x4 y0 = *reinterpret_cast<const f32x4*>(b + o0), y1 = *reinterpret_cast<const f32x4*>(b + o1);
      *reinterpret_cast<f32x4*>(c + o0) = x0 * y0; *reinterpret_cast<f32x4*>(c + o1) = x1 * y1;
      *reinterpret_cast<f32x4*>(d + o0) = x0 + y0; *reinterpret_cast<f32x4*>(d + o1) = x1 + y1;
      acc += x0.x + y1.w;
    }
  }
  if (acc == 12345.678f) c[0] = acc;
}

int main() {
  const int64_t Mp = 98304;                       // 768 workgroups = 3 rounds
  const size_t n = (size_t)L * Mp * H;
  float *a, *b, *c, *d;
  hipMalloc(&a, n * 4); hipMalloc(&b, n * 4); hipMalloc(&c, n * 4); hipMalloc(&d, n * 4);
  hipMemset(a, 0, n * 4); hipMemset(b, 0, n * 4);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  for (int mode = 0; mode < 2; ++mode) {
    for (int rep = 0; rep < 3; ++rep) {
      hipEventRecord(e0);
      if (mode) k<true><<<Mp / 128, 256>>>(a, b, c, d, Mp); else k<false><<<Mp / 128, 256>>>(a, b, c, d, Mp);
      hipEventRecord(e1); hipEventSynchronize(e1);
      float ms; hipEventElapsedTime(&ms, e0, e1);
      if (rep == 2) printf("%s layout: %.3f ms, %.2f TB/s (2 tensors read + 2 written, %.2f GB)\n", mode ? "blocked [M/32][16][32][16]" : "point-major [M][256]    ", ms, 4.0 * n * 4 / ms / 1e9, 4.0 * n * 4 / 1e9);
    }
  }
  return 0;
}
