// Microbenchmark (round 2): saved-tensor access patterns of the K-outer bf16x3 ops at the REAL occupancy of those kernels
// (one 4-wave workgroup per CU, forced by a 128 KB LDS allocation), 2 tensors read + 2 written per k-chunk as in backward sweep 1.
//   mode 0  point-major [M][256]: lane (p,hi) touches 16 B at row p, 1 KB stride between lanes (what the library does)
//   mode 1  blocked [M/32][16][32][16]: 2 KB contiguous per wave and k-chunk, lanes 64 B apart
//   mode 2  lane-native [M/32][16][2][64][4]: every wave instruction moves 1 KB contiguous (lane i <-> 16 B at base + 16 i)
// AHEAD = k-chunks of loads kept in flight before their use (register ring).
//   hipcc --offload-arch=gfx950 -O3 layout_bench2.hip -o layout_bench2 && ./layout_bench2
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
typedef float f32x4 __attribute__((ext_vector_type(4)));
constexpr int H = 256, L = 8, KC = 16;

template <int MODE>
__device__ __forceinline__ void offs(int64_t lo, int64_t tile, int64_t m, int kc, int lane, int64_t& o0, int64_t& o1) {
  const int hi = lane >> 5, p = lane & 31;
  if (MODE == 0) { o0 = lo + m * H + 16 * kc + 4 * hi; o1 = o0 + 8; }
  else if (MODE == 1) { o0 = lo + ((tile * KC + kc) * 32 + p) * 16 + 8 * hi; o1 = o0 + 4; }
  else { o0 = lo + ((tile * KC + kc) * 2) * 256 + lane * 4; o1 = o0 + 256; }
}

template <int MODE, int AHEAD>
__global__ __launch_bounds__(256) void k(const float* __restrict__ a, const float* __restrict__ b, float* __restrict__ c, float* __restrict__ d, int64_t Mp, int rounds) {
  extern __shared__ float lds[];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  float acc = 0.f;
  for (int r = 0; r < rounds; ++r) {
    const int64_t tile = ((int64_t)r * gridDim.x + blockIdx.x) * 4 + wave;
    const int64_t m = tile * 32 + (lane & 31);
    for (int l = 0; l < L; ++l) {
      const int64_t lo = (int64_t)l * Mp * H;
      f32x4 x0[AHEAD], x1[AHEAD], y0[AHEAD], y1[AHEAD];
#pragma unroll
      for (int i = 0; i < AHEAD; ++i) {
        int64_t o0, o1; offs<MODE>(lo, tile, m, i, lane, o0, o1);
        x0[i] = *reinterpret_cast<const f32x4*>(a + o0); x1[i] = *reinterpret_cast<const f32x4*>(a + o1);
        y0[i] = *reinterpret_cast<const f32x4*>(b + o0); y1[i] = *reinterpret_cast<const f32x4*>(b + o1);
      }
#pragma unroll
      for (int kc = 0; kc < KC; ++kc) {
        int64_t o0, o1; offs<MODE>(lo, tile, m, kc, lane, o0, o1);
        const f32x4 u0 = x0[kc % AHEAD], u1 = x1[kc % AHEAD], v0 = y0[kc % AHEAD], v1 = y1[kc % AHEAD];
        if (kc + AHEAD < KC) {
          int64_t q0, q1; offs<MODE>(lo, tile, m, kc + AHEAD, lane, q0, q1);
          x0[kc % AHEAD] = *reinterpret_cast<const f32x4*>(a + q0); x1[kc % AHEAD] = *reinterpret_cast<const f32x4*>(a + q1);
          y0[kc % AHEAD] = *reinterpret_cast<const f32x4*>(b + q0); y1[kc % AHEAD] = *reinterpret_cast<const f32x4*>(b + q1);
        }
        *reinterpret_cast<f32x4*>(c + o0) = u0 * v0; *reinterpret_cast<f32x4*>(c + o1) = u1 * v1;
        *reinterpret_cast<f32x4*>(d + o0) = u0 + v0; *reinterpret_cast<f32x4*>(d + o1) = u1 + v1;
        acc += u0.x + v1.w;
      }
    }
  }
  if (acc == 12345.678f) { c[0] = acc; lds[threadIdx.x] = acc; }
}

template <int MODE, int AHEAD>
void run(const char* name, float* a, float* b, float* c, float* d, int64_t Mp, size_t n, int lds_bytes) {
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  hipFuncSetAttribute((const void*)k<MODE, AHEAD>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024 - 512);
  const int grid = 256, rounds = (int)(Mp / 128 / grid);
  float best = 1e9f;
  for (int rep = 0; rep < 4; ++rep) {
    hipEventRecord(e0);
    k<MODE, AHEAD><<<grid, 256, lds_bytes>>>(a, b, c, d, Mp, rounds);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    if (rep && ms < best) best = ms;
  }
  printf("%-34s ahead %d  lds %3d KB: %.3f ms  %.2f TB/s\n", name, AHEAD, lds_bytes / 1024, best, 4.0 * n * 4 / best / 1e9);
}

int main() {
  const int64_t Mp = 98304;                       // 768 tiles of 128 points = 3 rounds on 256 CUs
  const size_t n = (size_t)L * Mp * H;
  float *a, *b, *c, *d;
  hipMalloc(&a, n * 4); hipMalloc(&b, n * 4); hipMalloc(&c, n * 4); hipMalloc(&d, n * 4);
  hipMemset(a, 0, n * 4); hipMemset(b, 0, n * 4);
  printf("2 tensors read + 2 written, %.2f GB per launch, 256 persistent workgroups x 4 waves\n", 4.0 * n * 4 / 1e9);
  for (int lds : {128 * 1024, 32 * 1024}) {        // 1 workgroup per CU (as the MLP kernels) / up to 4 per CU
    run<0, 1>("point-major [M][256]", a, b, c, d, Mp, n, lds);
    run<0, 3>("point-major [M][256]", a, b, c, d, Mp, n, lds);
    run<1, 3>("blocked [M/32][16][32][16]", a, b, c, d, Mp, n, lds);
    run<2, 1>("lane-native [M/32][16][2][64][4]", a, b, c, d, Mp, n, lds);
    run<2, 3>("lane-native [M/32][16][2][64][4]", a, b, c, d, Mp, n, lds);
  }
  return 0;
}
