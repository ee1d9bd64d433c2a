// Micro-benchmark (development tool, not part of the library): what do HBM loads and stores cost at the occupancy and access shape of the
// saved-tensor kernels?  Round 5's knock-outs priced a store pass of the backward sweeps at about four load passes
// (profiles/r5_step0_knockouts.txt); this isolates the memory system's side of that: R tensors read + W tensors written per "layer step",
// no arithmetic, 4-wave workgroups, one (LDS = 64 KB) or two per CU, 16 B per lane in the blocked layout's shape (lane (p, hi): 64 B lane
// stride, two instructions per 2 KB run) or lane-linear, store flavour plain / nt / sc1 / sc0 sc1.
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 scripts/ubench/hbm_mix.hip -o /tmp/hbm_mix && /tmp/hbm_mix
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef float f32x4 __attribute__((ext_vector_type(4)));

template <int FL>
__device__ __forceinline__ void st16(float* p, f32x4 v) {
  if (FL == 0) *reinterpret_cast<f32x4*>(p) = v;
  else if (FL == 1) __builtin_nontemporal_store(v, reinterpret_cast<f32x4*>(p));
  else if (FL == 2) asm volatile("global_store_dwordx4 %0, %1, off sc1" ::"v"(p), "v"(v) : "memory");
  else asm volatile("global_store_dwordx4 %0, %1, off sc0 sc1" ::"v"(p), "v"(v) : "memory");
}
template <int NT>
__device__ __forceinline__ f32x4 ld16(const float* p) {
  if (NT) return __builtin_nontemporal_load(reinterpret_cast<const f32x4*>(p));
  return *reinterpret_cast<const f32x4*>(p);
}

// A workgroup owns 128 points; per "layer" (8 of them) and k-chunk (16) it reads R tensors and writes W tensors, 2 x 16 B per lane and tensor.
// tensors are [layer][block of 32 points][k-chunk][32 points][16 floats] (the blocked layout); LINEAR: lane L takes bytes 16 L of each 1 KB piece
template <int R, int W, int FL, bool LINEAR, int LNT = 1, bool LLIN = LINEAR>
__global__ __launch_bounds__(256) void mix_kernel(const float* __restrict__ src, float* __restrict__ dst, long tensor_floats, long layer_floats, float* sink) {
  extern __shared__ float lds[];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const long blk = (long)blockIdx.x * 4 + wave;
  const int off = LINEAR ? lane * 4 : (lane & 31) * 16 + 4 * (lane >> 5);          // stores
  const int second = LINEAR ? 256 : 8;
  const int loff = LLIN ? lane * 4 : (lane & 31) * 16 + 4 * (lane >> 5);           // loads
  const int lsecond = LLIN ? 256 : 8;
  f32x4 acc = {0.f, 0.f, 0.f, 0.f};
  for (int l = 0; l < 8; ++l) {
    const long base = (long)l * layer_floats + blk * 8192;
#pragma unroll 4
    for (int kc = 0; kc < 16; ++kc) {
      f32x4 v[R > 0 ? R : 1][2];
#pragma unroll
      for (int t = 0; t < R; ++t) {
        const float* p = src + t * tensor_floats + base + kc * 512 + loff;
        v[t][0] = ld16<LNT>(p); v[t][1] = ld16<LNT>(p + lsecond);
      }
#pragma unroll
      for (int t = 0; t < R; ++t) acc += v[t][0] * 1.0001f + v[t][1];
#pragma unroll
      for (int t = 0; t < W; ++t) {
        float* p = dst + t * tensor_floats + base + kc * 512 + off;
        st16<FL>(p, acc + (float)t); st16<FL>(p + second, acc - (float)t);
      }
    }
  }
  if (acc[0] == 123.456f) sink[threadIdx.x] = acc[1];
}

static float* g_src; static float* g_dst; static float* g_sink;
static const long PTS = 51200, LAYER = PTS * 256, TENSOR = LAYER * 8;      // one point range of the training step: 420 MB per tensor

template <int R, int W, int FL, bool LINEAR, int LNT = 1, bool LLIN = LINEAR>
void run(const char* what, int lds_bytes) {
  auto k = mix_kernel<R, W, FL, LINEAR, LNT, LLIN>;
  hipFuncSetAttribute((const void*)k, hipFuncAttributeMaxDynamicSharedMemorySize, lds_bytes);
  const int grid = (int)(PTS / 128);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  k<<<grid, 256, lds_bytes>>>(g_src, g_dst, TENSOR, LAYER, g_sink);
  hipDeviceSynchronize();
  float best = 1e30f;
  for (int rep = 0; rep < 5; ++rep) {
    hipEventRecord(e0);
    k<<<grid, 256, lds_bytes>>>(g_src, g_dst, TENSOR, LAYER, g_sink);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    if (ms < best) best = ms;
  }
  const double gb = (double)(R + W) * TENSOR * 4 / 1e9;
  printf("%-26s R=%d W=%d  loads %-5s %-7s  stores %-6s %-7s LDS %3d KB: %7.1f us  %5.2f GB  %5.2f TB/s   (%6.1f us per pass)\n", what, R, W,
         LNT ? "nt" : "plain", LLIN ? "linear" : "blocked", FL == 0 ? "plain" : FL == 1 ? "nt" : FL == 2 ? "sc1" : "sc0sc1", LINEAR ? "linear" : "blocked", lds_bytes / 1024,
         best * 1e3, gb, gb / best, best * 1e3 / (R + W));
  fflush(stdout);
}

int main() {
  hipMalloc(&g_src, (size_t)4 * TENSOR * 4); hipMalloc(&g_dst, (size_t)3 * TENSOR * 4); hipMalloc(&g_sink, 4096);
  hipMemset(g_src, 0, (size_t)4 * TENSOR * 4); hipMemset(g_dst, 0, (size_t)3 * TENSOR * 4);
  const int L1 = 65536, L2 = 32768;      // one / two workgroups per CU (400 workgroups of one range: 1.56 rounds at one per CU)
  printf("== loads only\n");
  run<1, 0, 1, false>("reads", L1); run<2, 0, 1, false>("reads", L1); run<4, 0, 1, false>("reads", L1); run<4, 0, 1, false>("reads", L2);
  printf("== stores only\n");
  run<0, 1, 1, false>("writes", L1); run<0, 2, 1, false>("writes", L1); run<0, 2, 0, false>("writes", L1); run<0, 2, 2, false>("writes", L1);
  run<0, 2, 3, false>("writes", L1); run<0, 2, 1, true>("writes", L1); run<0, 2, 1, false>("writes", L2);
  printf("== the sweeps' mixes (round 4: sweep 1 = 2 R + 2 W, sweep 2 = 2 R + 1 W; round 5: 1 R + 1 W, 3 R + 1 W)\n");
  run<2, 2, 1, false>("round-4 sweep 1", L1); run<2, 1, 1, false>("round-4 sweep 2", L1);
  run<1, 1, 1, false>("round-5 sweep 1", L1); run<3, 1, 1, false>("round-5 sweep 2", L1);
  run<3, 1, 0, false>("round-5 sweep 2", L1); run<3, 1, 2, false>("round-5 sweep 2", L1); run<3, 1, 3, false>("round-5 sweep 2", L1);
  run<3, 1, 1, true>("round-5 sweep 2", L1); run<3, 1, 1, false>("round-5 sweep 2", L2);
  run<2, 2, 1, false>("round-4 sweep 1", L2); run<1, 1, 1, false>("round-5 sweep 1", L2);
  printf("== load / store flavour and shape, separately (64 KB LDS)\n");
  //          R  W  FL  SLIN  LNT  LLIN
  run<4, 0, 1, false, 0, false>("wgrad-like", L1); run<4, 0, 1, false, 1, true>("wgrad-like", L1); run<4, 0, 1, false, 0, true>("wgrad-like", L1);
  run<1, 1, 1, false, 0, false>("igrad / sweep 1", L1); run<1, 1, 0, false, 1, false>("igrad / sweep 1", L1); run<1, 1, 0, false, 0, false>("igrad / sweep 1", L1);
  run<1, 1, 1, false, 1, true>("igrad / sweep 1", L1); run<1, 1, 1, true, 1, false>("igrad / sweep 1", L1); run<1, 1, 0, true, 0, true>("igrad / sweep 1", L1);
  run<3, 1, 1, false, 0, false>("sweep 2", L1); run<3, 1, 0, false, 0, false>("sweep 2", L1); run<3, 1, 1, false, 1, true>("sweep 2", L1);
  run<3, 1, 1, true, 1, false>("sweep 2", L1); run<3, 1, 0, true, 0, true>("sweep 2", L1);
  run<2, 2, 0, false, 1, false>("round-4 sweep 1", L1); run<2, 2, 0, false, 0, false>("round-4 sweep 1", L1); run<2, 2, 1, false, 1, true>("round-4 sweep 1", L1);
  run<0, 1, 0, false>("forward with saves", L1); run<0, 1, 1, true>("forward with saves", L1);
  return 0;
}
