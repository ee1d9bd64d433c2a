// Micro-benchmark (development tool, not part of the library): does the 256 MiB Infinity Cache hold what a kernel has just WRITTEN (or read)
// for the next kernel?  The backward of the training step is a producer -> consumer chain over saved tensors of 420 MB per point range
// (sweep 1 writes G(hbar), sweep 2 reads it; sweep 2 writes G(a), the weight gradients read it); if a consumer that follows its producer
// closely enough reads from the Infinity Cache instead of HBM, a depth-first schedule over SMALLER point ranges would pay.
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 scripts/ubench/mall_probe.hip -o /tmp/mall_probe && /tmp/mall_probe
// For every footprint X: time of a read pass of X bytes (a) right after a write pass of the same X bytes, (b) right after a read pass of the
// same bytes, (c) cold (a 2 GB sweep of other memory in between).  One 4-wave workgroup per CU-slot x 4 (1024 workgroups, grid-stride), 16 B per lane.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include <algorithm>
typedef float f32x4 __attribute__((ext_vector_type(4)));

template <int NT>
__global__ __launch_bounds__(256) void wr_kernel(float* __restrict__ dst, long n16, float v) {
  const long stride = (long)gridDim.x * 256;
  for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n16; i += stride) {
    f32x4 x = {v, v + 1.f, v + 2.f, (float)i};
    if (NT) __builtin_nontemporal_store(x, reinterpret_cast<f32x4*>(dst) + i);
    else reinterpret_cast<f32x4*>(dst)[i] = x;
  }
}
template <int NT>
__global__ __launch_bounds__(256) void rd_kernel(const float* __restrict__ src, long n16, float* sink) {
  const long stride = (long)gridDim.x * 256;
  f32x4 acc = {0.f, 0.f, 0.f, 0.f};
  for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n16; i += stride) {
    f32x4 x = NT ? __builtin_nontemporal_load(reinterpret_cast<const f32x4*>(src) + i) : reinterpret_cast<const f32x4*>(src)[i];
    acc += x;
  }
  if (acc[0] == 123.456f) sink[threadIdx.x] = acc[1];
}

#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); return 1; } } while (0)

int main() {
  const long MAXB = 2048L << 20;
  float *buf, *other, *sink;
  CK(hipMalloc(&buf, MAXB)); CK(hipMalloc(&other, MAXB)); CK(hipMalloc(&sink, 4096));
  CK(hipMemset(buf, 0, MAXB)); CK(hipMemset(other, 0, MAXB));
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  const int G = 1024;
  auto med = [](std::vector<float>& v) { std::sort(v.begin(), v.end()); return v[v.size() / 2]; };
  printf("footprint MB | read after WRITE (plain st)  | read after WRITE (nt st)     | read after READ              | read COLD                    | nt read after plain write\n");
  for (long mb : {16L, 32L, 64L, 96L, 128L, 160L, 192L, 256L, 384L, 512L, 1024L}) {
    const long bytes = mb << 20, n16 = bytes / 16;
    float res[5];
    for (int mode = 0; mode < 5; ++mode) {
      std::vector<float> ts;
      for (int rep = 0; rep < 7; ++rep) {
        // flush: sweep 2 GB of other memory
        rd_kernel<0><<<G, 256>>>(other, MAXB / 16, sink);
        if (mode == 0 || mode == 4) wr_kernel<0><<<G, 256>>>(buf, n16, (float)rep);
        else if (mode == 1) wr_kernel<1><<<G, 256>>>(buf, n16, (float)rep);
        else if (mode == 2) rd_kernel<0><<<G, 256>>>(buf, n16, sink);
        CK(hipEventRecord(e0));
        if (mode == 4) rd_kernel<1><<<G, 256>>>(buf, n16, sink); else rd_kernel<0><<<G, 256>>>(buf, n16, sink);
        CK(hipEventRecord(e1));
        CK(hipEventSynchronize(e1));
        float ms; CK(hipEventElapsedTime(&ms, e0, e1));
        ts.push_back(ms);
      }
      res[mode] = med(ts);
    }
    printf("%8ld     |", mb);
    for (int mode : {0, 1, 2, 3, 4}) printf(" %7.1f us = %6.2f TB/s      |", res[mode] * 1e3, bytes / (res[mode] * 1e-3) / 1e12);
    printf("\n");
  }
  // write pass after a read of the same bytes (does a store into a cached line cost less?) and plain write bandwidth by footprint
  printf("footprint MB | write COLD (plain) | write COLD (nt) | write after read (plain)\n");
  for (long mb : {32L, 128L, 512L, 1024L}) {
    const long bytes = mb << 20, n16 = bytes / 16;
    printf("%8ld     |", mb);
    for (int mode = 0; mode < 3; ++mode) {
      std::vector<float> ts;
      for (int rep = 0; rep < 7; ++rep) {
        rd_kernel<0><<<G, 256>>>(other, MAXB / 16, sink);
        if (mode == 2) rd_kernel<0><<<G, 256>>>(buf, n16, sink);
        CK(hipEventRecord(e0));
        if (mode == 1) wr_kernel<1><<<G, 256>>>(buf, n16, 1.f); else wr_kernel<0><<<G, 256>>>(buf, n16, 1.f);
        CK(hipEventRecord(e1));
        CK(hipEventSynchronize(e1));
        float ms; CK(hipEventElapsedTime(&ms, e0, e1));
        ts.push_back(ms);
      }
      float m = med(ts);
      printf(" %7.1f us = %5.2f TB/s |", m * 1e3, bytes / (m * 1e-3) / 1e12);
    }
    printf("\n");
  }
  return 0;
}
