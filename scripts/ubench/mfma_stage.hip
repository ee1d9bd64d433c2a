// Micro-benchmark (development tool, not part of the library): what does the MI355X matrix pipe sustain for the instruction mix of a
// bf16x3 "stage" -- 96 x v_mfma_f32_32x32x16_bf16 per wave on 16 accumulator tiles (all 256 AGPRs), one wave per SIMD, 4 waves per
// workgroup, one workgroup per CU -- as the other work of the stage is added one item at a time?
//   bit 0: 24 ds_read_b128 per stage (15 up front, 9 spread)      bit 1: one s_barrier per stage
//   bit 2: 2 fp32 VALU per MFMA                                    bit 3: 8 global_load_dwordx4 per stage (consumed one stage later)
//   bit 4: 12 ds_write_b128 per stage                              bit 5: 4 fp32 VALU per MFMA (instead of 2)
//   bit 6: operands with random mantissas and signs (the switching activity of real data: what the power limit leaves of the clock)
// Build + run:  hipcc --offload-arch=gfx950 -O3 -std=c++17 -mllvm -pragma-unroll-threshold=1000000 scripts/ubench/mfma_stage.hip -o /tmp/mfma_stage && /tmp/mfma_stage
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

__device__ __forceinline__ f32x16 mfma(u32x4 a, u32x4 b, f32x16 c) {
  return __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, a), __builtin_bit_cast(bf16x8, b), c, 0, 0, 0);
}

template <int MODE>
__global__ __launch_bounds__(256) void stage_kernel(const float* __restrict__ src, float* __restrict__ out, int nst) {
  extern __shared__ __attribute__((aligned(16))) float lds[];
  const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
  f32x16 acc[4][4];
#pragma unroll
  for (int a = 0; a < 4; ++a)
#pragma unroll
    for (int b = 0; b < 4; ++b)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[a][b][r] = 0.f;
  auto rnd = [](unsigned x) { x ^= x >> 16; x *= 0x7feb352du; x ^= x >> 15; x *= 0x846ca68bu; x ^= x >> 16; return x; };
  // random sign + 7 mantissa bits per bf16, exponent of 1.0: values in +-[1, 2)
  auto rbf = [&](unsigned x) { const unsigned r = rnd(x); return (r & 0x807f807fu) | 0x3f803f80u; };
  for (int i = tid; i < 24576; i += 256) lds[i] = (MODE & 64) ? __uint_as_float(rbf(i * 7919u + blockIdx.x)) : 1.0f + (float)(i & 7);
  __syncthreads();
  u32x4 ap[4][3], bp[2][3];
#pragma unroll
  for (int p = 0; p < 3; ++p) {
#pragma unroll
    for (int t = 0; t < 4; ++t) {
      const unsigned k = (unsigned)(tid * 131 + t * 17 + p * 5);
      ap[t][p] = (MODE & 64) ? u32x4{rbf(k), rbf(k + 1000003u), rbf(k + 2000003u), rbf(k + 3000017u)}
                             : u32x4{0x3f803f80u + (unsigned)lane, 0x3f803f80u, 0x3f803f80u, 0x3f803f80u};
    }
    const unsigned kb = (unsigned)(tid * 257 + p * 11 + 77);
    bp[0][p] = (MODE & 64) ? u32x4{rbf(kb), rbf(kb + 5000011u), rbf(kb + 6000011u), rbf(kb + 7000003u)} : u32x4{0x3f803f80u, 0x3f803f80u + (unsigned)p, 0x3f803f80u, 0x3f803f80u};
    bp[1][p] = (MODE & 64) ? u32x4{rbf(kb + 31u), rbf(kb + 5100011u), rbf(kb + 6100011u), rbf(kb + 7100003u)} : bp[0][p];
  }
  auto plane = [&](int s, int q, int tile, int p) __attribute__((always_inline)) -> u32x4 {
    return *reinterpret_cast<const u32x4*>(lds + (s & 1) * 12288 + q * 3072 + (tile * 3 + p) * 256 + lane * 4);
  };
  f32x4 raw[8];
  const float* g = src + (size_t)blockIdx.x * 65536 + w * 16384 + lane * 4;
#pragma unroll
  for (int i = 0; i < 8; ++i) raw[i] = f32x4{1.f, 2.f, 3.f, 4.f};
  float va = 1.0f + lane, vb = 0.5f;
  for (int s = 0; s < nst; ++s) {
    if (MODE & 2) __syncthreads();
    if (MODE & 1) {
#pragma unroll
      for (int p = 0; p < 3; ++p) {
        bp[0][p] = plane(s, 2 + (w & 1), 0, p);
#pragma unroll
        for (int t = 0; t < 4; ++t) ap[t][p] = plane(s, w >> 1, t, p);
      }
    }
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int u = 0; u < 96; ++u) {
      const int tb = u / 24, gg = u % 24, q = gg / 4, ta = u % 4;
      const int sa = (q == 2 || q == 5) ? 1 : (q == 4 ? 2 : 0);
      const int sb = (q == 1 || q == 5) ? 1 : (q == 3 ? 2 : 0);
      acc[ta][tb] = mfma(ap[ta][sa], bp[tb & 1][sb], acc[ta][tb]);
      // raw[j] is refilled in unit 12 j + 5 and read in units 12 j .. 12 j + 4 of the NEXT stage: load-to-use distance ~ one stage, as in wgrad3p
      const float rv = (u % 12) < 5 ? raw[u / 12][u & 3] : 0.75f;
      if (MODE & 4) { va = fmaf(va, vb, rv); vb = fmaf(vb, va, 1.0f); }
      if (MODE & 32) { va = fmaf(va, vb, rv); vb = fmaf(vb, va, 1.0f); va = fmaf(va, vb, 0.25f); vb = fmaf(vb, va, 2.0f); }
      if ((MODE & 8) && (u % 12) == 5) raw[u / 12] = *reinterpret_cast<const f32x4*>(g + (size_t)(s & 15) * 1024 + (u / 12) * 256);
      if ((MODE & 16) && (u % 8) == 3)
        *reinterpret_cast<u32x4*>(lds + ((s + 1) & 1) * 12288 + w * 3072 + (u / 8) * 256 + lane * 4) = u32x4{__float_as_uint(va), 0x3f803f80u, 0x3f803f80u, 0x3f803f80u};
      if ((MODE & 1) && tb < 3 && gg >= 12 && gg < 15) bp[(tb + 1) & 1][gg - 12] = plane(s, 2 + (w & 1), tb + 1, gg - 12);
      __builtin_amdgcn_sched_barrier(0);
    }
  }
  float sum = va + vb;
#pragma unroll
  for (int a = 0; a < 4; ++a)
#pragma unroll
    for (int b = 0; b < 4; ++b)
#pragma unroll
      for (int r = 0; r < 16; ++r) sum += acc[a][b][r];
#pragma unroll
  for (int i = 0; i < 8; ++i) sum += raw[i][0];
  if (sum == 123.456f) out[blockIdx.x * 256 + tid] = sum;
}

template <int MODE>
void run(const char* what, const float* src, float* out, int nst, int wgs) {
  hipFuncSetAttribute(reinterpret_cast<const void*>(stage_kernel<MODE>), hipFuncAttributeMaxDynamicSharedMemorySize, 98304);
  hipEvent_t e0, e1;
  hipEventCreate(&e0); hipEventCreate(&e1);
  stage_kernel<MODE><<<wgs, 256, 98304>>>(src, out, nst);
  hipDeviceSynchronize();
  float best = 1e30f;
  for (int rep = 0; rep < 5; ++rep) {
    hipEventRecord(e0);
    stage_kernel<MODE><<<wgs, 256, 98304>>>(src, out, nst);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms;
    hipEventElapsedTime(&ms, e0, e1);
    if (ms < best) best = ms;
  }
  const double mfmas = (double)wgs * 4 * nst * 96;
  const double tf = mfmas * 32768.0 / (best * 1e-3) / 1e12;
  // cycles per MFMA and SIMD at the wall clock, if the chip ran at 2.4 GHz: 1024 SIMDs
  const double cyc = best * 1e-3 * 2.4e9 / ((double)wgs / 256 * nst * 96);
  printf("mode %2d  %-62s %8.3f ms  %7.1f TFLOP/s bf16 (%.3f of 2500)  %.1f cycles@2.4GHz per MFMA\n", MODE, what, best, tf, tf / 2500.0, cyc);
  fflush(stdout);
}

int main() {
  float *src, *out;
  const int wgs = 256 * 4, nst = 512;
  hipMalloc(&src, (size_t)(wgs + 1) * 65536 * 4);
  hipMemset(src, 0, (size_t)(wgs + 1) * 65536 * 4);
  hipMalloc(&out, (size_t)wgs * 256 * 4);
  run<0>("MFMA only (operands in registers)", src, out, nst, wgs);
  run<1>("+ 24 ds_read_b128 / stage", src, out, nst, wgs);
  run<3>("+ 24 ds_read_b128 + barrier", src, out, nst, wgs);
  run<2>("barrier only", src, out, nst, wgs);
  run<4>("2 VALU / MFMA", src, out, nst, wgs);
  run<32>("4 VALU / MFMA", src, out, nst, wgs);
  run<36>("6 VALU / MFMA", src, out, nst, wgs);
  run<8>("8 global_load_dwordx4 / stage", src, out, nst, wgs);
  run<16>("12 ds_write_b128 / stage", src, out, nst, wgs);
  run<12>("2 VALU + global loads", src, out, nst, wgs);
  run<14>("2 VALU + global loads + barrier", src, out, nst, wgs);
  run<7>("reads + barrier + 2 VALU", src, out, nst, wgs);
  run<23>("reads + barrier + 2 VALU + writes", src, out, nst, wgs);
  run<31>("reads + barrier + 2 VALU + writes + global loads (the real stage)", src, out, nst, wgs);
  run<59>("same with 4 VALU / MFMA", src, out, nst, wgs);
  run<64>("MFMA only, random operand bits", src, out, nst, wgs);
  run<65>("random operand bits + 24 ds_read_b128 / stage", src, out, nst, wgs);
  run<67>("random operand bits + reads + barrier", src, out, nst, wgs);
  run<95>("random operand bits, the real stage (2 VALU / MFMA)", src, out, nst, wgs);
  run<123>("random operand bits, the real stage (4 VALU / MFMA)", src, out, nst, wgs);
  return 0;
}
