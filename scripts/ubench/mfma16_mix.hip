// Micro-benchmark (development tool, not part of the library): what does a SIMD sustain on v_mfma_f32_16x16x32_bf16 when one or two waves
// share it and each MFMA is accompanied by the other instructions of a 16-point-wave bf16x3 group (x3h.h)?
//   template <NW waves per workgroup (4 = one wave per SIMD, 8 = two), MODE>
//   bit 0: 1 fp32 VALU per MFMA, each independent of its neighbours (reads two registers that no recent instruction wrote)
//   bit 1: the VALU ops form ONE dependent chain instead (every op reads the previous result)
//   bit 2: 1 transcendental (v_exp_f32) per 6 MFMAs on top
//   bit 3: 1 ds_read_b128 per 2 MFMAs (A operands from LDS, two groups ahead)
//   bit 4: 2 VALU per MFMA instead of 1
//   bit 7: the VALU ops read registers that MFMAs wrote in an earlier phase (the previous layer's accumulators, as the B preparation does)
//   bit 8: the VALU results are written into the B-operand registers of the NEXT stage's MFMAs (double buffered, as the split planes are)
//   bit 6: accumulator reuse distance 2 (tile pair, two products each) instead of 16
//   bit 5: v_cvt_pk_bf16_f32 + v_lshlrev + v_sub (the split's instruction kinds) instead of v_fma
// operands are random bf16 (the switching activity of real data).  One fenced unit per MFMA, as in dense_x3h.
// Build + run:  hipcc --offload-arch=gfx950 -O3 -std=c++17 -mllvm -pragma-unroll-threshold=1000000 -fno-slp-vectorize scripts/ubench/mfma16_mix.hip -o /tmp/mfma16_mix && /tmp/mfma16_mix
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

__device__ __forceinline__ f32x4 mfma(u32x4 a, u32x4 b, f32x4 c) {
  return __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, a), __builtin_bit_cast(bf16x8, b), c, 0, 0, 0);
}
__device__ __forceinline__ unsigned pk_bf16(float a, float b) {
  const f32x2 v = {a, b};
  return __builtin_bit_cast(unsigned, __builtin_convertvector(v, bf16x2));
}

template <int NW, int MODE>
__global__ __launch_bounds__(NW * 64, 2) void mix_kernel(float* __restrict__ out, int nst) {
  extern __shared__ __attribute__((aligned(16))) float lds[];
  const int tid = threadIdx.x, lane = tid & 63;
  auto rnd = [](unsigned x) { x ^= x >> 16; x *= 0x7feb352du; x ^= x >> 15; x *= 0x846ca68bu; x ^= x >> 16; return x; };
  auto rbf = [&](unsigned x) { const unsigned r = rnd(x); return (r & 0x807f807fu) | 0x3f803f80u; };
  for (int i = tid; i < 16384; i += NW * 64) lds[i] = __uint_as_float(rbf(i * 7919u + blockIdx.x));
  __syncthreads();
  f32x4 acc[16];
#pragma unroll
  for (int t = 0; t < 16; ++t) acc[t] = f32x4{0.f, 0.f, 0.f, 0.f};
  u32x4 a[4], b[3], bn[3];
  f32x4 accP[16];
#pragma unroll
  for (int i = 0; i < 4; ++i) { const unsigned k = tid * 131u + i * 17u; a[i] = u32x4{rbf(k), rbf(k + 1000003u), rbf(k + 2000003u), rbf(k + 3000017u)}; }
#pragma unroll
  for (int i = 0; i < 3; ++i) { const unsigned k = tid * 257u + i * 11u + 77u; b[i] = u32x4{rbf(k), rbf(k + 5000011u), rbf(k + 6000011u), rbf(k + 7000003u)}; }
  if (MODE & 128) {      // "previous layer": accumulators written by MFMAs once, read by VALU ops from then on
#pragma unroll
    for (int t = 0; t < 16; ++t) accP[t] = mfma(a[t & 3], b[t % 3], f32x4{0.f, 0.f, 0.f, 0.f});
  }
#pragma unroll
  for (int i = 0; i < 3; ++i) bn[i] = b[i];
  float v[16];
#pragma unroll
  for (int i = 0; i < 16; ++i) v[i] = 1.0f + 0.001f * (float)(lane + i);
  float chain = 1.0f + lane * 1e-3f;
  const u32x4* base = reinterpret_cast<const u32x4*>(lds) + lane;
  for (int s = 0; s < nst; ++s) {
#pragma unroll
    for (int u = 0; u < 96; ++u) {
      const int t = (MODE & 64) ? 2 * ((u >> 2) & 7) + (u & 1) : (u & 15);      // bit 6: a tile is accumulated into again two MFMAs later (the groups of dense_x3h)
      acc[t] = mfma(a[u & 3], b[u % 3], acc[t]);
      if (MODE & 8) { if ((u & 1) == 0) a[(u / 2 + 2) & 3] = base[((s & 3) * 48 + u / 2) * 64]; }
      constexpr int NV = (MODE & 16) ? 2 : ((MODE & 3) ? 1 : 0);
#pragma unroll
      for (int q = 0; q < NV; ++q) {
        const int i = (u * 2 + q) & 15;
        if (MODE & 2) {
          chain = fmaf(chain, 0.999f, 0.001f);
        } else if (MODE & 32) {
          const int w = (u * 2 + q) % 3;
          if (w == 0) { const unsigned p = pk_bf16(v[i], v[(i + 1) & 15]); v[(i + 8) & 15] = __uint_as_float(p << 16); }
          else if (w == 1) v[(i + 8) & 15] = v[i] - v[(i + 3) & 15];
          else v[(i + 8) & 15] = __uint_as_float(__float_as_uint(v[i]) & 0xffff0000u);
        } else {
          const float src = (MODE & 128) ? accP[(u >> 2) & 15][u & 3] : v[i];
          const float r = fmaf(src, 0.999f, v[(i + 1) & 15]);      // reads registers written >= 7 VALU ops ago
          if (MODE & 256) bn[u % 3][(u / 3) & 3] = (__float_as_uint(r) & 0x007f007fu) | 0x3f803f80u;
          else v[(i + 8) & 15] = r;
        }
      }
      if ((MODE & 4) && (u % 6) == 0) v[(u / 6) & 15] = __builtin_amdgcn_exp2f(v[(u / 6 + 5) & 15] * -0.01f);
      __builtin_amdgcn_sched_barrier(0);
    }
    if (MODE & 256) {
#pragma unroll
      for (int i = 0; i < 3; ++i) { const u32x4 t = b[i]; b[i] = bn[i]; bn[i] = t; }
    }
  }
  float sum = chain;
#pragma unroll
  for (int t = 0; t < 16; ++t) sum += acc[t][0] + acc[t][1] + acc[t][2] + acc[t][3];
#pragma unroll
  for (int i = 0; i < 16; ++i) sum += v[i];
  if (sum == 123.456f) out[blockIdx.x * NW * 64 + tid] = sum;
}

template <int NW, int MODE>
void run(const char* what, float* out, int nst, int wgs) {
  hipFuncSetAttribute(reinterpret_cast<const void*>(mix_kernel<NW, MODE>), hipFuncAttributeMaxDynamicSharedMemorySize, 65536);
  hipEvent_t e0, e1;
  hipEventCreate(&e0); hipEventCreate(&e1);
  mix_kernel<NW, MODE><<<wgs, NW * 64, 65536>>>(out, nst);
  hipDeviceSynchronize();
  float best = 1e30f;
  for (int rep = 0; rep < 5; ++rep) {
    hipEventRecord(e0);
    mix_kernel<NW, MODE><<<wgs, NW * 64, 65536>>>(out, nst);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms;
    hipEventElapsedTime(&ms, e0, e1);
    if (ms < best) best = ms;
  }
  const double mfmas = (double)wgs * NW * nst * 96;
  const double tf = mfmas * 16384.0 / (best * 1e-3) / 1e12;
  printf("waves/SIMD %d mode %2d  %-58s %8.3f ms  %7.1f TFLOP/s bf16 (%.3f of 2500)\n", NW / 4, MODE, what, best, tf, tf / 2500.0);
  fflush(stdout);
}

template <int NW>
void all(float* out) {
  const int wgs = 256 * 4, nst = NW == 8 ? 512 : 1024;
  run<NW, 0>("MFMA only", out, nst, wgs);
  run<NW, 1>("+ 1 independent VALU / MFMA", out, nst, wgs);
  run<NW, 2>("+ 1 VALU / MFMA, one dependent chain", out, nst, wgs);
  run<NW, 17>("+ 2 independent VALU / MFMA", out, nst, wgs);
  run<NW, 33>("+ 1 VALU / MFMA of the split's kinds (cvt_pk, shift, sub, and)", out, nst, wgs);
  run<NW, 5>("+ 1 VALU / MFMA + 1 v_exp / 6 MFMA", out, nst, wgs);
  run<NW, 8>("+ 1 ds_read_b128 / 2 MFMA", out, nst, wgs);
  run<NW, 9>("+ 1 VALU + reads", out, nst, wgs);
  run<NW, 13>("+ 1 VALU + v_exp + reads", out, nst, wgs);
  run<NW, 45>("+ split-kind VALU + v_exp + reads", out, nst, wgs);
  run<NW, 64>("MFMA only, accumulator reuse distance 2", out, nst, wgs);
  run<NW, 129>("1 VALU / MFMA reading old MFMA results", out, nst, wgs);
  run<NW, 257>("1 VALU / MFMA writing the next stage's B operands", out, nst, wgs);
  run<NW, 385>("1 VALU / MFMA, both", out, nst, wgs);
  run<NW, 461>("both + v_exp + reads + reuse distance 2", out, nst, wgs);
  run<NW, 109>("split-kind VALU + v_exp + reads, reuse distance 2", out, nst, wgs);
}

int main() {
  float* out;
  hipMalloc(&out, (size_t)1024 * 512 * 4);
  all<8>(out);
  all<4>(out);
  return 0;
}
