// Micro-benchmark (development tool, not part of the library): do the saved-tensor loads and stores of a 32-point-wave kernel overlap with
// its MFMA stream?  scripts/ubench/hbm_mix.hip measured the pure access patterns (sweep 1's 1 R + 1 W: 177 us per point range); the kernels
// themselves take MFMA time + about that (sweep 1: 330 us at 0.40 busy = 132 us of MFMA).  This model kernel approaches the real one from
// the ideal side: one 4-wave workgroup per CU (96 KB of LDS), per wave and k-chunk 12 fenced groups of 4 MFMAs (32x32x16 bf16, 8 accumulator
// tiles: the arithmetic of one bf16x3 k-chunk); the loads of k-chunk kc + AHEAD (R tensors, 2 x 16 B per lane in the blocked layout's shape)
// go behind group LG, the stores of k-chunk kc (W tensors; values made from the loaded ones, as in the sweeps) behind group SG.
// Part 2: latency probes -- cycles from issuing 2 loads / 2 stores / both to s_waitcnt vmcnt(0), one wave per SIMD, the whole chip active.
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -mllvm -pragma-unroll-threshold=1000000 scripts/ubench/mfma_paced.hip -o /tmp/mfma_paced && /tmp/mfma_paced
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

__device__ __forceinline__ f32x16 mfma_bf16(u32x4 a, u32x4 b, f32x16 c) {
  return __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, a), __builtin_bit_cast(bf16x8, b), c, 0, 0, 0);
}
template <int FL>
__device__ __forceinline__ void st16(float* p, f32x4 v) {
  if (FL == 0) *reinterpret_cast<f32x4*>(p) = v;
  else __builtin_nontemporal_store(v, reinterpret_cast<f32x4*>(p));
}
template <int NT>
__device__ __forceinline__ f32x4 ld16(const float* p) {
  if (NT) return __builtin_nontemporal_load(reinterpret_cast<const f32x4*>(p));
  return *reinterpret_cast<const f32x4*>(p);
}

enum { F_FIXED = 1, F_PRIO = 2, F_LDFIXED = 4, F_NOMFMA = 8 };

// R tensors read, W written, FL store flavour (0 plain, 1 nt), AHEAD k-chunks of load-ahead, LG / SG: the MFMA group (0..11) behind which the
// loads / stores are issued, FEAT feature bits
template <int R, int W, int FL, int AHEAD, int LG, int SG, int FEAT>
__global__ __launch_bounds__(256) void paced_kernel(const float* __restrict__ src, float* __restrict__ dst, long tensor_floats, long layer_floats, float* sink) {
  constexpr int RING = AHEAD + 1;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const long blk = (long)blockIdx.x * 4 + wave;
  const int off = (lane & 31) * 16 + 4 * (lane >> 5);
  f32x16 acc[8];
#pragma unroll
  for (int i = 0; i < 8; ++i)
#pragma unroll
    for (int j = 0; j < 16; ++j) acc[i][j] = 0.f;
  const u32x4 b = {0x3f803f80u + lane, 0x3f803f80u, 0x3f003f00u, 0x3e803e80u};
  f32x4 q[RING][R > 0 ? R : 1][2];
  const float* sbase = src + blk * 8192 + off;
  float* dbase = dst + blk * 8192 + off;
  // k-chunk kk of the walk (0 .. 127): layer kk / 16, chunk kk % 16
  auto loads = [&](int slot, const float* lbase, int kc) {
#pragma unroll
    for (int t = 0; t < R; ++t) {
      const float* p = (FEAT & F_LDFIXED) ? sbase + t * tensor_floats : lbase + t * tensor_floats + kc * 512;
      q[slot][t][0] = ld16<0>(p);
      q[slot][t][1] = ld16<0>(p + 8);
    }
  };
  if (R) {
#pragma unroll
    for (int k0 = 0; k0 < AHEAD; ++k0) loads(k0, sbase, k0);
  }
  for (int l = 0; l < 8; ++l) {
    const float* lb = sbase + (long)l * layer_floats;
    const float* lbn = sbase + (long)(l < 7 ? l + 1 : l) * layer_floats;
    float* db = dbase + (long)l * layer_floats;
#pragma unroll
    for (int kc = 0; kc < 16; ++kc) {
      u32x4 a = {0x3f803f80u, 0x3f803f80u ^ (unsigned)kc, 0x3f003f00u, 0x3e803e80u};
      f32x4 v0 = {1.f, 2.f, 3.f, 4.f}, v1 = {5.f, 6.f, 7.f, 8.f};
#pragma unroll
      for (int g = 0; g < 12; ++g) {
        if (R && g == LG) {
          if (kc + AHEAD < 16) loads((kc + AHEAD) % RING, lb, kc + AHEAD);
          else loads((kc + AHEAD) % RING, lbn, kc + AHEAD - 16);
        }
        if (g == 1) {      // consume the loads of this k-chunk: every component, they become the A operand and the stored values
#pragma unroll
          for (int t = 0; t < R; ++t) {
            v0 += q[kc % RING][t][0] * 1.5f; v1 += q[kc % RING][t][1] * 0.5f;
          }
          a.x ^= __builtin_bit_cast(unsigned, v0.x + v0.y + v0.z + v0.w) & 0xffu;
          a.y ^= __builtin_bit_cast(unsigned, v1.x + v1.y + v1.z + v1.w) & 0xffu;
        }
        if (!(FEAT & F_NOMFMA)) {
#pragma unroll
          for (int m = 0; m < 4; ++m) acc[(4 * g + m) % 8] = mfma_bf16(a, b, acc[(4 * g + m) % 8]);
        }
        if (W && g == SG) {
          if (FEAT & F_PRIO) __builtin_amdgcn_s_setprio(3);
#pragma unroll
          for (int t = 0; t < W; ++t) {
            float* p = (FEAT & F_FIXED) ? dbase + t * tensor_floats : db + t * tensor_floats + kc * 512;
            st16<FL>(p, v0 + (float)t); st16<FL>(p + 8, v1 - (float)t);
          }
          if (FEAT & F_PRIO) __builtin_amdgcn_s_setprio(0);
        }
        __builtin_amdgcn_sched_barrier(0);
      }
    }
  }
  float r = 0.f;
#pragma unroll
  for (int i = 0; i < 8; ++i) r += acc[i][0] + acc[i][15];
  if (r == 123.456f) sink[threadIdx.x] = r;
}

// ---- the same walk with the weight stream of the real kernels: stages of 16 groups (32 KB of weights by LDS DMA, 8 pieces of 4 KB, two per group
// in the first four groups of the previous stage; two ds_read_b128 per group and wave, two groups ahead), a barrier per stage.
//   WAIT  0: s_waitcnt vmcnt(0) in front of every stage barrier (the real kernels)   1: counted -- only what is OLDER than the stage's last DMA piece
//   SPOS  where the stores of a k-chunk go: 0 = right after the next stage barrier (the real kernels' stash/flush), 1 = behind the DMA pieces of the
//         next stage (group 4), 2 = where the values exist (group 1 of the k-chunk)
template <int N> __device__ __forceinline__ void wait_vm() {
  __builtin_amdgcn_s_waitcnt((N & 15) | ((N >> 4) << 14) | 0x0F70);
}
__device__ __forceinline__ void wait_vm_n(int n) {      // n is a compile-time value after unrolling
  switch (n < 0 ? 0 : n > 24 ? 24 : n) {
    case 0: wait_vm<0>(); break; case 1: wait_vm<1>(); break; case 2: wait_vm<2>(); break; case 3: wait_vm<3>(); break;
    case 4: wait_vm<4>(); break; case 5: wait_vm<5>(); break; case 6: wait_vm<6>(); break; case 7: wait_vm<7>(); break;
    case 8: wait_vm<8>(); break; case 9: wait_vm<9>(); break; case 10: wait_vm<10>(); break; case 11: wait_vm<11>(); break;
    case 12: wait_vm<12>(); break; case 13: wait_vm<13>(); break; case 14: wait_vm<14>(); break; case 15: wait_vm<15>(); break;
    case 16: wait_vm<16>(); break; case 17: wait_vm<17>(); break; case 18: wait_vm<18>(); break; case 19: wait_vm<19>(); break;
    case 20: wait_vm<20>(); break; case 21: wait_vm<21>(); break; case 22: wait_vm<22>(); break; case 23: wait_vm<23>(); break;
    default: wait_vm<24>(); break;
  }
}
template <int R, int W, int FL, int WAIT, int SPOS, int SGR>      // SGR: MFMA groups per stage (16: 32 KB stages, as built; 32: 64 KB stages)
__global__ __launch_bounds__(256) void staged_kernel(const float* __restrict__ src, float* __restrict__ dst, const float* __restrict__ wts, long tensor_floats,
                                                     long layer_floats, float* sink) {
  extern __shared__ float lds[];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const long blk = (long)blockIdx.x * 4 + wave;
  const int off = (lane & 31) * 16 + 4 * (lane >> 5);
  f32x16 acc[8];
#pragma unroll
  for (int i = 0; i < 8; ++i)
#pragma unroll
    for (int j = 0; j < 16; ++j) acc[i][j] = 0.f;
  f32x4 q[2][R > 0 ? R : 1][2];
  const float* sbase = src + blk * 8192 + off;
  float* dbase = dst + blk * 8192 + off;
  __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(wts), 0, 0x7ffffff0, 0x00020000);
  const int wv = __builtin_amdgcn_readfirstlane(wave);
  auto dma = [&](int buf, int stage, int piece) {      // stage 0..11 of the layer's 384 KB
    float* d = lds + buf * (SGR * 512) + piece * 1024 + wv * 256;
    __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, (__attribute__((address_space(3))) void*)d, 16, tid * 16, (stage * (SGR * 2048)) % 393216 + piece * 4096, 0, 0);
  };
  auto loads = [&](int slot, const float* lbase, int kc) {
#pragma unroll
    for (int t = 0; t < R; ++t) {
      const float* p = lbase + t * tensor_floats + kc * 512;
      q[slot][t][0] = ld16<0>(p);
      q[slot][t][1] = ld16<0>(p + 8);
    }
  };
  if (R) loads(0, sbase, 0);
#pragma unroll
  for (int pc = 0; pc < SGR / 2; ++pc) dma(0, 0, pc);
  f32x4 pv[4][2];      // stores waiting for their slot
  const u32x4 bconst = {0x3f803f80u + lane, 0x3f803f80u, 0x3f003f00u, 0x3e803e80u};
  for (int l = 0; l < 8; ++l) {
    const float* lb = sbase + (long)l * layer_floats;
    const float* lbn = sbase + (long)(l < 7 ? l + 1 : l) * layer_floats;
    float* db = dbase + (long)l * layer_floats;
    int vm_young = 0;      // VMEM instructions issued after the last DMA piece of the stage that the next barrier hands over
    int pend0 = -1, pend1 = -1, pend2 = -1, pend3 = -1;
    u32x4 b = bconst;
    f32x4 v0 = {1.f, 2.f, 3.f, 4.f}, v1 = {5.f, 6.f, 7.f, 8.f};
    auto store_kc = [&](int kc, const f32x4 (&pvv)[2]) __attribute__((always_inline)) {
#pragma unroll
      for (int t = 0; t < W; ++t) {
        float* p = db + t * tensor_floats + kc * 512;
        st16<FL>(p, pvv[0] + (float)t); st16<FL>(p + 8, pvv[1] - (float)t);
      }
      vm_young += 2 * W;
    };
    auto flush = [&]() __attribute__((always_inline)) {
      if (pend0 >= 0) { store_kc(pend0, pv[0]); pend0 = -1; }
      if (pend1 >= 0) { const f32x4 t2[2] = {pv[1][0], pv[1][1]}; store_kc(pend1, t2); pend1 = -1; }
      if (pend2 >= 0) { const f32x4 t2[2] = {pv[2][0], pv[2][1]}; store_kc(pend2, t2); pend2 = -1; }
      if (pend3 >= 0) { const f32x4 t2[2] = {pv[3][0], pv[3][1]}; store_kc(pend3, t2); pend3 = -1; }
    };
#pragma unroll
    for (int gg = 0; gg < 192; ++gg) {
      const int st = gg / SGR, gs = gg % SGR, kc = gg / 12, g = gg % 12;
      const u32x4* cur = reinterpret_cast<const u32x4*>(lds + (st & 1) * (SGR * 512)) + lane;
      u32x4 ring[2][2];
      if (gs == 0) {
        if (WAIT == 0 || st == 0) wait_vm<0>(); else wait_vm_n(vm_young);
        __builtin_amdgcn_s_barrier();
        vm_young = 0;
        if (SPOS == 0) flush();
        ring[0][0] = cur[0]; ring[0][1] = cur[64]; ring[1][0] = cur[128]; ring[1][1] = cur[192];
      }
      if (gs < SGR / 4) {      // next stage's pieces (the last stage of a layer fetches the first one of the next layer: same buffer here)
        dma((st + 1) & 1, st + 1, 2 * gs); dma((st + 1) & 1, st + 1, 2 * gs + 1);
        if (gs == SGR / 4 - 1) vm_young = 0;
      }
      if (gs == SGR / 4 && SPOS == 1) flush();
      if (R && g == 0) {
        if (kc + 1 < 16) loads((kc + 1) & 1, lb, kc + 1); else loads(0, lbn, 0);
        vm_young += 2 * R;
      }
      if (g == 1) {
        v0 = f32x4{1.f, 2.f, 3.f, 4.f}; v1 = f32x4{5.f, 6.f, 7.f, 8.f};
#pragma unroll
        for (int t = 0; t < R; ++t) { v0 += q[kc & 1][t][0] * 1.5f; v1 += q[kc & 1][t][1] * 0.5f; }
        b = bconst;
        b.x ^= __builtin_bit_cast(unsigned, v0.x + v0.y + v0.z + v0.w) & 0xffu;
        b.y ^= __builtin_bit_cast(unsigned, v1.x + v1.y + v1.z + v1.w) & 0xffu;
        if (W) {
          if (SPOS == 2) { const f32x4 t2[2] = {v0, v1}; store_kc(kc, t2); }
          else if (pend0 < 0) { pv[0][0] = v0; pv[0][1] = v1; pend0 = kc; }
          else if (pend1 < 0) { pv[1][0] = v0; pv[1][1] = v1; pend1 = kc; }
          else if (pend2 < 0) { pv[2][0] = v0; pv[2][1] = v1; pend2 = kc; }
          else { pv[3][0] = v0; pv[3][1] = v1; pend3 = kc; }
        }
      }
      const u32x4 a0 = ring[gs & 1][0], a1 = ring[gs & 1][1];
      if (gs + 2 < SGR) { ring[gs & 1][0] = cur[(2 * (gs + 2)) * 64]; ring[gs & 1][1] = cur[(2 * (gs + 2) + 1) * 64]; }
      acc[(4 * g) % 8] = mfma_bf16(a0, b, acc[(4 * g) % 8]);
      acc[(4 * g + 1) % 8] = mfma_bf16(a1, b, acc[(4 * g + 1) % 8]);
      acc[(4 * g + 2) % 8] = mfma_bf16(a0, b, acc[(4 * g + 2) % 8]);
      acc[(4 * g + 3) % 8] = mfma_bf16(a1, b, acc[(4 * g + 3) % 8]);
      __builtin_amdgcn_sched_barrier(0);
    }
    flush();
  }
  float r = 0.f;
#pragma unroll
  for (int i = 0; i < 8; ++i) r += acc[i][0] + acc[i][15];
  if (r == 123.456f) sink[threadIdx.x] = r;
}

// ---- Round 6: the model above + the VALU work of the real sweeps, piece by piece (VERDICT r5 task 1: "instrument, do not guess").
// staged_kernel<R, W, nt, counted wait, stores behind the DMA pieces> is the best model of round 5 (240 / 350 us for the sweeps' shapes);
// the real kernels took 314 / 479 us.  Here the B operand of k-chunk kc + 1 is PRODUCED during the MFMAs of k-chunk kc the way x3.h:
// dense_x3g does it -- twelve pieces, one per MFMA group: eight values (u = 0..7), then four split items (two values -> three bf16
// planes) -- and the stored values are those eight values.  VM selects how much of a value's arithmetic is there:
//   1: the three-way split only (value = loaded h * accumulator)           2: + sigma = 1 - exp2(c h)          (sweep 1's value())
//   3: + G2: v_rcp, compare / select, three multiplies, fma               (sweep 2's value(); needs R = 3)
//   DEP: the value also reads an accumulator register of the PREVIOUS layer's tiles (a second accumulator set is live: 128 more registers)
typedef __bf16 bf16x2_t __attribute__((ext_vector_type(2)));
typedef float f32x2_t __attribute__((ext_vector_type(2)));
__device__ __forceinline__ unsigned pk_bf16_(float a, float b) {
  const f32x2_t v = {a, b};
  return __builtin_bit_cast(unsigned, __builtin_convertvector(v, bf16x2_t));
}
__device__ __forceinline__ float bf_lo(unsigned p) { return __builtin_bit_cast(float, p << 16); }
__device__ __forceinline__ float bf_hi(unsigned p) { return __builtin_bit_cast(float, p & 0xffff0000u); }
template <int R, int W, int VM, bool DEP>
__global__ __launch_bounds__(256) void valu_kernel(const float* __restrict__ src, float* __restrict__ dst, const float* __restrict__ wts, long tensor_floats,
                                                   long layer_floats, float* sink) {
  constexpr int SGR = 16;
  extern __shared__ float lds[];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const long blk = (long)blockIdx.x * 4 + wave;
  const int off = (lane & 31) * 16 + 4 * (lane >> 5);
  f32x16 acc[8], accP[8];
#pragma unroll
  for (int i = 0; i < 8; ++i)
#pragma unroll
    for (int j = 0; j < 16; ++j) { acc[i][j] = 0.f; accP[i][j] = 1.0f + 0.001f * (float)(i + j + lane); }
  f32x4 q[2][R > 0 ? R : 1][2];
  const float* sbase = src + blk * 8192 + off;
  float* dbase = dst + blk * 8192 + off;
  __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(wts), 0, 0x7ffffff0, 0x00020000);
  const int wv = __builtin_amdgcn_readfirstlane(wave);
  auto dma = [&](int buf, int stage, int piece) {
    float* d = lds + buf * (SGR * 512) + piece * 1024 + wv * 256;
    __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, (__attribute__((address_space(3))) void*)d, 16, tid * 16, (stage * (SGR * 2048)) % 393216 + piece * 4096, 0, 0);
  };
  auto loads = [&](int slot, const float* lbase, int kc) {
#pragma unroll
    for (int t = 0; t < R; ++t) {
      const float* p = lbase + t * tensor_floats + kc * 512;
      q[slot][t][0] = ld16<0>(p);
      q[slot][t][1] = ld16<0>(p + 8);
    }
  };
  if (R) loads(0, sbase, 0);
#pragma unroll
  for (int pc = 0; pc < SGR / 2; ++pc) dma(0, 0, pc);
  float sv[2][8];      // stash of the values whose stores wait for their slot
  u32x4 bq[2][3];
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int p = 0; p < 3; ++p) bq[i][p] = u32x4{0x3f803f80u + lane, 0x3f803f80u, 0x3f003f00u, 0x3e803e80u};
  for (int l = 0; l < 8; ++l) {
    const float* lb = sbase + (long)l * layer_floats;
    float* db = dbase + (long)l * layer_floats;
    int vm_young = 0, pend0 = -1, pend1 = -1;
    float v[8];
    auto store_kc = [&](int kc, const float (&x)[8]) __attribute__((always_inline)) {
#pragma unroll
      for (int t = 0; t < W; ++t) {
        float* p = db + t * tensor_floats + kc * 512;
        st16<1>(p, f32x4{x[0], x[1], x[2], x[3]}); st16<1>(p + 8, f32x4{x[4], x[5], x[6], x[7]});
      }
      vm_young += 2 * W;
    };
    auto flush = [&]() __attribute__((always_inline)) {
      if (pend0 >= 0) { store_kc(pend0, sv[0]); pend0 = -1; }
      if (pend1 >= 0) { store_kc(pend1, sv[1]); pend1 = -1; }
    };
    // piece u of the B preparation of k-chunk kc (values from the loads of kc, planes into bq[kc & 1])
    auto piece = [&](int kc, int u) __attribute__((always_inline)) {
      if (kc >= 16) return;
      if (R && u == 0 && kc + 1 < 16) { loads((kc + 1) & 1, lb, kc + 1); vm_young += 2 * R; }      // X3_AHEAD = 1: the next k-chunk's sources
      if (u < 8) {
        const float h = q[kc & 1][0][u >> 2][u & 3];
        float x = DEP ? accP[kc >> 1][8 * (kc & 1) + u] : 1.25f;
        if (VM == 1) v[u] = x * h;
        else {
          const float sg = 1.0f - __builtin_amdgcn_exp2f(-144.269504f * h);
          if (VM == 2) v[u] = x * sg;
          else {
            const float gu = q[kc & 1][R > 1 ? 1 : 0][u >> 2][u & 3], ab = q[kc & 1][R > 2 ? 2 : 0][u >> 2][u & 3];
            const float ga = sg > 0.f ? gu * __builtin_amdgcn_rcpf(sg) : 0.f;
            const float g2 = ga * ab * (100.f * (1.0f - sg));
            v[u] = fmaf(x, sg, g2);
          }
        }
      } else {
        if (u == 8 && W) {
          if (pend0 >= 0 && pend1 >= 0) flush();
          if (pend0 < 0) {
#pragma unroll
            for (int i = 0; i < 8; ++i) sv[0][i] = v[i];
            pend0 = kc;
          } else {
#pragma unroll
            for (int i = 0; i < 8; ++i) sv[1][i] = v[i];
            pend1 = kc;
          }
        }
        const int i = u - 8;
        const unsigned p0 = pk_bf16_(v[2 * i], v[2 * i + 1]);
        float ra = v[2 * i] - bf_lo(p0), rb = v[2 * i + 1] - bf_hi(p0);
        const unsigned p1 = pk_bf16_(ra, rb);
        ra -= bf_lo(p1); rb -= bf_hi(p1);
        bq[kc & 1][0][i] = p0; bq[kc & 1][1][i] = p1; bq[kc & 1][2][i] = pk_bf16_(ra, rb);
      }
    };
    if (R && l > 0) loads(0, lb, 0);               // (the un-hidden prologue of an op, as in dense_x3g: its first sources, then their B preparation)
#pragma unroll
    for (int u = 0; u < 12; ++u) piece(0, u);
    bool flushed = false;
#pragma unroll
    for (int gg = 0; gg < 192; ++gg) {
      const int st = gg / SGR, gs = gg % SGR, kc = gg / 12, g = gg % 12;
      const u32x4* cur = reinterpret_cast<const u32x4*>(lds + (st & 1) * (SGR * 512)) + lane;
      u32x4 ring[2][2];
      if (gs == 0) {
        if (st == 0) wait_vm<0>(); else wait_vm_n(vm_young);
        __builtin_amdgcn_s_barrier();
        flushed = false;
        ring[0][0] = cur[0]; ring[0][1] = cur[64]; ring[1][0] = cur[128]; ring[1][1] = cur[192];
      }
      const u32x4 a0 = ring[gs & 1][0], a1 = ring[gs & 1][1];
      if (gs + 2 < SGR) { ring[gs & 1][0] = cur[(2 * (gs + 2)) * 64]; ring[gs & 1][1] = cur[(2 * (gs + 2) + 1) * 64]; }
      const int sp = g % 3;      // the products of a group as in dense_x3g: W_sp with planes 0 (,1) of the activations
      acc[(4 * g) % 8] = mfma_bf16(a0, bq[kc & 1][0], acc[(4 * g) % 8]);
      acc[(4 * g + 1) % 8] = mfma_bf16(a1, bq[kc & 1][0], acc[(4 * g + 1) % 8]);
      acc[(4 * g + 2) % 8] = mfma_bf16(a0, bq[kc & 1][sp < 2 ? 1 : 2], acc[(4 * g + 2) % 8]);
      acc[(4 * g + 3) % 8] = mfma_bf16(a1, bq[kc & 1][sp < 2 ? 1 : 2], acc[(4 * g + 3) % 8]);
      if (gs < SGR / 4) {
        dma((st + 1) & 1, st + 1, 2 * gs); dma((st + 1) & 1, st + 1, 2 * gs + 1);
        if (gs == SGR / 4 - 1) { vm_young = 0; __builtin_amdgcn_sched_barrier(0); }
      }
      if (gs >= SGR / 4 - 1 && !flushed) { flush(); flushed = true; }
      piece(kc + 1, g);
      __builtin_amdgcn_sched_barrier(0);
    }
    flush();
    if (DEP) {
#pragma unroll
      for (int i = 0; i < 8; ++i) accP[i] = acc[i];
#pragma unroll
      for (int i = 0; i < 8; ++i)
#pragma unroll
        for (int j = 0; j < 16; ++j) acc[i][j] = 0.f;
    }
  }
  float r = 0.f;
#pragma unroll
  for (int i = 0; i < 8; ++i) r += acc[i][0] + acc[i][15] + accP[i][3];
  if (r == 123.456f) sink[threadIdx.x] = r;
}

// ---- latency probes: MODE 0: 2 loads; 1: 2 stores; 2: 2 stores then 2 loads; 3: 2 loads then 2 stores.  PACE: s_sleep units between probes
template <int MODE, int FL>
__global__ __launch_bounds__(256) void probe_kernel(const float* __restrict__ src, float* __restrict__ dst, long layer_floats, int pace, unsigned long long* out) {
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const long blk = (long)blockIdx.x * 4 + wave;
  const int off = (lane & 31) * 16 + 4 * (lane >> 5);
  unsigned long long sum = 0, mx = 0;
  float sink = 0.f;
  for (int l = 0; l < 8; ++l) {
    for (int kc = 0; kc < 16; ++kc) {
      const float* p = src + (long)l * layer_floats + blk * 8192 + kc * 512 + off;
      float* d = dst + (long)l * layer_floats + blk * 8192 + kc * 512 + off;
      f32x4 v = {1.f + sink, 2.f, 3.f, 4.f};
      f32x4 x0, x1;
      __builtin_amdgcn_s_waitcnt(0);
      __builtin_amdgcn_sched_barrier(0);
      const unsigned long long t0 = __builtin_amdgcn_s_memtime();
      __builtin_amdgcn_sched_barrier(0);
      if (MODE == 1 || MODE == 2) { st16<FL>(d, v); st16<FL>(d + 8, v); }
      if (MODE == 0 || MODE == 2 || MODE == 3) { x0 = ld16<0>(p); x1 = ld16<0>(p + 8); }
      if (MODE == 3) { st16<FL>(d, v); st16<FL>(d + 8, v); }
      __builtin_amdgcn_sched_barrier(0);
      __builtin_amdgcn_s_waitcnt(0x0F70);      // vmcnt(0)
      __builtin_amdgcn_sched_barrier(0);
      const unsigned long long t1 = __builtin_amdgcn_s_memtime();
      __builtin_amdgcn_s_waitcnt(0);
      __builtin_amdgcn_sched_barrier(0);
      if (MODE != 1) sink += x0.x + x1.y;
      const unsigned long long dt = t1 - t0;
      sum += dt; mx = dt > mx ? dt : mx;
      for (int s = 0; s < pace; ++s) __builtin_amdgcn_s_sleep(16);      // 16 x 64 cycles
    }
  }
  if (lane == 0) { out[2 * blk] = sum; out[2 * blk + 1] = mx; }
  if (sink == 123.456f) dst[0] = sink;
}

static float* g_src; static float* g_dst; static float* g_sink; static unsigned long long* g_out;
static const long PTS = 51200, LAYER = PTS * 256, TENSOR = LAYER * 8;      // one point range of the training step: 420 MB per tensor

template <int R, int W, int FL, int AHEAD, int LG, int SG, int FEAT>
void run(const char* what) {
  auto k = paced_kernel<R, W, FL, AHEAD, LG, SG, FEAT>;
  const int lds_bytes = 96 * 1024;
  hipFuncSetAttribute((const void*)k, hipFuncAttributeMaxDynamicSharedMemorySize, lds_bytes);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  float best[2] = {1e30f, 1e30f};
  for (int which = 0; which < 2; ++which) {
    const int grid = which ? 256 : (int)(PTS / 128);      // the range's 400 workgroups; one full round of the chip
    k<<<grid, 256, lds_bytes>>>(g_src, g_dst, TENSOR, LAYER, g_sink);
    hipDeviceSynchronize();
    for (int rep = 0; rep < 5; ++rep) {
      hipEventRecord(e0);
      k<<<grid, 256, lds_bytes>>>(g_src, g_dst, TENSOR, LAYER, g_sink);
      hipEventRecord(e1); hipEventSynchronize(e1);
      float ms; hipEventElapsedTime(&ms, e0, e1);
      if (ms < best[which]) best[which] = ms;
    }
  }
  const double gb1 = (double)(R + W) * 256 * 128 * 256 * 8 * 4 / 1e9;
  printf("%-22s R=%d W=%d st=%-5s ahead=%d loads@g%-2d stores@g%-2d%s%s%s%s: %7.1f us per range (400 wg);  one round (256 wg) %7.1f us = %5.2f TB/s\n", what, R, W,
         FL ? "nt" : "plain", AHEAD, LG, SG, FEAT & F_FIXED ? " fixed-store-region" : "", FEAT & F_LDFIXED ? " fixed-load-region" : "",
         FEAT & F_PRIO ? " setprio" : "", FEAT & F_NOMFMA ? " NO-MFMA" : "", best[0] * 1e3, best[1] * 1e3, gb1 / best[1]);
  fflush(stdout);
}


static float* g_wts;
template <int R, int W, int FL, int WAIT, int SPOS, int SGR>
void run_staged(const char* what) {
  auto k = staged_kernel<R, W, FL, WAIT, SPOS, SGR>;
  const int lds_bytes = SGR == 16 ? 96 * 1024 : 128 * 1024;
  hipFuncSetAttribute((const void*)k, hipFuncAttributeMaxDynamicSharedMemorySize, lds_bytes);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  float best[2] = {1e30f, 1e30f};
  for (int which = 0; which < 2; ++which) {
    const int grid = which ? 256 : (int)(PTS / 128);
    k<<<grid, 256, lds_bytes>>>(g_src, g_dst, g_wts, TENSOR, LAYER, g_sink);
    hipDeviceSynchronize();
    for (int rep = 0; rep < 5; ++rep) {
      hipEventRecord(e0);
      k<<<grid, 256, lds_bytes>>>(g_src, g_dst, g_wts, TENSOR, LAYER, g_sink);
      hipEventRecord(e1); hipEventSynchronize(e1);
      float ms; hipEventElapsedTime(&ms, e0, e1);
      if (ms < best[which]) best[which] = ms;
    }
  }
  const double gb1 = (double)(R + W) * 256 * 128 * 256 * 8 * 4 / 1e9;
  printf("%-22s R=%d W=%d st=%-5s weight stream, %d KB stages, wait %-10s stores %-22s: %7.1f us per range (400 wg);  one round (256 wg) %7.1f us = %5.2f TB/s\n", what, R, W,
         FL ? "nt" : "plain", SGR * 2, WAIT ? "counted" : "vmcnt(0)", SPOS == 0 ? "after the next barrier" : SPOS == 1 ? "behind the DMA pieces" : "where values exist",
         best[0] * 1e3, best[1] * 1e3, gb1 / best[1]);
  fflush(stdout);
}

template <int R, int W, int VM, bool DEP>
void run_valu(const char* what) {
  auto k = valu_kernel<R, W, VM, DEP>;
  const int lds_bytes = 96 * 1024;
  hipFuncSetAttribute((const void*)k, hipFuncAttributeMaxDynamicSharedMemorySize, lds_bytes);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  float best[2] = {1e30f, 1e30f};
  for (int which = 0; which < 2; ++which) {
    const int grid = which ? 256 : (int)(PTS / 128);
    k<<<grid, 256, lds_bytes>>>(g_src, g_dst, g_wts, TENSOR, LAYER, g_sink);
    hipDeviceSynchronize();
    for (int rep = 0; rep < 5; ++rep) {
      hipEventRecord(e0);
      k<<<grid, 256, lds_bytes>>>(g_src, g_dst, g_wts, TENSOR, LAYER, g_sink);
      hipEventRecord(e1); hipEventSynchronize(e1);
      float ms; hipEventElapsedTime(&ms, e0, e1);
      if (ms < best[which]) best[which] = ms;
    }
  }
  hipFuncAttributes fa; hipFuncGetAttributes(&fa, (const void*)k);
  printf("%-10s R=%d W=%d  VALU mix %d (%s)%s: %7.1f us per range (400 wg);  one round (256 wg) %7.1f us   [%d VGPRs, %zu B scratch]\n", what, R, W, VM,
         VM == 1 ? "three-way split" : VM == 2 ? "+ sigma = 1 - exp2" : "+ rcp, select, G2", DEP ? " + previous layer's accumulators live" : "", best[0] * 1e3, best[1] * 1e3,
         fa.numRegs, (size_t)fa.localSizeBytes);
  fflush(stdout);
}

template <int MODE, int FL>
void probe(const char* what, int pace) {
  auto k = probe_kernel<MODE, FL>;
  const int lds_bytes = 96 * 1024;
  hipFuncSetAttribute((const void*)k, hipFuncAttributeMaxDynamicSharedMemorySize, lds_bytes);
  std::vector<unsigned long long> h(2 * 1024);
  k<<<256, 256, lds_bytes>>>(g_src, g_dst, LAYER, pace, g_out);
  hipDeviceSynchronize();
  k<<<256, 256, lds_bytes>>>(g_src, g_dst, LAYER, pace, g_out);
  hipDeviceSynchronize();
  hipMemcpy(h.data(), g_out, h.size() * 8, hipMemcpyDeviceToHost);
  double s = 0; unsigned long long mx = 0;
  for (int i = 0; i < 1024; ++i) { s += (double)h[2 * i]; mx = h[2 * i + 1] > mx ? h[2 * i + 1] : mx; }
  printf("probe %-30s st=%-5s pace=%2d x 1024 cycles: mean %8.0f memtime ticks from issue to vmcnt(0), max %8llu   (1024 waves x 128 probes)\n", what, FL ? "nt" : "plain",
         pace, s / (1024.0 * 128), mx);
  fflush(stdout);
}

// PACED_RANDOM=1: the tensors and the weight stream hold pseudo-random values instead of zeros (real kernels multiply real data: the matrix
// pipe's and the memory system's power draw depend on it -- profiles/r6_sweeps_vs_model.txt)
__global__ void fill_random(float* p, long n, unsigned seed, float lo, float hi) {
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) {
    unsigned x = (unsigned)i * 2654435761u + seed;
    x ^= x >> 16; x *= 0x7feb352du; x ^= x >> 15; x *= 0x846ca68bu; x ^= x >> 16;
    p[i] = lo + (hi - lo) * (float)(x >> 8) * (1.0f / 16777216.0f);
  }
}

int main() {
  hipMalloc(&g_src, (size_t)3 * TENSOR * 4); hipMalloc(&g_dst, (size_t)2 * TENSOR * 4); hipMalloc(&g_sink, 4096); hipMalloc(&g_out, 2 * 1024 * 8);
  hipMemset(g_src, 0, (size_t)3 * TENSOR * 4); hipMemset(g_dst, 0, (size_t)2 * TENSOR * 4);
  hipMalloc(&g_wts, 12 * 32768 + 65536); hipMemset(g_wts, 0, 12 * 32768 + 65536);
  if (getenv("PACED_RANDOM")) {
    fill_random<<<4096, 256>>>(g_src, 3 * TENSOR, 1u, 0.0f, 0.05f);          // h-like: small positive activations (sigma = 1 - exp(-100 h) spans (0, 1))
    fill_random<<<4096, 256>>>(g_wts, (12 * 32768 + 65536) / 4, 7u, -1.0f, 1.0f);
    hipDeviceSynchronize();
    printf("(tensors and weights: pseudo-random values)\n");
  }
  const bool r6_only = getenv("PACED_R6_ONLY") != nullptr;
  if (!r6_only) {
  printf("== the two sides alone\n");
  run<0, 0, 1, 1, 0, 11, 0>("MFMA only");
  run<1, 1, 1, 1, 0, 11, F_NOMFMA>("memory only");
  run<3, 1, 1, 1, 0, 11, F_NOMFMA>("memory only");
  printf("== MFMA stream + accesses\n");
  run<1, 0, 1, 1, 0, 11, 0>("loads");
  run<3, 0, 1, 1, 0, 11, 0>("loads");
  run<0, 1, 1, 1, 0, 11, 0>("stores");
  run<1, 1, 1, 1, 0, 11, 0>("sweep 1");
  run<1, 1, 0, 1, 0, 11, 0>("sweep 1");
  run<3, 1, 1, 1, 0, 11, 0>("sweep 2");
  run<2, 2, 1, 1, 0, 11, 0>("round-4 sweep 1");
  printf("== where the price of the mix comes from\n");
  run<1, 1, 1, 1, 0, 11, F_FIXED>("sweep 1");
  run<1, 1, 1, 1, 0, 11, F_LDFIXED>("sweep 1");
  run<1, 1, 1, 1, 0, 11, F_PRIO>("sweep 1");
  run<3, 1, 1, 1, 0, 11, F_FIXED>("sweep 2");
  run<3, 1, 1, 1, 0, 11, F_PRIO>("sweep 2");
  printf("== placement and depth\n");
  run<1, 1, 1, 1, 0, 1, 0>("sweep 1");       // stores right after the values exist (group 1), loads at group 0
  run<1, 1, 1, 1, 6, 11, 0>("sweep 1");      // loads in the middle
  run<1, 1, 1, 1, 11, 1, 0>("sweep 1");      // stores early, loads late
  run<1, 1, 1, 2, 0, 11, 0>("sweep 1");
  run<1, 1, 1, 3, 0, 11, 0>("sweep 1");
  run<1, 1, 1, 4, 0, 11, 0>("sweep 1");
  run<3, 1, 1, 2, 0, 11, 0>("sweep 2");
  run<3, 1, 1, 3, 0, 11, 0>("sweep 2");
  run<3, 1, 1, 4, 0, 11, 0>("sweep 2");
  run<3, 1, 1, 2, 0, 1, 0>("sweep 2");
  printf("== with the weight stream (LDS DMA from L2, ds_reads, a barrier per 16 groups)\n");
  run_staged<0, 0, 1, 0, 0, 16>("MFMA only");
  run_staged<1, 0, 1, 0, 0, 16>("loads");
  run_staged<1, 0, 1, 1, 0, 16>("loads");
  run_staged<0, 1, 1, 0, 0, 16>("stores");
  run_staged<0, 1, 1, 1, 1, 16>("stores");
  run_staged<1, 1, 1, 0, 0, 16>("sweep 1");
  run_staged<1, 1, 0, 0, 0, 16>("sweep 1");
  run_staged<1, 1, 1, 0, 2, 16>("sweep 1");
  run_staged<1, 1, 1, 1, 0, 16>("sweep 1");
  run_staged<1, 1, 1, 1, 1, 16>("sweep 1");
  run_staged<1, 1, 0, 1, 1, 16>("sweep 1");
  run_staged<1, 1, 1, 1, 2, 16>("sweep 1");
  run_staged<3, 1, 1, 0, 0, 16>("sweep 2");
  run_staged<3, 1, 1, 1, 0, 16>("sweep 2");
  run_staged<3, 1, 1, 1, 1, 16>("sweep 2");
  run_staged<3, 1, 1, 1, 2, 16>("sweep 2");
  printf("== 64 KB stages: half the barriers\n");
  run_staged<0, 0, 1, 0, 0, 32>("MFMA only");
  run_staged<1, 1, 1, 0, 0, 32>("sweep 1");
  run_staged<1, 1, 1, 0, 2, 32>("sweep 1");
  run_staged<1, 1, 1, 1, 1, 32>("sweep 1");
  run_staged<1, 1, 1, 1, 2, 32>("sweep 1");
  run_staged<3, 1, 1, 0, 0, 32>("sweep 2");
  run_staged<3, 1, 1, 0, 2, 32>("sweep 2");
  run_staged<3, 1, 1, 1, 1, 32>("sweep 2");
  }
  printf("== round 6: the best model (counted waits, stores behind the DMA pieces, 32 KB stages) + the VALU work of the real B preparation, piece by piece\n");
  run_staged<1, 1, 1, 1, 1, 16>("sweep 1");
  run_valu<1, 1, 1, false>("sweep 1");
  run_valu<1, 1, 2, false>("sweep 1");
  run_valu<1, 1, 2, true>("sweep 1");
  run_staged<3, 1, 1, 1, 1, 16>("sweep 2");
  run_valu<3, 1, 1, false>("sweep 2");
  run_valu<3, 1, 2, false>("sweep 2");
  run_valu<3, 1, 3, false>("sweep 2");
  run_valu<3, 1, 3, true>("sweep 2");
  if (r6_only) return 0;
  printf("== latency probes (one wave per SIMD, 256 workgroups)\n");
  probe<0, 1>("2 loads", 0); probe<1, 1>("2 stores", 0); probe<1, 0>("2 stores", 0); probe<2, 1>("2 stores, 2 loads", 0); probe<3, 1>("2 loads, 2 stores", 0);
  probe<0, 1>("2 loads", 2); probe<1, 1>("2 stores", 2); probe<1, 0>("2 stores", 2); probe<2, 1>("2 stores, 2 loads", 2); probe<3, 1>("2 loads, 2 stores", 2);
  return 0;
}
