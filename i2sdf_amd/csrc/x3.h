// bf16x3 split arithmetic for the no-gradient SDF forward (the sampler's inner loop).
//
// fp32 products at bf16 MFMA speed: every fp32 operand is split into three bf16 terms x = x0 + x1 + x2 (round-to-nearest
// residuals, |x - x0 - x1 - x2| <= 2^-25 |x|), and W*h is accumulated in fp32 from the six partial products whose combined
// order is <= 2:  W0h0, W0h1, W1h0, W0h2, W2h0, W1h1  (the dropped ones are <= 2^-24 |W||h|, i.e. at fp32 rounding level).
// Six v_mfma_f32_32x32x16_bf16 (32 cycles each) replace eight v_mfma_f32_32x32x2_f32 (64 cycles each) per 32x32x16 block:
// 192 instead of 512 matrix-pipe cycles.
//
// Loop structure (differs from the fp32 kernels, see common.h): K-outer.  All NT output tiles of a layer accumulate
// concurrently (NT x 16 accumulator registers); the previous layer's PRE-activation accumulators stay live next to them, and
// the B operand of k-chunk kc (16 reduction indices) is made on the fly -- read 8 accumulator registers, softplus100, split --
// one k-chunk ahead of its use, dealt into the MFMA shadows.  Weights are split at pack time (SEG_WFWD3).
//
// Index maps (kg = lane>>5, element j = 0..7 of the 8 bf16 a lane holds):
//   k-chunk kc, element j   <->  reduction index 16*kc + (j&3) + 8*(j>>2) + 4*kg
//                           ==   D-layout tile kc/2, register 8*(kc&1) + j           (so accumulators feed straight back)
// Stream of one op: [NT*4 bias chunks, fp32, D layout][for kc: for g < NT/2: for split s < 3: for e < 2: chunk of tile 2g+e]
// padded to whole stages; a chunk = 64 lanes x 16 B = the A operand (32 rows x 16 k, one split plane) of one MFMA.
#pragma once
#include "common.h"

namespace i2sdf {

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
typedef unsigned u32x2 __attribute__((ext_vector_type(2)));

// global loads of the B-operand sources are issued X3_AHEAD k-chunks before their use (ring of X3_RING register slots):
// measured: a distance of 1 is enough (2 and 3 change nothing, on the kernels and on a model of them: scripts/ubench/mfma_paced.hip); what stalled
// these ops was the full vmcnt(0) drain at every stage barrier -- of the loads just issued and of the stage's stores -- hence the counted stage
// wait below and the stash / flush scheme of the stores in dense_x3g
constexpr int X3_AHEAD = 1, X3_RING = 2;
// (a source may ask for more: `static constexpr int AHEAD` of the Src, ring of AHEAD + 1 slots -- sweep 2 reads three tensors per k-chunk)
#ifndef X3_SW2_AHEAD
#define X3_SW2_AHEAD 1
#endif
// the final drain of a sweep (x3_drain: loads -> values -> stores of the last layer's sixteen k-chunks, no MFMAs in between): k-chunks of load-ahead
#ifndef X3_DRAIN_AHEAD
#define X3_DRAIN_AHEAD 1
#endif
#ifndef X3_SW2_SHARE_E
#define X3_SW2_SHARE_E 0
#endif
#ifndef X3_COUNTED_WAIT
#define X3_COUNTED_WAIT 1
#endif
// sources whose saved-tensor stores are UNCONDITIONAL instructions (Src::COUNTED: rows of padding points are written too -- every per-point
// workspace has Mp >= M rows, include/i2sdf.h) are counted as well and issued behind the stage's DMA pieces: they then cross one more barrier
#ifndef X3_COUNT_STORES
#define X3_COUNT_STORES 1
#endif
__host__ __device__ constexpr int x3_op_chunks(int NT, int KC16) { return round_up(NT * 4 + KC16 * NT * 3, SC); }

__device__ __forceinline__ f32x16 mfma_bf16(u32x4 a, u32x4 b, f32x16 c) {
  return __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, a), __builtin_bit_cast(bf16x8, b), c, 0, 0, 0);
}
__device__ __forceinline__ unsigned pk_bf16(float a, float b) {          // v_cvt_pk_bf16_f32: low half = a, high half = b (RNE)
  const f32x2 v = {a, b};
  return __builtin_bit_cast(unsigned, __builtin_convertvector(v, bf16x2));
}
__device__ __forceinline__ float bf16_lo(unsigned p) { return __builtin_bit_cast(float, p << 16); }
__device__ __forceinline__ float bf16_hi(unsigned p) { return __builtin_bit_cast(float, p & 0xffff0000u); }
// two values -> their three split planes, packed pairwise
__device__ __forceinline__ void split3_pair(float xa, float xb, unsigned& p0, unsigned& p1, unsigned& p2) {
  p0 = pk_bf16(xa, xb);
  float ra = xa - bf16_lo(p0), rb = xb - bf16_hi(p0);      // exact
  p1 = pk_bf16(ra, rb);
  ra -= bf16_lo(p1); rb -= bf16_hi(p1);                     // exact
  p2 = pk_bf16(ra, rb);
}

// two values -> their two leading split planes (bf16x2: x = x0 + x1 + O(2^-18 |x|))
__device__ __forceinline__ void split2_pair(float xa, float xb, unsigned& p0, unsigned& p1) {
  p0 = pk_bf16(xa, xb);
  p1 = pk_bf16(xa - bf16_lo(p0), xb - bf16_hi(p0));
}

// this lane's B-operand values of an input vector given in reduction-index order (NC16 k-chunks)
template <int NC16>
__device__ __forceinline__ void x3_select_pe(const float (&full)[NC16 * 16], float (&sel)[NC16 * 8], int hi) {
#pragma unroll
  for (int kc = 0; kc < NC16; ++kc)
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      const int base = 16 * kc + (j & 3) + 8 * (j >> 2);
      sel[8 * kc + j] = hi ? full[base + 4] : full[base];
    }
}

// ---------------------------------------------------------------------------------------------
// K-outer bf16x3 op:  acc[NT] (+)= W * src.
//   BIAS  : 1 = the stream starts with NT*4 fp32 bias chunks that initialise acc; 2 = the chunks are there but are skipped
//           and acc starts at zero (backward sweeps over the forward stream); 0 = weight chunks only, acc is used as the
//           caller left it (zeroed, or holding a running sum such as pbar).
//   Src   : where the B operand comes from.  `float value(kc, u)` returns value u (0..7) of k-chunk kc for this lane --
//           reduction index 16*kc + (u&3) + 8*(u>>2) + 4*hi -- and `int done(kc, v)` is called once all eight are known
//           (global stores of the saved tensors).  Both are dealt into the MFMA shadows one k-chunk ahead of their use;
//           `int ahead(kc)` is called X3_AHEAD k-chunks ahead (issue global loads there; returns how many it issued unconditionally).
// ---------------------------------------------------------------------------------------------
template <int NT, int KC16, int BIAS, class Src>
__device__ __forceinline__ void dense_x3g(WStream& ws, Src& src, f32x16 (&acc_io)[NT], int tid) {
  f32x16 acc[NT];
  if (BIAS == 0) {
#pragma unroll
    for (int nt = 0; nt < NT; ++nt) acc[nt] = acc_io[nt];
  } else if (BIAS == 2) {
#pragma unroll
    for (int nt = 0; nt < NT; ++nt)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[nt][r] = 0.f;
  }
  static_assert(NT % 2 == 0, "tiles are processed in pairs");
  constexpr int NB = BIAS != 0 ? NT * 4 : 0, G = NT / 2, PPK = G * 3, NPAIR = KC16 * PPK, NW = NPAIR * 2;
  constexpr int TOT = round_up(NB + NW, SC), NS = TOT / SC, PFP = 2;
  const int lane = tid & 63;
  float v[8], vx[8];
  u32x4 bq[2][3];
  // Stores of the source (saved tensors) are not issued where their values become known but stashed and flushed once per stage: right behind
  // the stage's last DMA piece when the source's stores are unconditional instructions (LATE: they are then YOUNGER than the pieces and the
  // counted wait of the next barrier lets them fly), else right after the next stage barrier (a full stage away from the drain they are part of).
  float sv[2][8], sx[2][8];
  u32x4 d0 = {0u, 0u, 0u, 0u}, d1 = {0u, 0u, 0u, 0u};      // the weights of the sp = 0 group, kept for the deferred W0*h2 pair
  int pk0 = -1, pk1 = -1;               // k-chunks whose stores are pending (compile-time after unrolling)
  int vm_young = 0;                     // unconditional vector-memory instructions since the last DMA piece (compile-time after unrolling)
  constexpr bool LATE = X3_COUNTED_WAIT && X3_COUNT_STORES && Src::COUNTED;      // stores behind the stage's DMA pieces, counted
  auto flush = [&]() __attribute__((always_inline)) {
    if (pk0 >= 0) { vm_young += src.done(pk0, sv[0], sx[0]); pk0 = -1; }
    if (pk1 >= 0) { vm_young += src.done(pk1, sv[1], sx[1]); pk1 = -1; }
  };
  constexpr int AH = Src::AHEAD;
  auto prep = [&](int kc, int u, u32x4 (&b)[3]) __attribute__((always_inline)) {
    if (kc >= KC16) return;
    if (u == 0 && kc + AH < KC16) vm_young += src.ahead(kc + AH);
    if (u < 8) v[u] = src.value(kc, u, vx[u]);
    else {
      if (u == 8 && Src::STORES) {
        if (pk0 >= 0 && pk1 >= 0) flush();
        if (pk0 < 0) {
#pragma unroll
          for (int i = 0; i < 8; ++i) { sv[0][i] = v[i]; sx[0][i] = vx[i]; }
          pk0 = kc;
        } else {
#pragma unroll
          for (int i = 0; i < 8; ++i) { sv[1][i] = v[i]; sx[1][i] = vx[i]; }
          pk1 = kc;
        }
      }
      const int i = u - 8;
      unsigned p0, p1, p2;
      split3_pair(v[2 * i], v[2 * i + 1], p0, p1, p2);
      b[0][i] = p0; b[1][i] = p1; b[2][i] = p2;
    }
  };
#pragma unroll
  for (int k0 = 0; k0 < AH; ++k0)
    if (k0 < KC16) (void)src.ahead(k0);
#pragma unroll
  for (int u = 0; u < 12; ++u) prep(0, u, bq[0]);
  // pairs [p0, p1) of stage s are weight chunks (the rest: bias chunks in front, padding behind) -- compile-time after unrolling
  auto first_pair = [](int s) { return (s * SC < NB) ? ((NB - s * SC < SC) ? (NB - s * SC) / 2 : SC / 2) : 0; };
  auto end_pair = [](int s) { return (NB + NW - s * SC < SC) ? ((NB + NW - s * SC > 0) ? (NB + NW - s * SC) / 2 : 0) : SC / 2; };
  u32x4 ring[PFP][2];                   // A operands (one pair of tiles, one split plane) read PFP groups ahead of their MFMAs
#pragma unroll
  for (int s = 0; s < NS; ++s) {
    const int p0 = first_pair(s), p1 = end_pair(s);
    // The stage hand-over: the DMA pieces of this stage must have landed (hipcc cannot be trusted to wait for them by itself, common.h:
    // WStream::advance).  Counted form (round 5): `s_waitcnt vmcnt(vm_young); s_barrier`, vm_young = the vector-memory instructions this
    // wave has issued since the stage's last piece -- the source loads and saved-tensor stores behind it fly across the barrier.
#if X3_COUNTED_WAIT
    // the first stage of an op was fetched by the previous op (or begin()): full drain there
    const u32x4* cur = reinterpret_cast<const u32x4*>(s == 0 ? ws.advance_barrier() : ws.advance_barrier_young(vm_young)) + lane;
#else
    const u32x4* cur = reinterpret_cast<const u32x4*>(ws.advance_barrier()) + lane;
#endif
    if (!LATE) flush();
    bool flushed = false;
#pragma unroll
    for (int j = 0; j < SC; ++j) {
      const int c = s * SC + j;
      if (c < NB && BIAS == 1) {
        const int nt = c / 4, q = c % 4;
        const f32x4 b = __builtin_bit_cast(f32x4, cur[j * 64]);
        acc[nt][4 * q + 0] = b.x; acc[nt][4 * q + 1] = b.y; acc[nt][4 * q + 2] = b.z; acc[nt][4 * q + 3] = b.w;
      }
    }
#pragma unroll
    for (int i = 0; i < PFP; ++i)
      if (p0 + i < p1) { ring[i][0] = cur[(2 * (p0 + i)) * 64]; ring[i][1] = cur[(2 * (p0 + i) + 1) * 64]; }
    __builtin_amdgcn_sched_barrier(0);
    int npiece = 0;
#pragma unroll
    for (int jp = 0; jp < SC / 2; ++jp) {
      if (jp >= p0 && jp < p1) {
        const int w = (s * SC + 2 * jp - NB) / 2;
        const int kc = w / PPK, g = (w / 3) % G, sp = w % 3, nt = 2 * g;
        const u32x4 a0 = ring[(jp - p0) % PFP][0], a1 = ring[(jp - p0) % PFP][1];
        const u32x4 (&b)[3] = bq[kc & 1];
        if (jp + PFP < p1) {
          ring[(jp - p0) % PFP][0] = cur[(2 * (jp + PFP)) * 64];
          ring[(jp - p0) % PFP][1] = cur[(2 * (jp + PFP) + 1) * 64];
        }
        acc[nt] = mfma_bf16(a0, b[0], acc[nt]);
        acc[nt + 1] = mfma_bf16(a1, b[0], acc[nt + 1]);
        if (sp < 2) {
          acc[nt] = mfma_bf16(a0, b[1], acc[nt]);
          acc[nt + 1] = mfma_bf16(a1, b[1], acc[nt + 1]);
        }
        // Four MFMAs in every group: the W0*h2 pair of the sp = 0 group (six products) is issued two groups later, in the sp = 2
        // group (two products of its own), with the sp = 0 weights kept in registers meanwhile.  Every group carries the same share
        // of the B preparation (one softplus or one split item, 7-11 VALU) and two LDS reads: behind two MFMAs they do not fit the
        // 64 cycles of matrix-pipe time, behind six most of it goes unused.
        if (sp == 0) { d0 = a0; d1 = a1; }
        if (sp == 2) {
          acc[nt] = mfma_bf16(d0, b[2], acc[nt]);
          acc[nt + 1] = mfma_bf16(d1, b[2], acc[nt + 1]);
        }
#pragma unroll
        for (int q = 0; q < 2; ++q)
          if (npiece < WStream::NPIECE) {       // next stage's DMA: 2 pieces per group, from the first group on
            ws.issue_piece(npiece, tid); ++npiece; vm_young = 0;
            // everything counted into vm_young must be emitted BEHIND the stage's last piece: a plain global access does not alias the
            // LDS DMA, so without a fence the scheduler may hoist one above it and the counted wait would over-count
            if (npiece == WStream::NPIECE) __builtin_amdgcn_sched_barrier(0);
          }
        if (LATE && !flushed && npiece == WStream::NPIECE) { flush(); flushed = true; }
        {
          const int pi = w % PPK;
#pragma unroll
          for (int u = 0; u < 12; ++u)
            if (u * PPK / 12 == pi) prep(kc + 1, u, bq[(kc + 1) & 1]);
        }
        __builtin_amdgcn_sched_barrier(0);
      }
    }
    if (npiece < WStream::NPIECE) {
#pragma unroll
      for (int i = 0; i < WStream::NPIECE; ++i)
        if (i >= npiece) { ws.issue_piece(i, tid); vm_young = 0; }
      __builtin_amdgcn_sched_barrier(0);      // (as above: counted accesses stay behind the last piece)
    }
    if (LATE && !flushed) flush();
    ws.advance_done();
  }
  flush();
#pragma unroll
  for (int nt = 0; nt < NT; ++nt) acc_io[nt] = acc[nt];
}

// apply a source to every k-chunk without a consuming op (the last epilogue of a chain: loads, products, stores)
template <int KC16, class Src>
__device__ __forceinline__ void x3_drain(Src& src) {
  float v[8], vx[8];
  constexpr int AH = Src::AHEAD;
#pragma unroll
  for (int k0 = 0; k0 < AH; ++k0)
    if (k0 < KC16) (void)src.ahead(k0);
#pragma unroll
  for (int kc = 0; kc < KC16; ++kc) {
    if (kc + AH < KC16) (void)src.ahead(kc + AH);
#pragma unroll
    for (int u = 0; u < 8; ++u) v[u] = src.value(kc, u, vx[u]);
    (void)src.done(kc, v, vx);
  }
}

// ---- packed 24-bit records of abar / G(hbar) / G(a) (I2SDF_OPT_SAVES24, mlp_common.h: "P24") ------------------------------------------
// The backward is bound by the bytes of its saved tensors (13 tensor passes per layer), and eight of those passes move tensors whose only
// fp32-exact consumer does not exist: abar, G(hbar) and G(a) are operands of the weight-gradient GEMMs, which in their default form keep 16
// significant bits of every operand anyway (I2SDF_OPT_WGRAD_BF16X2), and abar / G(hbar) feed sweep 2 only through the second-order injection
// G2 = (G(hbar)/sigma) abar 100 (1 - sigma).  With the option on these three tensors are stored with 16 significant bits (round to nearest:
// relative error <= 2^-16 per element; the two bf16 planes the GEMMs split them into keep 2^-18) in 3 bytes per value.  h stays fp32 (sigma is
// recomputed from it in three kernels).  Layout of a 32-point block of one layer: 16 k-chunks of 1536 B =
//   [32 points][hi 0..1][4 dwords: the upper 16 bits of the lane's values u = 0..7, two per dword]   1024 B   (lane (p, hi) of a 32-point wave: one 16-B access)
//   [32 points][hi 0..1][2 dwords: bits 15..8 of the same values, four per dword]                       512 B   (one 8-B access)
// i.e. both accesses of a wave instruction are contiguous (1 KB / 512 B).  The weight-gradient kernels read the same bytes by column quad
// (wgrad.hip: p24 loaders): quad q of a point = values u = 4 (q >> 1) .. + 3 of lane hi = q & 1.
// row = the tensor's layer base + p24_row_off(m) (floats); the k-chunk stride is P24_KCS floats.
constexpr int P24_KCS = 384, P24_MID = 256, P24_BLOCK = 16 * P24_KCS;      // floats: k-chunk stride, offset of the mid-byte part, 32-point block
__host__ __device__ inline int64_t p24_row_off(int64_t m) { return (m >> 5) * P24_BLOCK + (m & 31) * 8; }
struct P24Rec { u32x4 h; unsigned m0, m1; };
__device__ __forceinline__ void p24_load8(const float* row, int kc, int hi, P24Rec& r) {
  r.h = *reinterpret_cast<const u32x4*>(row + P24_KCS * kc + 4 * hi);
  const u32x2 t = *reinterpret_cast<const u32x2*>(row + P24_KCS * kc + (P24_MID - 4 * (int)(threadIdx.x & 31) + 2 * hi));      // row carries 8 floats per point
  r.m0 = t[0]; r.m1 = t[1];
}
// value u (0..7) of a record: (upper 16 bits) << 16 | (mid byte) << 8
template <int U>
__device__ __forceinline__ float p24_value(const P24Rec& r) {
  constexpr unsigned sel = ((5u + 2u * (U & 1)) << 24) | ((4u + 2u * (U & 1)) << 16) | ((unsigned)(U & 3) << 8) | 0x0cu;
  return __builtin_bit_cast(float, __builtin_amdgcn_perm(r.h[U >> 1], U < 4 ? r.m0 : r.m1, sel));
}
__device__ __forceinline__ float p24_value(const P24Rec& r, int u) {      // u is a constant after unrolling
  switch (u) {
    case 0: return p24_value<0>(r); case 1: return p24_value<1>(r); case 2: return p24_value<2>(r); case 3: return p24_value<3>(r);
    case 4: return p24_value<4>(r); case 5: return p24_value<5>(r); case 6: return p24_value<6>(r); default: return p24_value<7>(r);
  }
}
// value tt (0..3) of a column quad as the weight-gradient kernels read it: h = upper halves of values (0,1 | 2,3), m = their four mid bytes
__device__ __forceinline__ float p24_quad_value(u32x2 h, unsigned m, int tt) {      // tt is a constant after unrolling
  const unsigned sel = ((5u + 2u * (tt & 1)) << 24) | ((4u + 2u * (tt & 1)) << 16) | ((unsigned)tt << 8) | 0x0cu;
  return __builtin_bit_cast(float, __builtin_amdgcn_perm(h[tt >> 1], m, sel));
}
__device__ __forceinline__ void p24_store8(float* row, int kc, int hi, const float (&v)[8]) {
  unsigned b[8];
#pragma unroll
  for (int u = 0; u < 8; ++u) b[u] = __builtin_bit_cast(unsigned, v[u]) + 0x80u;          // round to nearest at bit 8 (magnitude: sign-magnitude bits)
  u32x4 h;
#pragma unroll
  for (int j = 0; j < 4; ++j) h[j] = __builtin_amdgcn_perm(b[2 * j + 1], b[2 * j], 0x07060302u);       // upper halves of (b[2j], b[2j+1])
  u32x2 m;
#pragma unroll
  for (int j = 0; j < 2; ++j) {
    const unsigned t0 = __builtin_amdgcn_perm(b[4 * j + 1], b[4 * j], 0x0c0c0501u);                    // byte 1 of b[4j], b[4j+1]
    const unsigned t1 = __builtin_amdgcn_perm(b[4 * j + 3], b[4 * j + 2], 0x0c0c0501u);
    m[j] = t0 | (t1 << 16);
  }
  __builtin_nontemporal_store(__builtin_bit_cast(f32x4, h), reinterpret_cast<f32x4*>(row + P24_KCS * kc + 4 * hi));
  __builtin_nontemporal_store(__builtin_bit_cast(f32x2, m), reinterpret_cast<f32x2*>(row + P24_KCS * kc + (P24_MID - 4 * (int)(threadIdx.x & 31) + 2 * hi)));
}

// B-operand sources -------------------------------------------------------------------------------------------------
// softplus100 of the previous layer's pre-activations (D layout) for k-chunks < KACC, this lane's PE values beyond;
// stores the activations (h row of the saved tensor) as they are produced
template <int NT, int KACC, int NPE, bool ST = true>
struct X3FwdSrc {
  static constexpr bool STORES = ST;             // ST = false: no saved tensor at all (sampler / sdf-only queries)
  static constexpr bool COUNTED = false;
  static constexpr int AHEAD = X3_AHEAD;
  const f32x16 (&accP)[NT]; const float (&pe)[NPE]; float* hrow; int hi; bool valid; int kcs = 16;
  __device__ __forceinline__ int ahead(int) { return 0; }
  __device__ __forceinline__ float value(int kc, int u, float&) {
    if (kc < KACC) return softplus100(accP[(kc >> 1) < NT ? (kc >> 1) : 0][8 * (kc & 1) + u]);
    return pe[8 * (kc - KACC < 0 ? 0 : kc - KACC) + u];
  }
  __device__ __forceinline__ int done(int kc, const float (&v)[8], const float (&)[8]) {
    if (kc < KACC && hrow != nullptr && valid) {
      stg4(hrow + kcs * kc + 4 * hi, f32x4{v[0], v[1], v[2], v[3]});
      stg4(hrow + kcs * kc + 8 + 4 * hi, f32x4{v[4], v[5], v[6], v[7]});
    }
    return 0;
  }
};
// values held in registers in the fp32 kernels' B layout (register 4*c + t <-> index 8*c + 4*hi + t)
template <int NREG>
struct X3RegSrc {
  static constexpr bool STORES = false;
  static constexpr bool COUNTED = false;
  static constexpr int AHEAD = X3_AHEAD;
  const float (&r)[NREG];
  __device__ __forceinline__ int ahead(int) { return 0; }
  __device__ __forceinline__ float value(int kc, int u, float&) { return r[8 * kc + u]; }
  __device__ __forceinline__ int done(int, const float (&)[8], const float (&)[8]) { return 0; }
};
// reverse chain: abar = (previous op's accumulators) * sigma(h) with h re-read from the saved tensor; stores abar
// UNC: abrow is known to be a row of an (Mp, 256) tensor -> unconditional, counted stores (the training forward; eval renders pass no abars)
// P24: abrow is a row of the packed 24-bit layout (p24_store8; only with UNC)
template <int NT, bool UNC = false, bool P24 = false>
struct X3RevSrc {
  static constexpr bool STORES = true;
  static constexpr bool COUNTED = UNC;
  static constexpr int AHEAD = X3_AHEAD;
  const f32x16 (&accP)[NT]; const float* hrow; float* abrow; int hi; bool valid; int kcs = 16;
  f32x4 hq[X3_RING][2];
  // ahead(): returns the number of vector-memory instructions it issues unconditionally (dense_x3g's counted stage wait)
  __device__ __forceinline__ int ahead(int kc) {
    // (non-temporal: plain loads as in the sweeps measured no change for this kernel -- it is not bound by its loads; profiles/r5_hbm_mix.txt)
    hq[kc % X3_RING][0] = ldg4(hrow + kcs * kc + 4 * hi);
    hq[kc % X3_RING][1] = ldg4(hrow + kcs * kc + 8 + 4 * hi);
    return 2;
  }
  __device__ __forceinline__ float value(int kc, int u, float&) {
    return accP[kc >> 1][8 * (kc & 1) + u] * sp_sigma_from_h(hq[kc % X3_RING][u >> 2][u & 3]);
  }
  __device__ __forceinline__ int done(int kc, const float (&v)[8], const float (&)[8]) {
    if (P24) p24_store8(abrow, kc, hi, v);
    else if (UNC || (abrow != nullptr && valid)) {
      stg4(abrow + kcs * kc + 4 * hi, f32x4{v[0], v[1], v[2], v[3]});
      stg4(abrow + kcs * kc + 8 + 4 * hi, f32x4{v[4], v[5], v[6], v[7]});
    }
    return UNC ? 2 : 0;
  }
};


// ---- sources of the backward sweeps (appendix A.3) ------------------------------------------------------------------
// two f32x4 of a row covering this lane's 8 reduction indices of k-chunk kc; kcs = floats between consecutive k-chunks of the
// row (16: point-major [M][256]; 512: blocked [M/32][16][32][16], mlp_common.h save_row_off)
// Loads of the sweeps' sources: PLAIN (temporal), not non-temporal like the other saved-tensor accesses (common.h: ldg4).  A lane reads its
// point's 64-byte record of a k-chunk with TWO instructions (bytes [0, 32) and [32, 64) of the record: the MFMA D layout), so every 128-byte line
// is touched by two instructions a few cycles apart; a non-temporal load does not keep the line for the second one.  scripts/ubench/hbm_mix.hip
// (profiles/r5_hbm_mix.txt), pure access pattern of sweep 2 (3 tensors read, 1 written): 396 us per range with non-temporal loads, 340 us
// with plain ones; the kernels: i2sdf_sdf_backward 1.61 -> 1.46 ms.  (Stores stay non-temporal: no difference once the loads are plain.)
#ifndef X3_SWEEP_LD_PLAIN
#define X3_SWEEP_LD_PLAIN 1
#endif
__device__ __forceinline__ f32x4 x3_ldg4(const float* p) {
#if X3_SWEEP_LD_PLAIN
  return *reinterpret_cast<const f32x4*>(p);
#else
  return ldg4(p);
#endif
}
__device__ __forceinline__ void x3_load8(const float* row, int kc, int hi, f32x4 (&q)[2], int kcs = 16) {
  q[0] = x3_ldg4(row + kcs * kc + 4 * hi);
  q[1] = x3_ldg4(row + kcs * kc + 8 + 4 * hi);
}
__device__ __forceinline__ void x3_store8(float* row, int kc, int hi, const float (&v)[8], int kcs = 16) {
  stg4(row + kcs * kc + 4 * hi, f32x4{v[0], v[1], v[2], v[3]});
  stg4(row + kcs * kc + 8 + 4 * hi, f32x4{v[4], v[5], v[6], v[7]});
}

// sweep 1: from G(abar_l) (accumulators):  G(hbar_{l+1}) = G(abar_l) sigma_l  [value, stored to gurow]
// Round 5: the second-order injection G2(a_l) = G(abar_l) abar_l 100 (1 - sigma_l) is no longer written here (and re-read by sweep 2): sweep 2
// RECOVERS G(abar_l) = G(hbar_{l+1}) / sigma_l from the tensor stored above -- the same sigma bits from the same h, so the quotient is
// G(abar_l) to one rounding -- and forms G2 itself.  One store pass of this sweep and its read of abar become one more read pass of
// sweep 2; the knock-outs of profiles/r5_step0_knockouts.txt price a store pass of the sweeps at ~4x a load pass.
// AH: k-chunks of load-ahead (the final drain of a sweep has no MFMAs to hide a load behind: it asks for X3_DRAIN_AHEAD)
template <int NT, int KACC, int NREG, int AH = X3_AHEAD, bool P24 = false>
struct X3Sweep1Src {
  static constexpr bool STORES = true;
  static constexpr bool COUNTED = true;
  static constexpr int AHEAD = AH, RING = AH + 1;
  const f32x16 (&accP)[NT]; const float (&tailreg)[NREG];     // k-chunks >= KACC: registers in the fp32 kernels' B layout
  const float* hrow; float* gurow; int hi; int kcs = 16;
  f32x4 hq[RING][2];
  __device__ __forceinline__ int ahead(int kc) {
    if (kc < KACC) x3_load8(hrow, kc, hi, hq[kc % RING], kcs);
    return kc < KACC ? 2 : 0;
  }
  __device__ __forceinline__ float value(int kc, int u, float&) {
    if (kc >= KACC) return tailreg[8 * (kc - KACC < 0 ? 0 : kc - KACC) + u];
    const float ga = accP[(kc >> 1) < NT ? (kc >> 1) : 0][8 * (kc & 1) + u];
    return ga * sp_sigma_from_h(hq[kc % RING][u >> 2][u & 3]);
  }
  __device__ __forceinline__ int done(int kc, const float (&v)[8], const float (&)[8]) {      // unconditional: padding points write their own rows
    if (kc < KACC) { if (P24) p24_store8(gurow, kc, hi, v); else x3_store8(gurow, kc, hi, v, kcs); }
    return kc < KACC ? 2 : 0;
  }
};
// sweep 2: G(a_l) = (accumulators [+ sb * w_sdf]) * sigma_l + G2(a_l)   [value, stored to grow]
//          G2(a_l) = (G(hbar_{l+1}) / sigma_l) abar_l 100 (1 - sigma_l)   from sweep 1's stored G(hbar_{l+1}) (gurow) and the forward's abar_l (arow);
//          sigma_l = 0 (h_{l+1} underflowed to 0: abar_l = 0 and G(hbar_{l+1}) = 0 as well) -> G2 = 0
// P24: arow / grow are rows of the packed 24-bit layout (hrow stays fp32); G24: gurow too (the top layer's G(hbar_{L-1}) stays fp32: its other reader is a
// narrow weight-gradient task whose two jobs would otherwise differ in their operand format)
template <int NT, bool TOP, int AH = X3_SW2_AHEAD, bool P24 = false, bool G24 = P24>
struct X3Sweep2Src {
  static constexpr bool STORES = true;
  static constexpr bool COUNTED = true;
  static constexpr int AHEAD = AH, RING = AHEAD + 1;     // three tensors per k-chunk: the loads of AHEAD k-chunks in flight per wave
  const f32x16 (&accP)[NT]; const float* hrow; const float* gurow; const float* arow; float* grow; int hi;
  float sb; const float* wsdf;        // TOP: w_sdf in stream layout (chunk of 8 indices = 64 lanes x 16 B), + lane*4 applied
  int kcs = 16;
  f32x4 hq[RING][2], gq[G24 ? 1 : RING][2], aq[P24 ? 1 : RING][2], wq[RING][2];
  P24Rec gr[G24 ? RING : 1], ar[P24 ? RING : 1];
  __device__ __forceinline__ int ahead(int kc) {
    x3_load8(hrow, kc, hi, hq[kc % RING], kcs);
    if (G24) p24_load8(gurow, kc, hi, gr[kc % RING]); else x3_load8(gurow, kc, hi, gq[kc % RING], kcs);
    if (P24) p24_load8(arow, kc, hi, ar[kc % RING]); else x3_load8(arow, kc, hi, aq[kc % RING], kcs);
    if (TOP) {
      wq[kc % RING][0] = *reinterpret_cast<const f32x4*>(wsdf + (2 * kc) * CHUNK_FLOATS);
      wq[kc % RING][1] = *reinterpret_cast<const f32x4*>(wsdf + (2 * kc + 1) * CHUNK_FLOATS);
    }
    return TOP ? 8 : 6;
  }
  __device__ __forceinline__ float value(int kc, int u, float&) {
    float x = accP[kc >> 1][8 * (kc & 1) + u];
    if (TOP) x = fmaf(sb, wq[kc % RING][u >> 2][u & 3], x);
    const float gu_ = G24 ? p24_value(gr[G24 ? kc % RING : 0], u) : gq[G24 ? 0 : kc % RING][u >> 2][u & 3];
    const float ab_ = P24 ? p24_value(ar[P24 ? kc % RING : 0], u) : aq[P24 ? 0 : kc % RING][u >> 2][u & 3];
#if X3_SW2_SHARE_E
    // e = exp(-100 h) once: sigma = 1 - e, and (1 - sigma) IS e (exact where the fp32 subtraction 1 - sigma only rounds it again)
    const float e = __builtin_amdgcn_exp2f(-100.f * 1.44269504088896341f * hq[kc % RING][u >> 2][u & 3]);
    const float sg = 1.0f - e;
    const float ga = sg > 0.f ? gu_ * __builtin_amdgcn_rcpf(sg) : 0.f;       // G(abar_l)
    const float g2 = ga * ab_ * (100.f * e);
#else
    const float sg = sp_sigma_from_h(hq[kc % RING][u >> 2][u & 3]);
    const float ga = sg > 0.f ? gu_ * __builtin_amdgcn_rcpf(sg) : 0.f;       // G(abar_l)
    const float g2 = ga * ab_ * (100.f * (1.0f - sg));
#endif
    return fmaf(x, sg, g2);
  }
  __device__ __forceinline__ int done(int kc, const float (&v)[8], const float (&)[8]) {      // unconditional: padding points write their own rows
    if (P24) p24_store8(grow, kc, hi, v); else x3_store8(grow, kc, hi, v, kcs);
    return 2;
  }
};
// a point-major row in global memory (or zeros) as B operand
struct X3RowSrc {
  static constexpr bool STORES = false;
  static constexpr bool COUNTED = false;
  static constexpr int AHEAD = X3_AHEAD;
  const float* row; int hi; bool on;
  f32x4 q[X3_RING][2];
  __device__ __forceinline__ int ahead(int kc) {
    if (on) x3_load8(row, kc, hi, q[kc % X3_RING]);
    else { q[kc % X3_RING][0] = f32x4{0.f, 0.f, 0.f, 0.f}; q[kc % X3_RING][1] = f32x4{0.f, 0.f, 0.f, 0.f}; }
    return 0;      // conditional on a run-time flag: not counted
  }
  __device__ __forceinline__ float value(int kc, int u, float&) { return q[kc % X3_RING][u >> 2][u & 3]; }
  __device__ __forceinline__ int done(int, const float (&)[8], const float (&)[8]) { return 0; }
};

}  // namespace i2sdf
