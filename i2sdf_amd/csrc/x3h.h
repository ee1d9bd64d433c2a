// bf16x3 split arithmetic (x3.h) with 16-POINT waves: two waves per SIMD.
//
// The K-outer ops of x3.h give a wave 32 points and v_mfma_f32_32x32x16_bf16: two accumulator sets of 128 registers each (this
// layer, previous layer) plus operands = 400-490 registers, i.e. ONE wave per SIMD.  That wave issues in order: every stage
// barrier, every LDS latency behind it, every wait for a saved-tensor load and every burst of B-preparation VALU work leaves
// the matrix pipe idle (67-70 % busy in the compute-bound kernels, 27-36 % in the memory-bound sweeps), and the bytes one CU
// keeps in flight are what four waves can issue.  Here a wave owns 16 points and multiplies with v_mfma_f32_16x16x32_bf16:
//   tile   = 16 output features x 16 points = 4 accumulator registers (f32x4); a 256-wide layer = 16 tiles = 64 registers,
//            two sets = 128, and the whole wave fits 256 registers -> two waves per SIMD, 8 per CU, which fill each other's stalls;
//   k-chunk = 32 reduction indices; the D layout of a tile PAIR is the B layout of one k-chunk, so accumulators feed straight
//            back as in x3.h (lane = point p + 16*kg, kg = lane>>4):
//              tile nt, register r          <->  feature 16*nt + 4*kg + r
//              k-chunk c, element j (0..7)  <->  reduction index 32*c + 16*(j>>2) + 4*kg + (j&3)  ==  tile 2c + (j>>2), register j&3
//   weights: one chunk (64 lanes x 16 B) = the A operand (16 rows x 32 k, one split plane) of one MFMA, lane (i = lane&15, kg)
//            holding row 16*nt + i, reduction indices of k-chunk c as above.  Same bytes per layer as the 32-point stream, read
//            from LDS twice as often per point (48 ds_read_b128 per 96 MFMAs and wave: ~50 % of the LDS read rate of a CU).
// Stream of one op: [NT bias chunks, fp32, D layout][for c: for g < NT/2: for split s < 3: for e < 2: chunk of tile 2g+e], padded
// to whole stages.  The saved per-point tensors keep their layout (mlp_common.h): a lane reads / writes the 16 B at float 4*kg of
// the 16-float groups 2c and 2c+1 of its point -- 16 points x 64 B = 1 KB contiguous per wave instruction in the blocked layout --
// so kernels of both families can be mixed freely along a chain (and wgrad3p reads what either wrote).
//
// NW = waves per workgroup: 8 -- one workgroup per CU, one weight stage in LDS shared by all eight waves (4, i.e. two independent
// workgroups per CU with twice the L2 -> LDS weight traffic, measured 5 % slower on the sampler pass).
//
// What it bought (round 4, MI355X, profiles/r4_wave16_experiments.txt): the sampler pass 551 -> 517 us per 131 072 points (matrix pipe
// 68 -> 72 % busy at a higher clock), the training forward 314 -> 289 us, the radiance forward / backward 159 -> 147 / 183 -> 176 us per
// half batch.  The d sdf/dx chain and the backward sweeps were built the same way and measured SLOWER (they need the 256 registers for
// their saved-tensor operands: spills; and without spills sweep 2 ran 399 vs 387 us -- memory-bound kernels gain nothing from a second
// wave that moves the same bytes), so they stay on 32-point waves (mlp_x3.hip).  A micro-benchmark of the instruction mix
// (scripts/ubench/mfma16_mix.hip) holds 0.91-0.97 matrix-pipe occupancy; the kernels hold 0.72, and knock-outs attribute the gap to
// the B preparation (~1 VALU instruction per MFMA costs ~1.7 matrix-pipe cycles each whatever its placement: 0.80 -> 0.72), the stage
// barrier + DMA (0.06) and the LDS reads (0.03); with four waves per SIMD (a knock-out that frees the registers) the same code reaches 0.88.
#pragma once
#include "x3.h"

namespace i2sdf {

constexpr int HP = 16;                    // points per wave
constexpr int XH_AHEAD = 1, XH_RING = XH_AHEAD + 1;      // k-chunks between the global loads of a source and the B preparation that consumes them (2 measured no gain)
constexpr int SCH = SC;                    // chunks per LDS stage (80 KB stages = 5 per 256x256 op instead of 13 measured no gain: the loss is not per barrier)
constexpr int LDS_BYTES_H = LDS_BYTES;
template <int NW> using WStreamH = WStreamT<NW * 64, SCH>;

__host__ __device__ constexpr int x3h_op_chunks(int NT, int KC32, int PL = 3) { return round_up(NT + KC32 * NT * PL, SCH); }
__host__ __device__ constexpr int x3h_bwd_chunks(int KT, int KC32) { return round_up(KC32 * KT * 3, SCH); }
// row vectors: [NROWS*16 chunks: lane (.,kg) of chunk (row, nt) holds w_row[16 nt + 4 kg + 0..3]][1 scalar chunk]
__host__ __device__ constexpr int rowvec_h_chunks(int NTK, int nrows) { return round_up(nrows * NTK + 1, SCH); }

__device__ __forceinline__ f32x4 mfma_bf16h(u32x4 a, u32x4 b, f32x4 c) {
  return __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, a), __builtin_bit_cast(bf16x8, b), c, 0, 0, 0);
}

// this lane's B-operand values of an input vector given in reduction-index order (NC32 k-chunks of 32, zero padded)
template <int NC32>
__device__ __forceinline__ void x3h_select(const float (&full)[NC32 * 32], float (&sel)[NC32 * 8], int kg) {
  // bit selects (v_bfi_b32) on lane masks: written as ternaries the four-way choice becomes a table in scratch memory
  const unsigned m1 = (kg & 1) ? 0xffffffffu : 0u, m2 = (kg & 2) ? 0xffffffffu : 0u;
  auto pick = [](unsigned m, float a, float b) __attribute__((always_inline)) {      // m ? a : b
    return __builtin_bit_cast(float, (__builtin_bit_cast(unsigned, a) & m) | (__builtin_bit_cast(unsigned, b) & ~m));
  };
#pragma unroll
  for (int c = 0; c < NC32; ++c)
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      const int b = 32 * c + 16 * (j >> 2) + (j & 3);
      sel[8 * c + j] = pick(m2, pick(m1, full[b + 12], full[b + 8]), pick(m1, full[b + 4], full[b]));
    }
}

// ---------------------------------------------------------------------------------------------
// K-outer bf16x3 op on 16-point waves:  acc[NT] (+)= W * src      (BIAS and Src as in x3.h: dense_x3g)
// Per group: one tile pair, one split plane = two A chunks, four MFMAs (the W0*h2 pair of the sp = 0 group rides in the sp = 2
// group), one twelfth of the next k-chunk's B preparation in every other group, the next stage's DMA pieces in the first groups.
// ---------------------------------------------------------------------------------------------
// PL = split planes per operand: 3 = the fp32-equivalent form (six products); 2 = bf16x2 (x = x0 + x1 + O(2^-18 |x|), the three products
// W0h0 + W0h1 + W1h0; the weight stream then holds two planes per tile pair: SEG_WFWD2H).  Only the sampler's sdf-only passes may run with
// PL = 2 (I2SDF_OPT_SAMPLER_BF16X2): they choose depths, no returned value is computed from them.
template <int NT, int KC32, int BIAS, int NW, class Src, int PL = 3>
__device__ __forceinline__ void dense_x3h(WStreamH<NW>& ws, Src& src, f32x4 (&acc_io)[NT], int tid) {
  using WS = WStreamH<NW>;
  f32x4 acc[NT];
  if (BIAS == 0) {
#pragma unroll
    for (int nt = 0; nt < NT; ++nt) acc[nt] = acc_io[nt];
  } else if (BIAS == 2) {
#pragma unroll
    for (int nt = 0; nt < NT; ++nt) acc[nt] = f32x4{0.f, 0.f, 0.f, 0.f};
  }
  static_assert(NT % 2 == 0, "tiles are processed in pairs");
  static_assert(PL == 2 || PL == 3, "split planes");
  constexpr int NB = BIAS != 0 ? NT : 0, G = NT / 2, PPK = G * PL, NPAIR = KC32 * PPK, NWC = NPAIR * 2;
  constexpr int NM = G * (PL == 3 ? 12 : 6);        // MFMAs (= gaps) per k-chunk
  constexpr int TOT = round_up(NB + NWC, SCH), NS = TOT / SCH, PFP = 2;
  static_assert(NB % 2 == 0, "bias chunks come in pairs");
  const int lane = tid & 63;
  float v[8], vx[8];
  u32x4 bq[2][3];
  u32x4 d0 = {0u, 0u, 0u, 0u}, d1 = {0u, 0u, 0u, 0u};      // the weights of the sp = 0 group, kept for the deferred W0*h2 pair
  // B preparation of a k-chunk in NU = 32 pieces of 2-6 VALU instructions, dealt evenly over the MFMA gaps of the k-chunk (96 for a 256-wide op: one piece in every third gap).
  // Pieces 0..23: the 8 values as a three-deep software pipeline over the source functor's phases (for softplus: exp2 | log2 |
  // fma), each phase its own piece, so that no piece waits for a transcendental of its own and none is longer than ~24 cycles: a
  // wave that sits in a long VALU burst issues no MFMAs, and with two in-order waves per SIMD the matrix pipe idles whenever both
  // do.  Pieces 24..31 split the 4 value pairs, two halves each (leading plane + residuals; the two lower planes).
  constexpr int NU = 32;
  float ra[4], rb[4], t1[2], t2[2];
  int vm_young = 0;      // unconditional vector-memory instructions since the last DMA piece (x3.h: the counted stage wait)
  auto prep = [&](int kc, int j, u32x4 (&b)[3]) __attribute__((always_inline)) {
    if (kc >= KC32) return;
    if (j == 0 && kc + XH_AHEAD < KC32) vm_young += src.ahead(kc + XH_AHEAD);
    if (j < 24) {
      // slot s = 0..9 holds [p3(s-2)] [p2(s-1)] [p1(s)]; the 24 valid pieces in that order
      int s_ = 0, ph = 0, n = 0;
      for (int s2 = 0; s2 < 10; ++s2)
        for (int q = 0; q < 3; ++q) {
          const int u = q == 0 ? s2 - 2 : (q == 1 ? s2 - 1 : s2);
          if (u < 0 || u > 7) continue;
          if (n == j) { s_ = u; ph = q; }
          ++n;
        }
      const int u = s_;
      if (ph == 0) v[u] = src.p3(kc, u, t1[u & 1], t2[u & 1], vx[u]);
      else if (ph == 1) t2[u & 1] = src.p2(kc, u, t1[u & 1]);
      else t1[u & 1] = src.p1(kc, u);
    } else {
      if (j == 24 && Src::STORES) vm_young += src.done(kc, v, vx);      // (issuing them behind the next stage barrier instead -- the 32-point kernels' stash -- measured no gain here)
      const int i = (j - 24) >> 1;
      if (((j - 24) & 1) == 0) {
        const unsigned p0 = pk_bf16(v[2 * i], v[2 * i + 1]);
        ra[i] = v[2 * i] - bf16_lo(p0); rb[i] = v[2 * i + 1] - bf16_hi(p0);      // exact
        b[0][i] = p0;
      } else {
        const unsigned p1 = pk_bf16(ra[i], rb[i]);
        b[1][i] = p1;
        if (PL == 3) b[2][i] = pk_bf16(ra[i] - bf16_lo(p1), rb[i] - bf16_hi(p1));
      }
    }
  };
#pragma unroll
  for (int k0 = 0; k0 < XH_AHEAD; ++k0)
    if (k0 < KC32) (void)src.ahead(k0);
#pragma unroll
  for (int u = 0; u < NU; ++u) prep(0, u, bq[0]);
  auto first_pair = [](int s) { return (s * SCH < NB) ? ((NB - s * SCH < SCH) ? (NB - s * SCH) / 2 : SCH / 2) : 0; };
  auto end_pair = [](int s) { return (NB + NWC - s * SCH < SCH) ? ((NB + NWC - s * SCH > 0) ? (NB + NWC - s * SCH) / 2 : 0) : SCH / 2; };
  u32x4 ring[PFP][2];                   // A operands (one pair of tiles, one split plane) read PFP groups ahead of their MFMAs
#pragma unroll
  for (int s = 0; s < NS; ++s) {
    const int p0 = first_pair(s), p1 = end_pair(s);
#if X3_COUNTED_WAIT
    const u32x4* cur = reinterpret_cast<const u32x4*>(s == 0 ? ws.advance_barrier() : ws.advance_barrier_young(vm_young)) + lane;
#else
    const u32x4* cur = reinterpret_cast<const u32x4*>(ws.advance_barrier()) + lane;
#endif
#pragma unroll
    for (int j = 0; j < SCH; ++j) {
      const int c = s * SCH + j;
      if (c < NB && BIAS == 1) acc[c] = __builtin_bit_cast(f32x4, cur[j * 64]);
    }
#pragma unroll
    for (int i = 0; i < PFP; ++i)
      if (p0 + i < p1) { ring[i][0] = cur[(2 * (p0 + i)) * 64]; ring[i][1] = cur[(2 * (p0 + i) + 1) * 64]; }
    __builtin_amdgcn_sched_barrier(0);
    int npiece = 0;
#pragma unroll
    for (int jp = 0; jp < SCH / 2; ++jp) {
      if (jp >= p0 && jp < p1) {
        const int w = (s * SCH + 2 * jp - NB) / 2;
        const int kc = w / PPK, g = (w / PL) % G, sp = w % PL, nt = 2 * g;
        const u32x4 a0 = ring[(jp - p0) % PFP][0], a1 = ring[(jp - p0) % PFP][1];
        const u32x4 (&b)[3] = bq[kc & 1];
        if (jp + PFP < p1) {
          ring[(jp - p0) % PFP][0] = cur[(2 * (jp + PFP)) * 64];
          ring[(jp - p0) % PFP][1] = cur[(2 * (jp + PFP) + 1) * 64];
        }
        // four units, one MFMA each, fenced: consecutive MFMAs never share an accumulator (a dependent 16x16x32 issued right behind
        // its producer waits out the pipeline latency), and the group's other work is dealt behind them instead of clustering
        // (PL = 2: the W0 group has four units -- W0h0, W0h1 of both tiles -- the W1 group two)
        const int gbase = PL == 3 ? 4 * (w % PPK) : 6 * g + 4 * sp;
        auto gap = [&](int q) __attribute__((always_inline)) {        // MFMA gap gbase + q of the k-chunk (NM gaps): piece j sits in gap j*NM/NU
          const int gi = gbase + q;
#pragma unroll
          for (int j = 0; j < NU; ++j)
            if (j * NM / NU == gi) prep(kc + 1, j, bq[(kc + 1) & 1]);
        };
        acc[nt] = mfma_bf16h(a0, b[0], acc[nt]);
        gap(0);
        __builtin_amdgcn_sched_barrier(0);
        acc[nt + 1] = mfma_bf16h(a1, b[0], acc[nt + 1]);
        if (PL == 2 && sp == 1 && npiece < WS::NPIECE) {
          ws.issue_piece(npiece, tid); ++npiece; vm_young = 0;
          if (npiece == WS::NPIECE) __builtin_amdgcn_sched_barrier(0);
        }
        gap(1);
        __builtin_amdgcn_sched_barrier(0);
        if (PL == 3 || sp == 0) {
          if (PL == 3 && sp == 0) { d0 = a0; d1 = a1; }
          if (sp < 2) acc[nt] = mfma_bf16h(a0, b[1], acc[nt]);
          else acc[nt] = mfma_bf16h(d0, b[2], acc[nt]);
          if (npiece < WS::NPIECE) {          // next stage's DMA: one piece per group, from the first group on
            ws.issue_piece(npiece, tid); ++npiece; vm_young = 0;
            if (npiece == WS::NPIECE) __builtin_amdgcn_sched_barrier(0);      // counted accesses stay behind the stage's last piece (x3.h)
          }
          gap(2);
          __builtin_amdgcn_sched_barrier(0);
          if (sp < 2) acc[nt + 1] = mfma_bf16h(a1, b[1], acc[nt + 1]);
          else acc[nt + 1] = mfma_bf16h(d1, b[2], acc[nt + 1]);
          gap(3);
          __builtin_amdgcn_sched_barrier(0);
        }
      }
    }
    if (npiece < WS::NPIECE) {
#pragma unroll
      for (int i = 0; i < WS::NPIECE; ++i)
        if (i >= npiece) { ws.issue_piece(i, tid); vm_young = 0; }
      __builtin_amdgcn_sched_barrier(0);
    }
    ws.advance_done();
  }
#pragma unroll
  for (int nt = 0; nt < NT; ++nt) acc_io[nt] = acc[nt];
}

// Row-vector op on the D-layout activations of a 16-point wave: out[row] = sum_k w_row[k] * in[k] (+ scalar).
// in[4*nt + r] = value of feature 16*nt + 4*kg + r; the four kg groups of a point are summed by two lane exchanges.
template <int NROWS, int NTK, int NW>
__device__ __forceinline__ void rowvec_h(WStreamH<NW>& ws, const float (&in)[NTK * 4], float (&out)[NROWS], int tid) {
  constexpr int TOT = rowvec_h_chunks(NTK, NROWS), NS = TOT / SCH, NWC = NROWS * NTK;
  const int lane = tid & 63;
  float part[NROWS];
#pragma unroll
  for (int r = 0; r < NROWS; ++r) part[r] = 0.f;
  f32x4 sc = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
  for (int s = 0; s < NS; ++s) {
    const f32x4* cur = reinterpret_cast<const f32x4*>(ws.advance(tid)) + lane;
#pragma unroll
    for (int j = 0; j < SCH; ++j) {
      const int c = s * SCH + j;
      if (c < NWC) {
        const int row = c / NTK, nt = c % NTK;
        const f32x4 w = cur[j * 64];
        part[row] = fmaf(w.x, in[nt * 4 + 0], part[row]);
        part[row] = fmaf(w.y, in[nt * 4 + 1], part[row]);
        part[row] = fmaf(w.z, in[nt * 4 + 2], part[row]);
        part[row] = fmaf(w.w, in[nt * 4 + 3], part[row]);
      } else if (c == NWC) {
        sc = cur[j * 64];
      }
    }
  }
#pragma unroll
  for (int r = 0; r < NROWS; ++r) {
    float x = part[r] + __shfl_xor(part[r], 16);
    x += __shfl_xor(x, 32);
    out[r] = x + (r == 0 ? sc.x : r == 1 ? sc.y : r == 2 ? sc.z : sc.w);
  }
}

// two f32x4 of a saved row: this lane's 8 reduction indices of k-chunk c (kcs = floats between consecutive 16-float groups of the
// row: 16 point-major, 512 blocked)
__device__ __forceinline__ void x3h_load8(const float* row, int c, int kg, f32x4 (&q)[2], int kcs) {
  q[0] = ldg4(row + kcs * (2 * c) + 4 * kg);
  q[1] = ldg4(row + kcs * (2 * c + 1) + 4 * kg);
}
__device__ __forceinline__ void x3h_store8(float* row, int c, int kg, const float (&v)[8], int kcs) {
  stg4(row + kcs * (2 * c) + 4 * kg, f32x4{v[0], v[1], v[2], v[3]});
  stg4(row + kcs * (2 * c + 1) + 4 * kg, f32x4{v[4], v[5], v[6], v[7]});
}

// B-operand sources (the twins of x3.h's) ---------------------------------------------------------------------------------
// softplus100 of the previous layer's pre-activations for k-chunks < KACC, this lane's PE values beyond; stores h
template <int NT, int KACC, int NPE, bool ST = true>
struct XhFwdSrc {
  static constexpr bool STORES = ST;
  const f32x4 (&accP)[NT]; const float (&pe)[NPE]; float* hrow; int kg; bool valid; int kcs = 16;
  __device__ __forceinline__ int ahead(int) { return 0; }
  // softplus100 in three phases (common.h: softplus100): z = exp2(-|a| c) | l = log2(1 + z) | h = l k + max(a, 0)
  __device__ __forceinline__ float pre(int kc, int u) const { return accP[(2 * kc + (u >> 2)) < NT ? (2 * kc + (u >> 2)) : 0][u & 3]; }
  __device__ __forceinline__ float p1(int kc, int u) {
    return kc < KACC ? __builtin_amdgcn_exp2f(-fabsf(pre(kc, u)) * (100.f * 1.44269504088896341f)) : 0.f;
  }
  __device__ __forceinline__ float p2(int kc, int, float z) { return kc < KACC ? __builtin_amdgcn_logf(1.0f + z) : 0.f; }
  __device__ __forceinline__ float p3(int kc, int u, float, float l, float&) {
    if (kc < KACC) return fmaf(l, 0.693147180559945309f * 0.01f, relu0(pre(kc, u)));
    return pe[8 * (kc - KACC < 0 ? 0 : kc - KACC) + u];
  }
  __device__ __forceinline__ int done(int kc, const float (&v)[8], const float (&)[8]) {
    if (kc < KACC && hrow != nullptr && valid) x3h_store8(hrow, kc, kg, v, kcs);
    return 0;      // predicated: not counted (see sdf_train_fwd3h_kernel)
  }
};
// values held in registers in D layout order (register 4*nt + r <-> feature 16*nt + 4*kg + r; k-chunk c = registers 8c .. 8c+7)
template <int NREG>
struct XhRegSrc {
  static constexpr bool STORES = false;
  const float (&r)[NREG];
  __device__ __forceinline__ int ahead(int) { return 0; }
  __device__ __forceinline__ float p1(int, int) { return 0.f; }
  __device__ __forceinline__ float p2(int, int, float) { return 0.f; }
  __device__ __forceinline__ float p3(int kc, int u, float, float, float&) { return r[8 * kc + u]; }
  __device__ __forceinline__ int done(int, const float (&)[8], const float (&)[8]) { return 0; }
};

// ---- radiance net ----------------------------------------------------------------------------------------------------
// ReLU of the previous layer's pre-activations; stores the activations r (saved tensor)
template <int NT>
struct XhReluSrc {
  static constexpr bool STORES = true;
  const f32x4 (&accP)[NT]; float* rrow; int kg; bool valid; int kcs = 16;
  __device__ __forceinline__ int ahead(int) { return 0; }
  __device__ __forceinline__ float p1(int, int) { return 0.f; }
  __device__ __forceinline__ float p2(int, int, float) { return 0.f; }
  __device__ __forceinline__ float p3(int kc, int u, float, float, float&) { return relu0(accP[2 * kc + (u >> 2)][u & 3]); }
  __device__ __forceinline__ int done(int kc, const float (&v)[8], const float (&)[8]) {
    if (rrow != nullptr && valid) x3h_store8(rrow, kc, kg, v, kcs);
    return 0;
  }
};
// layer-0 input of the radiance net: NPV k-chunks of PE(view dir) held in registers, then the feature row from global memory
template <int NPV>
struct XhPeRowSrc {
  static constexpr bool STORES = false;
  const float (&pe)[NPV * 8]; const float* row; int kg;
  f32x4 q[XH_RING][2];
  __device__ __forceinline__ int ahead(int kc) { if (kc >= NPV) x3h_load8(row, kc - NPV, kg, q[kc % XH_RING], 16); return kc >= NPV ? 2 : 0; }
  __device__ __forceinline__ float p1(int, int) { return 0.f; }
  __device__ __forceinline__ float p2(int, int, float) { return 0.f; }
  __device__ __forceinline__ float p3(int kc, int u, float, float, float&) { return kc < NPV ? pe[8 * (kc < NPV ? kc : 0) + u] : q[kc % XH_RING][u >> 2][u & 3]; }
  __device__ __forceinline__ int done(int, const float (&)[8], const float (&)[8]) { return 0; }
};
// a point-major row in global memory as B operand, optionally through ReLU (the light-mask head's input: relu(feature))
template <bool RELU>
struct XhRowSrc {
  static constexpr bool STORES = false;
  const float* row; int kg;
  f32x4 q[XH_RING][2];
  __device__ __forceinline__ int ahead(int kc) { x3h_load8(row, kc, kg, q[kc % XH_RING], 16); return 2; }
  __device__ __forceinline__ float p1(int, int) { return 0.f; }
  __device__ __forceinline__ float p2(int, int, float) { return 0.f; }
  __device__ __forceinline__ float p3(int kc, int u, float, float, float&) {
    const float x = q[kc % XH_RING][u >> 2][u & 3];
    return RELU ? relu0(x) : x;
  }
  __device__ __forceinline__ int done(int, const float (&)[8], const float (&)[8]) { return 0; }
};
// radiance backward: G(a_l) = (previous op's accumulators) where the saved activation r is positive; stores G(a_l)
template <int NT>
struct XhMaskSrc {
  static constexpr bool STORES = true;
  const f32x4 (&accP)[NT]; const float* rrow; float* grow; int kg; bool valid; int kcs = 16;
  f32x4 q[XH_RING][2];
  __device__ __forceinline__ int ahead(int kc) { x3h_load8(rrow, kc, kg, q[kc % XH_RING], kcs); return 2; }
  __device__ __forceinline__ float p1(int, int) { return 0.f; }
  __device__ __forceinline__ float p2(int, int, float) { return 0.f; }
  __device__ __forceinline__ float p3(int kc, int u, float, float, float&) { return q[kc % XH_RING][u >> 2][u & 3] > 0.f ? accP[2 * kc + (u >> 2)][u & 3] : 0.f; }
  __device__ __forceinline__ int done(int kc, const float (&v)[8], const float (&)[8]) {      // unconditional: padding points write their own rows
    x3h_store8(grow, kc, kg, v, kcs);
    return X3_COUNT_STORES ? 2 : 0;
  }
};

// D-layout registers (register 4*nt + r <-> feature 16*nt + 4*kg + r) <-> a saved row (kcs as above) / a point-major row (kcs = 16)
template <int NTK>
__device__ __forceinline__ void store_regs_h(float* __restrict__ row, int kg, bool valid, const float (&r)[NTK * 4], int kcs) {
  if (!valid) return;
#pragma unroll
  for (int nt = 0; nt < NTK; ++nt) stg4(row + nt * kcs + 4 * kg, f32x4{r[4 * nt], r[4 * nt + 1], r[4 * nt + 2], r[4 * nt + 3]});
}
template <int NTK>
__device__ __forceinline__ void store_tile_h(float* __restrict__ row, int kg, bool valid, const f32x4 (&t)[NTK]) {
  if (!valid) return;
#pragma unroll
  for (int nt = 0; nt < NTK; ++nt) stg4(row + 16 * nt + 4 * kg, t[nt]);
}
}  // namespace i2sdf
