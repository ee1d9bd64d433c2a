// K-outer bf16x3 op whose B-operand SOURCES come through an LDS ring filled by DMA (x3.h has the version with VGPR loads).
//
// Why.  The memory-heavy bf16x3 kernels (backward sweeps, d sdf/dx chain, radiance backward) re-read 1-2 KB of saved tensors
// per point and layer.  With ordinary `global_load`s next to the weight stream's LDS DMA, hipcc (ROCm 7.2) can only wait
// `vmcnt(0)` -- at every stage barrier AND at every first use of a loaded value (loads, stores and DMA share one counter and
// complete out of order with respect to each other, so no counted wait is provably safe for it) -- which drains every prefetched
// load a k-chunk after it was issued: memory time and matrix time add up (r1: 3.3 TB/s, MFMA pipe 22-36 % busy).
//
// Here EVERY vector-memory read of the wave is an LDS DMA (`global_load_lds`): the weight stages (shared by the workgroup) and
// the wave's own source pieces (a private ring of X3R_SLOTS slots; lane L writes and later reads bytes [16L, 16L+16) of a 1 KB
// piece, so the ring is a per-lane FIFO with no layout constraint).  Reads return in issue order, so "piece X has landed" is
// exactly "at most n reads issued after X are outstanding": a COUNTED `s_waitcnt vmcnt(n)` with n known at compile time (all
// loops are unrolled), and X3R_AHEAD k-chunks (~2 us of work) of source traffic stay in flight across the stage barriers, which
// are raw `s_barrier`s.  Stores also sit on the counter and may complete late; they only make the wait conservative (a store
// that is still outstanding can hold the wait a little longer, never let it pass early): X outstanding implies the n later
// reads are outstanding too, i.e. counter >= n + 1.
#pragma once
#include "x3.h"

namespace i2sdf {

#define X3R_AHEAD 3                                    // k-chunks of source pieces in flight ahead of their use
#define X3R_SLOTS 4                                    // ring slots per wave (AHEAD + 1: the slot being read is never a DMA target)
constexpr int X3R_SLOT_FLOATS = 1024;                  // 4 KB: up to four 1 KB pieces (64 lanes x 16 B) per k-chunk
constexpr int X3R_RING_FLOATS = X3R_SLOTS * X3R_SLOT_FLOATS;
constexpr int LDS_BYTES_R = LDS_BYTES + 4 * X3R_RING_FLOATS * 4;     // weight double buffer + 4 wave rings = 128 KB
constexpr int X3R_WDMA = STAGE_FLOATS / (WG_THREADS * 4);            // DMA instructions per wave per weight stage (8)

#define I2SDF_VMW(n) case n: asm volatile("s_waitcnt vmcnt(" #n ")" ::: "memory"); break;
// counted wait; `n` is a compile-time constant after unrolling (the switch folds to one instruction)
__device__ __forceinline__ void wait_vm(int n) {
  switch (n) {
    I2SDF_VMW(0) I2SDF_VMW(1) I2SDF_VMW(2) I2SDF_VMW(3) I2SDF_VMW(4) I2SDF_VMW(5) I2SDF_VMW(6) I2SDF_VMW(7)
    I2SDF_VMW(8) I2SDF_VMW(9) I2SDF_VMW(10) I2SDF_VMW(11) I2SDF_VMW(12) I2SDF_VMW(13) I2SDF_VMW(14) I2SDF_VMW(15)
    I2SDF_VMW(16) I2SDF_VMW(17) I2SDF_VMW(18) I2SDF_VMW(19) I2SDF_VMW(20) I2SDF_VMW(21) I2SDF_VMW(22) I2SDF_VMW(23)
    I2SDF_VMW(24) I2SDF_VMW(25) I2SDF_VMW(26) I2SDF_VMW(27) I2SDF_VMW(28) I2SDF_VMW(29) I2SDF_VMW(30) I2SDF_VMW(31)
    I2SDF_VMW(32) I2SDF_VMW(33) I2SDF_VMW(34) I2SDF_VMW(35) I2SDF_VMW(36) I2SDF_VMW(37) I2SDF_VMW(38) I2SDF_VMW(39)
    I2SDF_VMW(40) I2SDF_VMW(41) I2SDF_VMW(42) I2SDF_VMW(43) I2SDF_VMW(44) I2SDF_VMW(45) I2SDF_VMW(46) I2SDF_VMW(47)
    I2SDF_VMW(48) I2SDF_VMW(49) I2SDF_VMW(50) I2SDF_VMW(51) I2SDF_VMW(52) I2SDF_VMW(53) I2SDF_VMW(54) I2SDF_VMW(55)
    default: asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); break;
  }
}
#undef I2SDF_VMW

// one 1 KB piece: lane L's 16 bytes at `src` -> bytes [16L, 16L+16) of the piece at wave-uniform `dst`
__device__ __forceinline__ void x3r_dma16(const float* src, float* dst) {
  __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)src, (__attribute__((address_space(3))) void*)dst, 16, 0, 0);
}
// the two pieces of a point-major row that cover this lane's 8 reduction indices of k-chunk kc (x3_load8's addresses)
__device__ __forceinline__ void x3r_issue8(const float* row, int kc, int hi, float* slot, int piece, int kcs = 16) {
  x3r_dma16(row + kcs * kc + 4 * hi, slot + piece * 256);
  x3r_dma16(row + kcs * kc + 8 + 4 * hi, slot + (piece + 1) * 256);
}
__device__ __forceinline__ void x3r_fetch8(const float* slot, int piece, int lane, f32x4 (&q)[2]) {
  q[0] = *reinterpret_cast<const f32x4*>(slot + piece * 256 + lane * 4);
  q[1] = *reinterpret_cast<const f32x4*>(slot + (piece + 1) * 256 + lane * 4);
}

// ---------------------------------------------------------------------------------------------
// acc[NT] (+)= W * src, K-outer, sources through the ring.  Same stream layout / BIAS modes as dense_x3g.
//   Src: static constexpr int NLD (DMA pieces per k-chunk, 0..4), bool STORES;
//        void issue(kc, slot)            enqueue the NLD pieces of k-chunk kc
//        void fetch(kc, slot, lane)      ds_read them into the functor's registers (after the counted wait)
//        float value(kc, u, float& x2)   value u (0..7) of k-chunk kc; void done(kc, v, x2) stores
// `ring` = this wave's X3R_RING_FLOATS floats of LDS.  Precondition: no source piece of this wave is in flight, and no read
// was issued after the DMA of this op's first weight stage (true after any dense_x3r / advance()-based op).
// ---------------------------------------------------------------------------------------------
template <int NT, int KC16, int BIAS, class Src>
__device__ __forceinline__ void dense_x3r(WStream& ws, Src& src, f32x16 (&acc_io)[NT], float* ring, int tid) {
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
  constexpr int TOT = round_up(NB + NW, SC), NS = TOT / SC, PFP = 2, NLD = Src::NLD;
  const int lane = tid & 63;
  float v[8], vx[8];
  u32x4 bq[2][3];
  float sv[2][8], sx[2][8];
  int pk0 = -1, pk1 = -1;               // k-chunks whose stores are pending (compile-time after unrolling)
  // read accounting (all compile-time after unrolling): `issued` = reads this wave has issued since the op began;
  // sidx[kc % SLOTS] = value of `issued` right after k-chunk kc's pieces; widx = right after the weight stage awaited next
  int issued = 0, widx = 0;
  int sidx[X3R_SLOTS] = {0, 0, 0, 0};
  auto flush = [&]() __attribute__((always_inline)) {
    if (pk0 >= 0) { src.done(pk0, sv[0], sx[0]); pk0 = -1; }
    if (pk1 >= 0) { src.done(pk1, sv[1], sx[1]); pk1 = -1; }
  };
  auto issue_src = [&](int kc) __attribute__((always_inline)) {
    if (kc >= KC16 || NLD == 0) return;
    src.issue(kc, ring + (kc % X3R_SLOTS) * X3R_SLOT_FLOATS);
    issued += NLD;
    sidx[kc % X3R_SLOTS] = issued;
  };
  auto prep = [&](int kc, int u, u32x4 (&b)[3]) __attribute__((always_inline)) {
    if (kc >= KC16) return;
    if (u == 0) {
      issue_src(kc + X3R_AHEAD);
      if (NLD > 0) {
        wait_vm(issued - sidx[kc % X3R_SLOTS]);            // reads issued after k-chunk kc's pieces may stay in flight
        src.fetch(kc, ring + (kc % X3R_SLOTS) * X3R_SLOT_FLOATS, lane);
      }
    }
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
  for (int k0 = 0; k0 < X3R_AHEAD; ++k0) issue_src(k0);
  // the first weight stage was enqueued by the previous op; only the prologue pieces above were issued after it
  widx = 0;
#pragma unroll
  for (int u = 0; u < 12; ++u) prep(0, u, bq[0]);
#pragma unroll
  for (int s = 0; s < NS; ++s) {
    // ---- stage barrier: my DMA pieces of this weight stage have landed (counted), everybody is done with the other buffer
    wait_vm(issued - widx);
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");                       // no LDS read of the new stage may be hoisted above the barrier
    const u32x4* cur = reinterpret_cast<const u32x4*>(ws.lds + ws.cur * STAGE_FLOATS) + lane;
    flush();
#pragma unroll
    for (int j = 0; j < SC; ++j) {
      const int c = s * SC + j;
      if (c < NB && BIAS == 1) {
        const int nt = c / 4, q = c % 4;
        const f32x4 b = __builtin_bit_cast(f32x4, cur[j * 64]);
        acc[nt][4 * q + 0] = b.x; acc[nt][4 * q + 1] = b.y; acc[nt][4 * q + 2] = b.z; acc[nt][4 * q + 3] = b.w;
      }
    }
    const int p0 = (s * SC < NB) ? ((NB - s * SC < SC) ? (NB - s * SC) / 2 : SC / 2) : 0;
    const int p1 = (NB + NW - s * SC < SC) ? ((NB + NW - s * SC > 0) ? (NB + NW - s * SC) / 2 : 0) : SC / 2;
    u32x4 wr[PFP][2];
#pragma unroll
    for (int i = 0; i < PFP; ++i)
      if (p0 + i < p1) { wr[i][0] = cur[(2 * (p0 + i)) * 64]; wr[i][1] = cur[(2 * (p0 + i) + 1) * 64]; }
    __builtin_amdgcn_sched_barrier(0);
    bool wissued = false;
    auto issue_w = [&]() __attribute__((always_inline)) {
      // next weight stage into the buffer everybody just left.  At the end of the kernel's stream nothing is fetched; the
      // accounting below still counts the pieces, so every later wait of this op is made safe by a full drain here.
      if (ws.left > 0) { ws.issue(ws.lds + (ws.cur ^ 1) * STAGE_FLOATS, tid); --ws.left; }
      else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      ws.cur ^= 1;
      issued += X3R_WDMA;
      widx = issued;
      wissued = true;
    };
#pragma unroll
    for (int jp = 0; jp < SC / 2; ++jp) {
      if (jp >= p0 && jp < p1) {
        const int w = (s * SC + 2 * jp - NB) / 2;
        const int kc = w / PPK, g = (w / 3) % G, sp = w % 3, nt = 2 * g;
        const u32x4 a0 = wr[(jp - p0) % PFP][0], a1 = wr[(jp - p0) % PFP][1];
        if (jp + PFP < p1) {
          wr[(jp - p0) % PFP][0] = cur[(2 * (jp + PFP)) * 64];
          wr[(jp - p0) % PFP][1] = cur[(2 * (jp + PFP) + 1) * 64];
        }
        const u32x4 (&b)[3] = bq[kc & 1];
        acc[nt] = mfma_bf16(a0, b[0], acc[nt]);
        acc[nt + 1] = mfma_bf16(a1, b[0], acc[nt + 1]);
        if (sp < 2) {
          acc[nt] = mfma_bf16(a0, b[1], acc[nt]);
          acc[nt + 1] = mfma_bf16(a1, b[1], acc[nt + 1]);
        }
        if (sp == 0) {
          acc[nt] = mfma_bf16(a0, b[2], acc[nt]);
          acc[nt + 1] = mfma_bf16(a1, b[2], acc[nt + 1]);
        }
        if (!wissued) issue_w();
        {
          const int pi = w % PPK;
#pragma unroll
          for (int u = 0; u < 12; ++u)
            if (u * PPK / 12 == pi) prep(kc + 1, u, bq[(kc + 1) & 1]);
        }
        __builtin_amdgcn_sched_barrier(0);
      }
    }
    if (!wissued) issue_w();
  }
  flush();
#pragma unroll
  for (int nt = 0; nt < NT; ++nt) acc_io[nt] = acc[nt];
}

// apply a ring source to every k-chunk without a consuming op (the last epilogue of a chain)
template <int KC16, class Src>
__device__ __forceinline__ void x3r_drain(Src& src, float* ring, int tid) {
  constexpr int NLD = Src::NLD;
  const int lane = tid & 63;
  float v[8], vx[8];
  int issued = 0;
  int sidx[X3R_SLOTS] = {0, 0, 0, 0};
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");       // end of the stream: the look-ahead weight stage (if any) is not used
#pragma unroll
  for (int k0 = 0; k0 < X3R_AHEAD; ++k0)
    if (k0 < KC16 && NLD > 0) { src.issue(k0, ring + (k0 % X3R_SLOTS) * X3R_SLOT_FLOATS); issued += NLD; sidx[k0 % X3R_SLOTS] = issued; }
#pragma unroll
  for (int kc = 0; kc < KC16; ++kc) {
    if (kc + X3R_AHEAD < KC16 && NLD > 0) {
      const int k2 = kc + X3R_AHEAD;
      src.issue(k2, ring + (k2 % X3R_SLOTS) * X3R_SLOT_FLOATS); issued += NLD; sidx[k2 % X3R_SLOTS] = issued;
    }
    if (NLD > 0) {
      wait_vm(issued - sidx[kc % X3R_SLOTS]);
      src.fetch(kc, ring + (kc % X3R_SLOTS) * X3R_SLOT_FLOATS, lane);
    }
#pragma unroll
    for (int u = 0; u < 8; ++u) v[u] = src.value(kc, u, vx[u]);
    src.done(kc, v, vx);
  }
}

// ---- ring sources ----------------------------------------------------------------------------------------------------
// values held in registers (no memory traffic)
template <int NREG>
struct X3rRegSrc {
  static constexpr int NLD = 0;
  static constexpr bool STORES = false;
  const float (&r)[NREG];
  __device__ __forceinline__ void issue(int, float*) {}
  __device__ __forceinline__ void fetch(int, const float*, int) {}
  __device__ __forceinline__ float value(int kc, int u, float&) { return r[8 * kc + u]; }
  __device__ __forceinline__ void done(int, const float (&)[8], const float (&)[8]) {}
};
// reverse chain (d sdf/dx): abar = (previous op's accumulators) * sigma(h), h from the saved tensor; stores abar
template <int NT>
struct X3rRevSrc {
  static constexpr int NLD = 2;
  static constexpr bool STORES = true;
  const f32x16 (&accP)[NT]; const float* hrow; float* abrow; int hi; bool valid; int kcs = 16;
  f32x4 hq[2];
  __device__ __forceinline__ void issue(int kc, float* slot) { x3r_issue8(hrow, kc, hi, slot, 0, kcs); }
  __device__ __forceinline__ void fetch(int, const float* slot, int lane) { x3r_fetch8(slot, 0, lane, hq); }
  __device__ __forceinline__ float value(int kc, int u, float&) { return accP[kc >> 1][8 * (kc & 1) + u] * sp_sigma_from_h(hq[u >> 2][u & 3]); }
  __device__ __forceinline__ void done(int kc, const float (&v)[8], const float (&)[8]) {
    if (abrow != nullptr && valid) x3_store8(abrow, kc, hi, v, kcs);
  }
};
// backward sweep 1 (see X3Sweep1Src)
template <int NT, int KACC, int NREG>
struct X3rSweep1Src {
  static constexpr int NLD = KACC > 0 ? 4 : 0;
  static constexpr bool STORES = true;
  const f32x16 (&accP)[NT]; const float (&tailreg)[NREG];
  const float* hrow; const float* arow; float* g2row; float* gurow; int hi; bool valid; int kcs = 16;
  f32x4 hq[2], aq[2];
  __device__ __forceinline__ void issue(int kc, float* slot) {
    // k-chunks >= KACC come from registers; their (unused) pieces re-read the last row chunk so that every k-chunk issues NLD reads
    const int kk = kc < KACC ? kc : KACC - 1;
    x3r_issue8(hrow, kk, hi, slot, 0, kcs);
    x3r_issue8(arow, kk, hi, slot, 2, kcs);
  }
  __device__ __forceinline__ void fetch(int kc, const float* slot, int lane) {
    if (kc < KACC) { x3r_fetch8(slot, 0, lane, hq); x3r_fetch8(slot, 2, lane, aq); }
  }
  __device__ __forceinline__ float value(int kc, int u, float& g2) {
    g2 = 0.f;
    if (kc >= KACC) return tailreg[8 * (kc - KACC < 0 ? 0 : kc - KACC) + u];
    const float ga = accP[(kc >> 1) < NT ? (kc >> 1) : 0][8 * (kc & 1) + u];
    const float sg = sp_sigma_from_h(hq[u >> 2][u & 3]);
    g2 = ga * aq[u >> 2][u & 3] * (100.f * (1.0f - sg));
    return ga * sg;
  }
  __device__ __forceinline__ void done(int kc, const float (&v)[8], const float (&g2)[8]) {
    if (kc < KACC && valid) { x3_store8(gurow, kc, hi, v, kcs); x3_store8(g2row, kc, hi, g2, kcs); }
  }
};
// backward sweep 2 (see X3Sweep2Src); TOP: + sbar * w_sdf, w_sdf read from the packed buffer (third DMA pair)
template <int NT, bool TOP>
struct X3rSweep2Src {
  static constexpr int NLD = 4;
  static constexpr bool STORES = true;
  const f32x16 (&accP)[NT]; const float* hrow; const float* g2row; float* grow; int hi; bool valid;
  float sb; const float* wsdf;        // (TOP form unused by the ring kernels: sbar w_sdf is added to the upstream before the op)
  int kcs = 16;
  f32x4 hq[2], gq[2], wq[2];
  __device__ __forceinline__ void issue(int kc, float* slot) {
    x3r_issue8(hrow, kc, hi, slot, 0, kcs);
    x3r_issue8(g2row, kc, hi, slot, 2, kcs);
  }
  __device__ __forceinline__ void fetch(int kc, const float* slot, int lane) {
    x3r_fetch8(slot, 0, lane, hq); x3r_fetch8(slot, 2, lane, gq);
  }
  __device__ __forceinline__ float value(int kc, int u, float&) {
    float x = accP[kc >> 1][8 * (kc & 1) + u];
    if (TOP) x = fmaf(sb, wq[u >> 2][u & 3], x);
    return fmaf(x, sp_sigma_from_h(hq[u >> 2][u & 3]), gq[u >> 2][u & 3]);
  }
  __device__ __forceinline__ void done(int kc, const float (&v)[8], const float (&)[8]) {
    if (valid) x3_store8(grow, kc, hi, v, kcs);
  }
};
// a point-major row in global memory (or zeros) as B operand
struct X3rRowSrc {
  static constexpr int NLD = 2;
  static constexpr bool STORES = false;
  const float* row; int hi; bool on;
  f32x4 q[2];
  __device__ __forceinline__ void issue(int kc, float* slot) { x3r_issue8(row, kc, hi, slot, 0); }
  __device__ __forceinline__ void fetch(int, const float* slot, int lane) {
    x3r_fetch8(slot, 0, lane, q);
    if (!on) { q[0] = f32x4{0.f, 0.f, 0.f, 0.f}; q[1] = f32x4{0.f, 0.f, 0.f, 0.f}; }
  }
  __device__ __forceinline__ float value(int, int u, float&) { return q[u >> 2][u & 3]; }
  __device__ __forceinline__ void done(int, const float (&)[8], const float (&)[8]) {}
};
// radiance backward: G(a_l) = (previous op's accumulators) where the saved activation r is positive; stores G(a_l)
template <int NT>
struct X3rMaskSrc {
  static constexpr int NLD = 2;
  static constexpr bool STORES = true;
  const f32x16 (&accP)[NT]; const float* rrow; float* grow; int hi; bool valid; int kcs = 16;
  f32x4 q[2];
  __device__ __forceinline__ void issue(int kc, float* slot) { x3r_issue8(rrow, kc, hi, slot, 0, kcs); }
  __device__ __forceinline__ void fetch(int, const float* slot, int lane) { x3r_fetch8(slot, 0, lane, q); }
  __device__ __forceinline__ float value(int kc, int u, float&) { return q[u >> 2][u & 3] > 0.f ? accP[kc >> 1][8 * (kc & 1) + u] : 0.f; }
  __device__ __forceinline__ void done(int kc, const float (&v)[8], const float (&)[8]) {
    if (valid) x3_store8(grow, kc, hi, v, kcs);
  }
};

}  // namespace i2sdf
