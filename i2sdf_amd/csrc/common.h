// Shared device-side building blocks for the MI355X (gfx950 / CDNA4) kernels.
//
// Execution model used by every MLP kernel in this library
// ---------------------------------------------------------
//   * one WAVE (64 lanes) owns 32 points for a whole pass through a network; a workgroup is 4 waves
//     (one per SIMD, up to 512 unified VGPR+AGPR each);
//   * activations never leave registers between layers.  A layer is out^T[N x 32] = W[N x K] * in^T[K x 32]
//     computed with v_mfma_f32_32x32x2_f32 (exact fp32, 157 TF/s peak): A operand = W, B operand = activations.
//     The MFMA C/D layout (lane = point m + 32*hi, register r <-> row (r&3) + 8*(r>>2) + 4*hi) is, after a fixed
//     permutation of the reduction index, exactly the B-operand layout of the next layer, so the output
//     registers of layer l (after the activation function) are fed straight back as B operands of layer l+1;
//   * weights are pre-packed ("streams") in the exact order the MFMAs consume them: one CHUNK = 64 lanes x 16 B
//     = the A operands of 4 consecutive MFMAs.  A workgroup DMAs the stream global->LDS (global_load_lds,
//     no VGPR round trip) in 32 KB stages, double buffered, one barrier per stage; the 4 waves share it.
//
// Register <-> index maps (hi = lane>>5, r = 4q+t):
//   D-layout tile nt, reg r   <->  feature index 32*nt + 8*q + 4*hi + t
//   chunk kc, element t        <->  reduction index 8*kc + 4*hi + t         (so chunk kc = 4*nt + q)
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

namespace i2sdf {

// 16-byte accesses to the SAVED per-point tensors (written once by one kernel, read once by a later one, hundreds of MB in between):
// I2SDF_NT = 1 marks them non-temporal, so that they do not displace the packed weight streams (4-5 MB, re-read by every workgroup)
// from the 4 MB L2 of an XCD.
#ifndef I2SDF_NT
#define I2SDF_NT 1      // measured (round 4, A/B in one run): step -0.03 ms, radiance forward / backward -3 / -6 %, sweeps unchanged
#endif
#ifndef I2SDF_NT_LD
#define I2SDF_NT_LD I2SDF_NT      // loads and stores separately (round 5, scripts/ubench/hbm_mix.hip)
#endif
#ifndef I2SDF_NT_ST
#define I2SDF_NT_ST I2SDF_NT
#endif
__device__ __forceinline__ f32x4 ldg4(const float* p) {
#if I2SDF_NT_LD
  return __builtin_nontemporal_load(reinterpret_cast<const f32x4*>(p));
#else
  return *reinterpret_cast<const f32x4*>(p);
#endif
}
__device__ __forceinline__ void stg4(float* p, f32x4 v) {
#if I2SDF_NT_ST
  __builtin_nontemporal_store(v, reinterpret_cast<f32x4*>(p));
#else
  *reinterpret_cast<f32x4*>(p) = v;
#endif
}

constexpr int SC = 32;                    // chunks per LDS stage
constexpr int CHUNK_FLOATS = 256;         // 64 lanes x 4 floats
constexpr int STAGE_FLOATS = SC * CHUNK_FLOATS;   // 32 KB
constexpr int LDS_BYTES = 2 * STAGE_FLOATS * 4;   // double buffer
constexpr int PF = 4;                     // ds_read prefetch depth (chunks)
constexpr int WG_THREADS = 256;
constexpr int PTS_PER_WAVE = 32;
constexpr int PTS_PER_WG = 128;

__host__ __device__ constexpr int round_up(int a, int b) { return (a + b - 1) / b * b; }
__host__ __device__ constexpr int cdiv(int a, int b) { return (a + b - 1) / b; }

// chunks of one dense op in a stream: [NT*4 bias chunks][NT*KC weight chunks], padded to whole stages
__host__ __device__ constexpr int op_chunks(int NT, int KC) { return round_up(NT * 4 + NT * KC, SC); }
// a row-vector op: [KC weight chunks][1 scalar chunk]
__host__ __device__ constexpr int rowvec_chunks(int KC, int nrows) { return round_up(nrows * KC + 1, SC); }

#define I2SDF_SGB(mask, n) __builtin_amdgcn_sched_group_barrier(mask, n, 0)
#define I2SDF_MASK_MFMA 0x008
#define I2SDF_MASK_DSREAD 0x100
#define I2SDF_MASK_VALU 0x002

__device__ __forceinline__ f32x16 mfma(float a, float b, f32x16 c) {
  return __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, c, 0, 0, 0);
}

// ---------------------------------------------------------------------------------------------
// Weight stream: linear walk over a packed buffer, global -> LDS by DMA, double buffered.
// ---------------------------------------------------------------------------------------------
// NTHR = threads of the workgroup, SCS = chunks (KB) per LDS stage
template <int NTHR, int SCS = SC>
struct WStreamT {
  static constexpr int STG = SCS * CHUNK_FLOATS;      // floats per stage
  // The DMA is `buffer_load_dwordx4 ... lds` (address = descriptor base + scalar byte offset + per-lane offset tid*16), not
  // `global_load_lds`: hipcc counts the global form as a FLAT access (it may touch LDS out of order), and while one is pending every
  // wait for an LDS READ result becomes s_waitcnt lgkmcnt(0) -- also for reads issued a few cycles earlier, so the read-ahead of the
  // MFMA loops was worth nothing.  Behind the buffer form the waits carry exact counts (lgkmcnt(2), (4) ...).
  __amdgpu_buffer_rsrc_t rs;   // the packed stream this kernel walks, from its first stage
  unsigned goff;      // byte offset (from the descriptor base) of the next stage to fetch
  float* lds;         // two stage buffers
  int cur;            // buffer that the NEXT advance() returns
  int left;           // stages still to be fetched
  int wv;             // this wave's index in the workgroup as a scalar (M0, the LDS address of a piece, is computed on the scalar unit)

  __device__ __forceinline__ void piece(float* dst_stage, unsigned stage_off, int i, int tid) {
    float* d = dst_stage + wv * 256 + i * NTHR * 4;     // wave-uniform LDS base; the hardware adds lane*16 B
    __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, (__attribute__((address_space(3))) void*)d, 16, tid * 16, stage_off + i * (NTHR * 16), 0, 0);
  }
  __device__ __forceinline__ void issue(float* dst, int tid) {
#pragma unroll
    for (int i = 0; i < STG / (NTHR * 4); ++i) piece(dst, goff, i, tid);
    goff += STG * 4;
  }
  // n_stages = total stages this kernel will consume from `base`
  __device__ __forceinline__ void begin(const float* base, float* lds_, int n_stages, int tid) {
    rs = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(base), 0, 0x7ffffff0, 0x00020000);
    goff = 0u; lds = lds_; cur = 0; left = n_stages;
    wv = __builtin_amdgcn_readfirstlane(tid >> 6);
    if (left > 0) { issue(lds, tid); --left; }
  }
  // Returns the LDS buffer holding the next stage.  All 4 waves must call this in lock step.
  __device__ __forceinline__ const float* advance(int tid) {
    // (a) my DMA pieces of this stage have landed: hipcc usually drains vmcnt in front of the barrier by itself, but it
    // tracks LDS DMA per address and was seen to leave the wait out (wgrad.hip) -- the protocol must not depend on that
    __builtin_amdgcn_s_waitcnt(0x0F70);      // vmcnt(0)
    __syncthreads();         // (b) every wave's did, (c) every wave finished reading the other buffer
    const float* ret = lds + cur * STG;
    if (left > 0) { issue(lds + (cur ^ 1) * STG, tid); --left; }
    cur ^= 1;
    return ret;
  }
  // piecewise form of advance_issue(): the NPIECE 4 KB pieces of the following stage one at a time (i = 0..NPIECE-1, each wave moves
  // 1 KB per piece), then advance_done() -- a piece costs the wave 60-180 cycles of issue (MI355X_MICROARCH.md), so the K-outer
  // bf16x3 ops put ONE piece behind each group of MFMAs instead of all eight behind the first group
  static constexpr int NPIECE = STG / (NTHR * 4);
  // Branch-free on purpose: a scalar branch around the DMA would cut the surrounding MFMA stream into basic blocks, and the
  // scheduling fences (sched_barrier) that deal the VALU work into the MFMA shadows only act inside one block.  When no stage is left
  // the piece re-reads the stage fetched last (valid memory) into the buffer nobody reads any more.
  __device__ __forceinline__ void issue_piece(int i, int tid) {
    piece(lds + (cur ^ 1) * STG, left > 0 ? goff : goff - STG * 4u, i, tid);
  }
  __device__ __forceinline__ void advance_done() {
    if (left > 0) { goff += STG * 4; --left; }
    cur ^= 1;
  }
  // split form: barrier now, DMA of the following stage a little later (from inside the MFMA stream)
  __device__ __forceinline__ const float* advance_barrier() {
    __builtin_amdgcn_s_waitcnt(0x0F70);      // vmcnt(0), see advance()
    __syncthreads();
    return lds + cur * STG;
  }
  // Counted form: only what is OLDER than this stage's last DMA piece has to be complete -- `young` = number of vector-memory instructions that
  // the wave has issued since that piece UNCONDITIONALLY (a compile-time value after unrolling; vmcnt retires in order on gfx9, so vmcnt(young)
  // implies that the pieces have landed; a smaller `young` than the truth only waits longer).  The source loads issued ahead of their use
  // (x3.h) then fly across the stage barrier instead of being drained at it.  scripts/ubench/mfma_paced.hip (profiles/r5_mfma_paced.txt).
  template <int N> static __device__ __forceinline__ void wait_barrier() {
    // one asm statement with a memory clobber: no LDS read of the new stage moves above it (__syncthreads() would add its own vmcnt(0)).
    // lgkmcnt(0): every wave must have FINISHED reading the other stage buffer before the next DMA overwrites it.  The reads of a stage are
    // all consumed by MFMAs in front of the barrier, so the compiler has always had the wait there by itself and this one is free -- but
    // the protocol must not rest on where the compiler puts it (MFMAs may legally sink below an asm statement)
    asm volatile("s_waitcnt vmcnt(%0) lgkmcnt(0)\n\ts_barrier" ::"n"(N) : "memory");
  }
  __device__ __forceinline__ const float* advance_barrier_young(int young) {
    switch (young < 0 ? 0 : young > 16 ? 16 : young) {
      case 0: wait_barrier<0>(); break; case 1: wait_barrier<1>(); break; case 2: wait_barrier<2>(); break; case 3: wait_barrier<3>(); break;
      case 4: wait_barrier<4>(); break; case 5: wait_barrier<5>(); break; case 6: wait_barrier<6>(); break; case 7: wait_barrier<7>(); break;
      case 8: wait_barrier<8>(); break; case 9: wait_barrier<9>(); break; case 10: wait_barrier<10>(); break; case 11: wait_barrier<11>(); break;
      case 12: wait_barrier<12>(); break; case 13: wait_barrier<13>(); break; case 14: wait_barrier<14>(); break; case 15: wait_barrier<15>(); break;
      default: wait_barrier<16>(); break;
    }
    return lds + cur * STG;
  }
  __device__ __forceinline__ void advance_issue(int tid) {
    if (left > 0) { issue(lds + (cur ^ 1) * STG, tid); --left; }
    cur ^= 1;
  }
  // skip `n` stages without computing (still in lock step)
  __device__ __forceinline__ void skip(int n, int tid) {
    for (int i = 0; i < n; ++i) (void)advance(tid);
  }
};
using WStream = WStreamT<WG_THREADS>;      // 4 waves per workgroup (every kernel but the 16-point-wave family, x3h.h)

// ---------------------------------------------------------------------------------------------
// One dense op:  acc[NT] (+)= bias + W * in     (32 points per wave, all in registers)
//   stream layout of the op: [NT*4 bias chunks][NT*KC weight chunks] padded to whole stages
//   MODE 0: acc = bias + W*in     MODE 1: acc = W*in (bias chunks skipped)    MODE 2: acc += W*in
// ---------------------------------------------------------------------------------------------
template <int NT, int KC, int MODE>
__device__ __forceinline__ void dense_op(WStream& ws, const float (&in)[KC * 4], f32x16 (&acc)[NT], int tid) {
  constexpr int NB = NT * 4, NW = NT * KC, TOT = op_chunks(NT, KC), NS = TOT / SC;
  const int lane = tid & 63;
#pragma unroll
  for (int s = 0; s < NS; ++s) {
    const f32x4* cur = reinterpret_cast<const f32x4*>(ws.advance(tid)) + lane;
    // ---- bias chunks of this stage
#pragma unroll
    for (int j = 0; j < SC; ++j) {
      const int c = s * SC + j;
      if (c < NB) {
        const int nt = c / 4, q = c % 4;
        if (MODE == 0) {
          f32x4 b = cur[j * 64];
          acc[nt][4 * q + 0] = b.x; acc[nt][4 * q + 1] = b.y; acc[nt][4 * q + 2] = b.z; acc[nt][4 * q + 3] = b.w;
        } else if (MODE == 1) {
          acc[nt][4 * q + 0] = 0.f; acc[nt][4 * q + 1] = 0.f; acc[nt][4 * q + 2] = 0.f; acc[nt][4 * q + 3] = 0.f;
        }
      }
    }
    // ---- weight chunks of this stage: [j0, j1) within the stage
    constexpr int dummy = 0; (void)dummy;
    const int j0 = (s * SC < NB) ? ((NB - s * SC < SC) ? NB - s * SC : SC) : 0;
    const int j1 = (NB + NW - s * SC < SC) ? ((NB + NW - s * SC > 0) ? NB + NW - s * SC : 0) : SC;
    f32x4 ab[PF];
#pragma unroll
    for (int i = 0; i < PF; ++i)
      if (j0 + i < j1) ab[i] = cur[(j0 + i) * 64];
#pragma unroll
    for (int j = 0; j < SC; ++j) {
      if (j >= j0 && j < j1) {
        const int w = s * SC + j - NB;
        const int nt = w / KC, kc = w % KC;
        f32x4 a = ab[(j - j0) % PF];
        if (j + PF < j1) ab[(j - j0) % PF] = cur[(j + PF) * 64];
        acc[nt] = mfma(a.x, in[kc * 4 + 0], acc[nt]);
        acc[nt] = mfma(a.y, in[kc * 4 + 1], acc[nt]);
        acc[nt] = mfma(a.z, in[kc * 4 + 2], acc[nt]);
        acc[nt] = mfma(a.w, in[kc * 4 + 3], acc[nt]);
      }
    }
    // pin the software pipeline: PF reads up front, then {4 MFMA, 1 read} per chunk
#pragma unroll
    for (int i = 0; i < PF; ++i)
      if (j0 + i < j1) I2SDF_SGB(I2SDF_MASK_DSREAD, 1);
#pragma unroll
    for (int j = 0; j < SC; ++j) {
      if (j >= j0 && j < j1) {
        I2SDF_SGB(I2SDF_MASK_MFMA, 1);
        if (j + PF < j1) I2SDF_SGB(I2SDF_MASK_DSREAD, 1);
        I2SDF_SGB(I2SDF_MASK_MFMA, 3);
      }
    }
  }
}

// ---------------------------------------------------------------------------------------------
// Dense op with an INTERLEAVED per-tile epilogue.
//   stream layout: [NB bias chunks (NB = NT*4 or 0)][NT*KC weight chunks] padded to whole stages
//   MODE 0: acc = bias + W*in   MODE 1: acc = W*in   MODE 2: acc += W*in
// Output tile nt is final once its last weight chunk has been consumed.  Its epilogue (activation function,
// products with saved tensors, global stores -- `epi.apply(nt, acc[nt])`, in place on the accumulator) is emitted at
// the start of the NEXT stage, in the same scheduling region as that stage's MFMAs, and sched_group_barrier
// spreads its VALU work into the 64-cycle gaps between MFMAs; the global loads it needs (`epi.prefetch(nt)`) are
// issued one stage earlier.  Only the epilogue of the tiles finishing in the last stage is exposed.
//   Epi: struct with  void prefetch(int nt);  void apply(int nt, f32x16& acc);   (nt is a constant after unrolling)
// ---------------------------------------------------------------------------------------------
struct NoEpi {
  __device__ __forceinline__ void prefetch(int) {}
  __device__ __forceinline__ void elem(int, f32x16&, int) {}
  __device__ __forceinline__ void apply(int, f32x16&) {}
};

template <int NT, int KC, int NB, int MODE, int VALU_PER_CHUNK, class Epi>
__device__ __forceinline__ void dense_op_epi(WStream& ws, const float (&in)[KC * 4], f32x16 (&acc)[NT], Epi& epi, int tid) {
  static_assert(NB == 0 || NB == NT * 4, "bias chunks");
  constexpr int NW = NT * KC, TOT = round_up(NB + NW, SC), NS = TOT / SC;
  const int lane = tid & 63;
  if (NB == 0 && MODE == 1) {
#pragma unroll
    for (int nt = 0; nt < NT; ++nt)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[nt][r] = 0.f;
  }
#pragma unroll
  for (int s = 0; s < NS; ++s) {
    // tile nt is complete before stage s  <=>  NB + (nt+1)*KC <= s*SC
    // tiles that completed during the previous stage get their epilogue during this stage.  If it is exactly one tile,
    // its 16 elements are dealt out between the stage's first 16 chunks (one element after each chunk's MFMAs, where the
    // 64-cycle MFMA gaps absorb the VALU work); otherwise the epilogues run in front of the MFMAs.
    int pend = -1, npend = 0;
#pragma unroll
    for (int nt = 0; nt < NT; ++nt) {
      const int last = NB + (nt + 1) * KC;
      if (s > 0 && last > (s - 1) * SC && last <= s * SC) { pend = nt; ++npend; }
    }
    const int j0 = (s * SC < NB) ? ((NB - s * SC < SC) ? NB - s * SC : SC) : 0;
    const int j1 = (NB + NW - s * SC < SC) ? ((NB + NW - s * SC > 0) ? NB + NW - s * SC : 0) : SC;
    const bool deal = (npend == 1) && (j1 - j0 >= 32);
    const f32x4* cur;
    if (deal) {
      cur = reinterpret_cast<const f32x4*>(ws.advance_barrier()) + lane;      // DMA + operand loads follow inside the chunk stream
    } else {
      cur = reinterpret_cast<const f32x4*>(ws.advance(tid)) + lane;
      // tiles completing during this stage: issue their operand loads now
#pragma unroll
      for (int nt = 0; nt < NT; ++nt) {
        const int last = NB + (nt + 1) * KC;
        if (last > s * SC && last <= (s + 1) * SC) epi.prefetch(nt);
      }
    }
    if (!deal) {
#pragma unroll
      for (int nt = 0; nt < NT; ++nt) {
        const int last = NB + (nt + 1) * KC;
        if (s > 0 && last > (s - 1) * SC && last <= s * SC) epi.apply(nt, acc[nt]);
      }
    }
    // ---- bias chunks of this stage
#pragma unroll
    for (int j = 0; j < SC; ++j) {
      const int c = s * SC + j;
      if (c < NB) {
        const int nt = c / 4, q = c % 4;
        if (MODE == 0) {
          f32x4 b = cur[j * 64];
          acc[nt][4 * q + 0] = b.x; acc[nt][4 * q + 1] = b.y; acc[nt][4 * q + 2] = b.z; acc[nt][4 * q + 3] = b.w;
        } else if (MODE == 1) {
          acc[nt][4 * q + 0] = 0.f; acc[nt][4 * q + 1] = 0.f; acc[nt][4 * q + 2] = 0.f; acc[nt][4 * q + 3] = 0.f;
        }
      }
    }
    f32x4 ab[PF];
#pragma unroll
    for (int i = 0; i < PF; ++i)
      if (j0 + i < j1) ab[i] = cur[(j0 + i) * 64];
    if (deal) __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int j = 0; j < SC; ++j) {
      if (j >= j0 && j < j1) {
        const int w = s * SC + j - NB;
        const int nt = w / KC, kc = w % KC;
        f32x4 a = ab[(j - j0) % PF];
        if (j + PF < j1) ab[(j - j0) % PF] = cur[(j + PF) * 64];
        acc[nt] = mfma(a.x, in[kc * 4 + 0], acc[nt]);
        acc[nt] = mfma(a.y, in[kc * 4 + 1], acc[nt]);
        acc[nt] = mfma(a.z, in[kc * 4 + 2], acc[nt]);
        acc[nt] = mfma(a.w, in[kc * 4 + 3], acc[nt]);
        if (deal) {
          const int jj = j - j0;
          if (jj == 1) ws.advance_issue(tid);                    // next stage's DMA, in the shadow of this chunk's MFMAs
          if (jj == 3) {
#pragma unroll
            for (int n2 = 0; n2 < NT; ++n2) {
              const int last = NB + (n2 + 1) * KC;
              if (last > s * SC && last <= (s + 1) * SC) epi.prefetch(n2);
            }
          }
          // one element per chunk during the FIRST half of the stage: its stores are then >= 16 chunks (4k cycles) old when
          // the next stage's barrier drains vmcnt
          if (jj < 16) epi.elem(pend, acc[pend < 0 ? 0 : pend], jj);
          // per-chunk scheduling region: {MFMA, ds_read, 3 MFMA} then this chunk's share of the epilogue
          I2SDF_SGB(I2SDF_MASK_MFMA, 1);
          if (j + PF < j1) I2SDF_SGB(I2SDF_MASK_DSREAD, 1);
          I2SDF_SGB(I2SDF_MASK_MFMA, 3);
          __builtin_amdgcn_sched_barrier(0);
        }
      }
    }
    if (!deal) {
#pragma unroll
      for (int i = 0; i < PF; ++i)
        if (j0 + i < j1) I2SDF_SGB(I2SDF_MASK_DSREAD, 1);
#pragma unroll
      for (int j = 0; j < SC; ++j) {
        if (j >= j0 && j < j1) {
          I2SDF_SGB(I2SDF_MASK_MFMA, 1);
          if (j + PF < j1) I2SDF_SGB(I2SDF_MASK_DSREAD, 1);
          I2SDF_SGB(I2SDF_MASK_MFMA, 3);
        }
      }
    }
  }
  // tiles that completed in the last stage (their loads were issued at its start)
#pragma unroll
  for (int nt = 0; nt < NT; ++nt) {
    const int last = NB + (nt + 1) * KC;
    if (last > (NS - 1) * SC) epi.apply(nt, acc[nt]);
  }
}

// Row-vector op: out[row] = sum_k w_row[k] * in[k] (+ scalar), VALU dot product + one cross-half exchange.
// stream layout: [NROWS*KC chunks: element t of chunk (row,kc) = w_row[8kc+4hi+t]][1 chunk: {s0,s1,s2,s3}]
template <int NROWS, int KC>
__device__ __forceinline__ void rowvec_op(WStream& ws, const float (&in)[KC * 4], float (&out)[NROWS], int tid) {
  constexpr int TOT = rowvec_chunks(KC, NROWS), NS = TOT / SC, NW = NROWS * KC;
  const int lane = tid & 63;
  float part[NROWS];
#pragma unroll
  for (int r = 0; r < NROWS; ++r) part[r] = 0.f;
  f32x4 sc = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
  for (int s = 0; s < NS; ++s) {
    const f32x4* cur = reinterpret_cast<const f32x4*>(ws.advance(tid)) + lane;
#pragma unroll
    for (int j = 0; j < SC; ++j) {
      const int c = s * SC + j;
      if (c < NW) {
        const int row = c / KC, kc = c % KC;
        f32x4 w = cur[j * 64];
        part[row] = fmaf(w.x, in[kc * 4 + 0], part[row]);
        part[row] = fmaf(w.y, in[kc * 4 + 1], part[row]);
        part[row] = fmaf(w.z, in[kc * 4 + 2], part[row]);
        part[row] = fmaf(w.w, in[kc * 4 + 3], part[row]);
      } else if (c == NW) {
        sc = cur[j * 64];
      }
    }
  }
#pragma unroll
  for (int r = 0; r < NROWS; ++r) {
    float v = part[r] + __shfl_xor(part[r], 32);
    out[r] = v + (r == 0 ? sc.x : r == 1 ? sc.y : r == 2 ? sc.z : sc.w);
  }
}

// ---------------------------------------------------------------------------------------------
// activations
// ---------------------------------------------------------------------------------------------
// nn.Softplus(beta=100, threshold=20): log1p(exp(100a))/100, identity above the threshold (mlp.py:76).
// Evaluated as max(a,0) + log1p(exp(-|100a|))/100 with the hardware exp2/log2: z = exp(-|100a|) is in (0,1], so 1+z is in
// (1,2] and log2(1+z) has an ABSOLUTE error of ~1 ulp(1) = 6e-8, i.e. 6e-10 on h after the /100 -- below fp32 resolution of
// the O(0.01..1) activations it is added to; above the threshold z < 2e-9 vanishes against 1 and h == a exactly (torch's
// threshold branch).  6 VALU ops, 2 of them transcendental.
// max(a, 0) as ONE instruction: fmaxf() makes hipcc put a canonicalising v_max_f32 x,x,x in front of the v_max (the operand
// could be a signalling NaN); beside MFMAs every VALU slot counts (8 of them per k-chunk in the bf16x3 forward).  Only in the
// translation units that define I2SDF_RELU_ASM (the bf16x3 kernels, built with the lifted unroll cap): the unroller prices an
// inline-asm statement far above one instruction, and in the other units the fully unrolled stage loops would fall back to rolled
// loops with their register arrays in scratch (measured: the sampler's forward 24x slower).
__device__ __forceinline__ float relu0(float a) {
#ifdef I2SDF_RELU_ASM
  float r;
  asm("v_max_f32 %0, 0, %1" : "=v"(r) : "v"(a));
  return r;
#else
  return fmaxf(a, 0.f);
#endif
}
__device__ __forceinline__ float softplus100(float a) {
  const float z = __builtin_amdgcn_exp2f(-fabsf(a) * (100.f * 1.44269504088896341f));
  const float l = __builtin_amdgcn_logf(1.0f + z);                 // log2
  return fmaf(l, 0.693147180559945309f * 0.01f, relu0(a));
}
// sigma = softplus100'(a) recovered from h = softplus100(a):  1 - exp(-100 h)   (exactly 1 in the threshold
// branch up to rounding: 1 - e^-20 rounds to 1.0f).
__device__ __forceinline__ float sp_sigma_from_h(float h) {
  return 1.0f - __builtin_amdgcn_exp2f(-100.f * 1.44269504088896341f * h);
}

// ---------------------------------------------------------------------------------------------
// positional encoding in B-operand layout -- embedder.py:6-38 (include_input, log-sampled powers of two,
// [sin, cos] per frequency, each block 3 wide).  PEC = chunks (8 reduction indices each), zero padded.
// ---------------------------------------------------------------------------------------------
template <int LF>
struct PE {
  static constexpr int DIM = 3 + 6 * LF;
  static constexpr int PEC = cdiv(DIM, 8);
};

// sin/cos(2^k x) for k = 0..LF-1 with ONE range reduction: r = x/(2 pi) in revolutions as a hi+lo pair (fma residual),
// 2^k r is exact, its fractional part is exact (t - rint(t)), and v_sin_f32 / v_cos_f32 take revolutions directly.
// Reduction error <= 1.2e-7 abs for |x| <= 6 (numpy emulation); ~25x cheaper than 2*LF*3 libm calls.
template <int LF>
__device__ __forceinline__ void pe_full(float x, float y, float z, float (&full)[PE<LF>::PEC * 8]) {
#pragma unroll
  for (int i = 0; i < PE<LF>::PEC * 8; ++i) full[i] = 0.f;
  full[0] = x; full[1] = y; full[2] = z;
  constexpr float INV2PI_HI = 0.15915493667125702f, INV2PI_LO = 6.4206382432985265e-09f;
  const float v[3] = {x, y, z};
#pragma unroll
  for (int i = 0; i < 3; ++i) {
    const float r_hi = v[i] * INV2PI_HI;
    const float r_lo = fmaf(v[i], INV2PI_HI, -r_hi) + v[i] * INV2PI_LO;
#pragma unroll
    for (int k = 0; k < LF; ++k) {
      const float sc = (float)(1 << k);
      const float t = r_hi * sc;
      const float ph = (t - rintf(t)) + r_lo * sc;
      full[3 + 6 * k + i] = __builtin_amdgcn_sinf(ph);
      full[3 + 6 * k + 3 + i] = __builtin_amdgcn_cosf(ph);
    }
  }
}
template <int NC>
__device__ __forceinline__ void to_b_layout(const float (&full)[NC * 8], float (&regs)[NC * 4], int hi) {
#pragma unroll
  for (int c = 0; c < NC; ++c)
#pragma unroll
    for (int t = 0; t < 4; ++t) regs[c * 4 + t] = hi ? full[8 * c + 4 + t] : full[8 * c + t];
}

// ---------------------------------------------------------------------------------------------
// D-layout tile <-> point-major HBM rows ([m][ld] fp32): 16 B per lane per (nt,q); a 128 B line is
// completed by the 4 q-stores of one tile.
// ---------------------------------------------------------------------------------------------
template <int NT>
__device__ __forceinline__ void store_tile(float* __restrict__ row, int hi, bool valid, const f32x16 (&t)[NT]) {
  if (!valid) return;
#pragma unroll
  for (int nt = 0; nt < NT; ++nt)
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      f32x4 v = {t[nt][4 * q], t[nt][4 * q + 1], t[nt][4 * q + 2], t[nt][4 * q + 3]};
      stg4(row + 32 * nt + 8 * q + 4 * hi, v);
    }
}
// `kcs` = floats between consecutive 16-wide k-chunks of the row: 16 in the point-major layout, 512 in the blocked layout of the
// saved tensors (mlp_common.h: save_row_off)
template <int NC>
__device__ __forceinline__ void store_regs(float* __restrict__ row, int hi, bool valid, const float (&r)[NC * 4], int kcs = 16) {
  if (!valid) return;
#pragma unroll
  for (int c = 0; c < NC; ++c) {
    f32x4 v = {r[4 * c], r[4 * c + 1], r[4 * c + 2], r[4 * c + 3]};
    stg4(row + (c >> 1) * kcs + (c & 1) * 8 + 4 * hi, v);
  }
}
template <int NC>
__device__ __forceinline__ void load_regs(const float* __restrict__ row, int hi, float (&r)[NC * 4], int kcs = 16) {
#pragma unroll
  for (int c = 0; c < NC; ++c) {
    f32x4 v = ldg4(row + (c >> 1) * kcs + (c & 1) * 8 + 4 * hi);
    r[4 * c] = v.x; r[4 * c + 1] = v.y; r[4 * c + 2] = v.z; r[4 * c + 3] = v.w;
  }
}

}  // namespace i2sdf
