// Weight-gradient GEMMs  dW[n][k] = sum_m X[m][n] * Y[m][k]  over the per-point tensors the backward kernels saved,
// followed by the split-M reduction fused with the weight-norm backward.
// In the reference these are the `mm`/`addmm` backward nodes autograd runs for every nn.Linear (mlp.py:97,222) plus
// the _weight_norm backward.  X, Y are point-major ([Mp][ld]); a wave owns a 128x128 output tile and streams
// point pairs straight from HBM into MFMA operands (fp32 32x32x2: the reduction index of the MFMA is the point).
#include <algorithm>
#include <stdlib.h>
#include "mlp_common.h"

using namespace i2sdf;

int i2sdf_hip_check(hipError_t e, const char* what);

namespace {

// The weight-gradient kernels stream their operands once (no line is touched twice: wgrad3p's lane quads read whole 64-byte records, both
// points of a 128-byte line in one instruction): NON-TEMPORAL loads, which leave the packed weight streams of the concurrently running
// sweeps in L2.  Round 5, A/B in one GPU call: i2sdf_weight_grads 1.04 -> 0.99-1.00 ms (profiles/r5_hbm_mix.txt; pure pattern: 6.7 vs 6.1 TB/s).
#ifndef WGN_AUX
#define WGN_AUX 2      // cache policy of the narrow split kernel's buffer loads (2 = nt, 0 = plain)
#endif
#ifndef W3_NT_LD
#define W3_NT_LD 1
#endif
// raw-row register sets of the 256x256 body: 1 = a pair's registers are refilled with the rows of stage s+2 while stage s runs (one stage in flight);
// 2 = two sets, refilled with stage s+3 (two stages in flight; +24-32 VGPRs) -- the kernel keeps ~56 VGPRs free
#ifndef W3_RAW_SETS
#define W3_RAW_SETS 1
#endif
#ifndef W3_P24_NT
#define W3_P24_NT 0      // the packed 24-bit operands' loads: plain (see w3p_job; A/B in profiles/r6_saves24.txt)
#endif
constexpr int WG_CH = I2SDF_WG_CH;   // points per split-M chunk (plan.h; plan.cpp: PART_ALIGN -- point ranges are cut at chunk boundaries)
constexpr int PFW = 6;               // point pairs in flight
constexpr int MAX_TASKS = 30;

// A / B point at column a_c0 / b_c0 of the operand's first row in the POINT-MAJOR sense; a_blk / b_blk = leading points of the
// operand tensor stored in the blocked layout (mlp_common.h; only 256-wide tensors, accessed as 128-column tiles, are ever blocked)
// a_p24 / b_p24: the operand's blocked points are packed 24-bit records (x3.h P24: abars, gus, gas under I2SDF_OPT_SAVES24; only 256-wide tensors,
// read by the split-arithmetic kernels)
struct WgJob { const float* A; const float* B; int32_t lda, ldb, a_w, b_w; int64_t m_count; int64_t a_blk, b_blk; int32_t a_c0, b_c0; int32_t a_p24, b_p24; };
struct WgTask {
  WgJob j[2];
  int32_t njobs, relu_b, has_bias, rows_store, cols_store, ldo;
  int32_t variant, pad;
  int64_t out_off, bias_off;
};
struct WgLaunch { WgTask t[MAX_TASKS]; int32_t n; int32_t chunk0; int64_t chunk_stride; float* partials; };     // chunk0: first chunk of this launch (point ranges)

// AM = 0: A tile 128 wide (16 B per lane, rows 4i+ta)      AM = 1: A narrower than 32 columns (4 B per lane, row i)
// BM = 0: B tile 128 wide (16 B per lane, cols 4j+tb)      BM = 1 / 2: B at most 32 / 64 columns wide (4 B per lane, col j + 32 tb)
// One wave accumulates the points [lo, hi) of every job of task t (lo/hi relative to the operands' first row; hi is capped by
// the job's point count).  PF = point pairs per load group.
template <int AM, int BM, int PF>
__device__ __forceinline__ void wgrad_accumulate(const WgTask& t, int64_t lo, int64_t hi_cap, int lane,
                                                 f32x16 (&acc)[AM ? 1 : 4][BM ? BM : 4], float (&bsum)[AM ? 1 : 4]) {
  constexpr int TA = AM ? 1 : 4, TB = BM ? BM : 4;
  const int i32 = lane & 31, hi = lane >> 5;
#pragma unroll
  for (int a = 0; a < TA; ++a)
#pragma unroll
    for (int b = 0; b < TB; ++b)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[a][b][r] = 0.f;
#pragma unroll
  for (int a = 0; a < TA; ++a) bsum[a] = 0.f;
  for (int jb = 0; jb < t.njobs; ++jb) {
    const WgJob job = t.j[jb];
    const int64_t m_lo = lo;
    const int64_t m_hi = (hi_cap < job.m_count) ? hi_cap : job.m_count;
    if (m_hi <= m_lo) continue;
    const int npairs = (int)((m_hi - m_lo + 1) / 2);
    // Buffer loads with hardware range checking: rows >= m_hi and columns >= the operand width read as 0, so the
    // loop body has no branches and the loads stay a whole group of pairs ahead of the MFMAs.
    const int rows = (int)(m_hi - m_lo);
    const __amdgpu_buffer_rsrc_t ra = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(job.A + m_lo * job.lda), 0,
                                                                      rows * job.lda * 4, 0x00020000);
    const __amdgpu_buffer_rsrc_t rb = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(job.B + m_lo * job.ldb), 0,
                                                                      rows * job.ldb * 4, 0x00020000);
    const unsigned OOB = 0x7fffffffu;
    const int a_col = AM ? i32 : 4 * i32;
    const unsigned a_off = (a_col < job.a_w) ? (unsigned)((hi * job.lda + a_col) * 4) : OOB;
    unsigned b_off[BM ? BM : 1];
    if (BM) {
#pragma unroll
      for (int tb = 0; tb < (BM ? BM : 1); ++tb) b_off[tb] = (i32 + 32 * tb < job.b_w) ? (unsigned)((hi * job.ldb + i32 + 32 * tb) * 4) : OOB;
    } else {
      b_off[0] = (4 * i32 < job.b_w) ? (unsigned)((hi * job.ldb + 4 * i32) * 4) : OOB;
    }
    const unsigned a_step = 2u * job.lda * 4u, b_step = 2u * job.ldb * 4u;
    // The 16-B-per-lane operands (128-column tiles of 256-wide tensors) may be stored blocked (mlp_common.h).  Chunks start on a
    // tile boundary, so the chunk's base address is the same in both layouts: point r of the chunk sits at (r/32)*8192 + (r%32)*16
    // floats, column c at (c/16)*512 + c%16 (blocked) or at r*ld + c (point-major).  One descriptor over the chunk serves both;
    // the offset is SELECTED (no branch in the load groups), rows beyond the job and columns beyond the width read as 0 (OOB).
    const bool ablk = !AM && m_lo < job.a_blk, bblk = !BM && m_lo < job.b_blk;
    const unsigned a_cb = ablk ? (unsigned)((((job.a_c0 + 4 * i32) >> 4) * 512 + ((4 * i32) & 15)) * 4) : (unsigned)((job.a_c0 + 4 * i32) * 4);
    const unsigned b_cb = bblk ? (unsigned)((((job.b_c0 + 4 * i32) >> 4) * 512 + ((4 * i32) & 15)) * 4) : (unsigned)((job.b_c0 + 4 * i32) * 4);
    const __amdgpu_buffer_rsrc_t ra_b = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(job.A - job.a_c0 + m_lo * job.lda), 0, 0x7ffffff0, 0x00020000);
    const __amdgpu_buffer_rsrc_t rb_b = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(job.B - job.b_c0 + m_lo * job.ldb), 0, 0x7ffffff0, 0x00020000);
    auto off16 = [&](unsigned cb, bool is_blk, unsigned ld, unsigned pn, bool col_ok) -> unsigned {
      const unsigned r = 2u * pn + (unsigned)hi;
      const unsigned ro = is_blk ? ((r >> 5) * 8192u + (r & 31u) * 16u) * 4u : r * ld * 4u;
      return (col_ok && (int)r < rows) ? cb + ro : OOB;
    };
    const float relu_lo = t.relu_b != 0 ? 0.f : -3.0e38f;           // branch-free optional ReLU on the B operand
    const float bias_w = (t.has_bias && jb == 0) ? 1.f : 0.f;       // branch-free optional column sums of A
    // Group double buffering: while the MFMAs of one group of point pairs run, the loads of the NEXT group are in flight.
    float a0[PF][TA], b0[PF][TB], a1[PF][TA], b1[PF][TB];
    auto ldw = [&](const __amdgpu_buffer_rsrc_t& rs, unsigned off, unsigned add) -> float {
      return __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rs, off == OOB ? OOB : off + add, 0, 0));
    };
    auto load_group = [&](float (&A)[PF][TA], float (&Bv)[PF][TB], unsigned pbase) {
#pragma unroll
      for (int u = 0; u < PF; ++u) {
        const unsigned pn = pbase + u;
        if (AM) A[u][0] = ldw(ra, a_off, pn * a_step);
        else {
          const f32x4 x = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(ra_b, off16(a_cb, ablk, (unsigned)job.lda, pn, a_off != OOB), 0, 0));
#pragma unroll
          for (int q = 0; q < TA; ++q) A[u][q] = x[q];
        }
        if (BM) {
#pragma unroll
          for (int tb = 0; tb < TB; ++tb) Bv[u][tb] = ldw(rb, b_off[tb < (BM ? BM : 1) ? tb : 0], pn * b_step);
        } else {
          const f32x4 x = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rb_b, off16(b_cb, bblk, (unsigned)job.ldb, pn, b_off[0] != OOB), 0, 0));
#pragma unroll
          for (int q = 0; q < TB; ++q) Bv[u][q] = x[q];
        }
      }
    };
    auto compute_group = [&](const float (&A)[PF][TA], const float (&Bv)[PF][TB]) {
#pragma unroll
      for (int u = 0; u < PF; ++u) {
        float bq[TB];
#pragma unroll
        for (int tb = 0; tb < TB; ++tb) bq[tb] = fmaxf(Bv[u][tb], relu_lo);
#pragma unroll
        for (int ta = 0; ta < TA; ++ta) bsum[ta] = fmaf(A[u][ta], bias_w, bsum[ta]);
#pragma unroll
        for (int ta = 0; ta < TA; ++ta)
#pragma unroll
          for (int tb = 0; tb < TB; ++tb) acc[ta][tb] = mfma(A[u][ta], bq[tb], acc[ta][tb]);
      }
    };
    load_group(a0, b0, 0);
    for (int p = 0; p < npairs; p += 2 * PF) {
      load_group(a1, b1, (unsigned)(p + PF));
      __builtin_amdgcn_sched_barrier(0);
      compute_group(a0, b0);
      __builtin_amdgcn_sched_barrier(0);
      load_group(a0, b0, (unsigned)(p + 2 * PF));
      __builtin_amdgcn_sched_barrier(0);
      compute_group(a1, b1);
      __builtin_amdgcn_sched_barrier(0);
    }
  }
}

// ---- accumulator tiles as hidden AGPR state (used by wgrad_accumulate_x below and by wgrad3p, where the comment block explains why) ----
#define W3_ALL_AGPRS "a0","a1","a2","a3","a4","a5","a6","a7","a8","a9","a10","a11","a12","a13","a14","a15","a16","a17","a18","a19","a20","a21","a22","a23","a24","a25","a26","a27","a28","a29","a30","a31","a32","a33","a34","a35","a36","a37","a38","a39","a40","a41","a42","a43","a44","a45","a46","a47","a48","a49","a50","a51","a52","a53","a54","a55","a56","a57","a58","a59","a60","a61","a62","a63","a64","a65","a66","a67","a68","a69","a70","a71","a72","a73","a74","a75","a76","a77","a78","a79","a80","a81","a82","a83","a84","a85","a86","a87","a88","a89","a90","a91","a92","a93","a94","a95","a96","a97","a98","a99","a100","a101","a102","a103","a104","a105","a106","a107","a108","a109","a110","a111","a112","a113","a114","a115","a116","a117","a118","a119","a120","a121","a122","a123","a124","a125","a126","a127","a128","a129","a130","a131","a132","a133","a134","a135","a136","a137","a138","a139","a140","a141","a142","a143","a144","a145","a146","a147","a148","a149","a150","a151","a152","a153","a154","a155","a156","a157","a158","a159","a160","a161","a162","a163","a164","a165","a166","a167","a168","a169","a170","a171","a172","a173","a174","a175","a176","a177","a178","a179","a180","a181","a182","a183","a184","a185","a186","a187","a188","a189","a190","a191","a192","a193","a194","a195","a196","a197","a198","a199","a200","a201","a202","a203","a204","a205","a206","a207","a208","a209","a210","a211","a212","a213","a214","a215","a216","a217","a218","a219","a220","a221","a222","a223","a224","a225","a226","a227","a228","a229","a230","a231","a232","a233","a234","a235","a236","a237","a238","a239","a240","a241","a242","a243","a244","a245","a246","a247","a248","a249","a250","a251","a252","a253","a254","a255"
__device__ __forceinline__ void w3_mfma(int k, f32x16& c, u32x4 a, u32x4 b) {
  (void)c;                                         // k is a constant after unrolling
  asm volatile("v_mfma_f32_32x32x16_bf16 a[%0:%1], %2, %3, a[%0:%1]" :: "n"(16 * k), "n"(16 * k + 15), "v"(a), "v"(b) : W3_ALL_AGPRS);
}
__device__ __forceinline__ void w3_acc_zero() {
  asm volatile(".set w3i, 0\n.rept 256\n\tv_accvgpr_write_b32 a[w3i], 0\n\t.set w3i, w3i+1\n.endr" ::: W3_ALL_AGPRS);
}
__device__ __forceinline__ float w3_acc_read(int n) {       // element n & 15 of tile n >> 4
  float x;
  asm volatile("v_accvgpr_read_b32 %0, a[%1]" : "=v"(x) : "n"(n));
  return x;
}
__device__ __forceinline__ void w3_mfma_drain() {
  asm volatile("s_nop 15\n\ts_nop 15" ::: "memory");
}
// the same MFMA behind a VALU producer of its operands: the two wait states the hazard recognizer would insert between a VALU write
// and an MFMA read of the same register are part of the statement (the compiler does not know that the asm is an MFMA)
__device__ __forceinline__ void w3_mfma_valu(int k, u32x4 a, u32x4 b) {
  asm volatile("s_nop 1\n\tv_mfma_f32_32x32x16_bf16 a[%0:%1], %2, %3, a[%0:%1]" :: "n"(16 * k), "n"(16 * k + 15), "v"(a), "v"(b) : W3_ALL_AGPRS);
}

// The same accumulation in bf16 split arithmetic (NPL = 2 or 3 planes per operand: three or six products per block, x3.h): the point
// is the reduction index of v_mfma_f32_32x32x16_bf16, 16 points per MFMA where the fp32 form takes 2.  Lane (i32, hi) holds the points
// 2q + hi (q = 0..7) of a 16-point stage -- the two halves of a wave read the two points of a 128-B line, as in wgrad3p -- in BOTH
// operands, so the product does not depend on the point <-> k-slot map; column masks, optional ReLU and the bias sums (fp32 VALU adds
// over the raw values) are those of wgrad_accumulate, and so is the accumulator layout (wgrad_store is shared).  Addressing: a per-lane
// byte offset per k-slot computed once per job (an out-of-range value for a masked column: the buffer load returns 0) plus ONE scalar
// offset per stage; no selects or branches in the stage loop; the rows of a ragged last stage are zeroed after the load.
// (core: leaves the tiles in a[16 k : 16 k + 15], k = ta + 4 tb, drained; wgrad_accumulate_x below reads them out)
// A24 (AM = 0 only): the A operand -- a 128-column tile of gas / abars, in both jobs of the task -- is packed 24-bit records (x3.h P24) in its blocked points
template <int AM, int BM, int NPL, bool A24 = false>
__device__ __forceinline__ void wgrad_accumulate_x_core(const WgTask& t, int64_t lo, int64_t hi_cap, int lane, float (&bsum)[AM ? 1 : 4]) {
  constexpr int TA = AM ? 1 : 4, TB = BM ? BM : 4, NBV = BM ? BM : 1;
  const int i32 = lane & 31, hi = lane >> 5;
  // The accumulator tiles live in a[16 k : 16 k + 15], k = ta + 4 tb, as state the compiler does not see (w3_mfma_valu / w3_acc_zero /
  // w3_acc_read): left to the register allocator they were shuttled between AGPRs and VGPRs ~1200 times per kernel and spilled
  // (the kernel ran 25 % SLOWER than its fp32 twin).  A tile is srcC again TA*TB >= 4 MFMAs after it was written.
  w3_acc_zero();
#pragma unroll
  for (int a = 0; a < TA; ++a) bsum[a] = 0.f;
  for (int jb = 0; jb < t.njobs; ++jb) {
    const WgJob job = t.j[jb];
    const int64_t m_lo = lo;
    const int64_t m_hi = (hi_cap < job.m_count) ? hi_cap : job.m_count;
    if (m_hi <= m_lo) continue;
    const int rows = (int)(m_hi - m_lo);
    const int nst = (rows + 15) / 16;
    const unsigned OOB = 0x7fffffffu;
    const bool ablk = !AM && m_lo < job.a_blk, bblk = !BM && m_lo < job.b_blk;
    // packed 24-bit records (x3.h P24; wave-uniform): the 128-column operand is read as 8 B of upper halves + 4 mid bytes per point and column quad,
    // into the first three slots of the raw row, and unpacked in front of the split (compute_stage)
    constexpr bool a24 = A24 && !AM;          // (host: a packed operand is blocked throughout, mlp_common.h sdf_saves24; B is never packed here)
    // descriptors over "everything behind the first row of this wave's range" (the column offset a_c0 / b_c0 of a 128-column tile is
    // folded into the per-lane offset for the blocked layout, where columns are not contiguous)
    const __amdgpu_buffer_rsrc_t ra = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(job.A - (AM ? 0 : job.a_c0) + m_lo * (a24 ? P24_BLOCK / 32 : job.lda)), 0, 0x7ffffff0, 0x00020000);
    const __amdgpu_buffer_rsrc_t rb = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(job.B - (BM ? 0 : job.b_c0) + m_lo * job.ldb), 0, 0x7ffffff0, 0x00020000);
    const char* pa24 = reinterpret_cast<const char*>(job.A - (AM ? 0 : job.a_c0) + m_lo * (P24_BLOCK / 32));      // (a24 only)
    // per-lane byte offsets of the 8 k-slots (point 2q + hi of a stage), stage 0
    // (P24: the mid-byte offset of a k-slot follows from its upper-half offset oh: (oh >> 1) + mid_k, mid_k = (k-chunk base >> 1) + 1024 per lane)
    unsigned aoff[8], boff[8][NBV], amid_k = 0;
    auto p24_offs = [&](unsigned col, unsigned c, unsigned& oh, unsigned& mid_k) __attribute__((always_inline)) {
      const unsigned kc = col >> 4, q = (col >> 2) & 3u;
      oh = (kc * P24_KCS + c * 8u + (q & 1u) * 4u + (q >> 1) * 2u) * 4u;
      mid_k = (kc * P24_KCS * 4u >> 1) + P24_MID * 4u;
    };
#pragma unroll
    for (int q = 0; q < 8; ++q) {
      const unsigned c = (unsigned)(2 * q + hi);
      if (AM) aoff[q] = (i32 < job.a_w) ? (c * (unsigned)job.lda + (unsigned)i32) * 4u : OOB;
      else {
        const unsigned col = (unsigned)(job.a_c0 + 4 * i32);
        aoff[q] = (4 * i32 < job.a_w) ? (ablk ? ((col >> 4) * 512u + (col & 15u) + c * 16u) * 4u : (c * (unsigned)job.lda + col) * 4u) : OOB;
        if (a24) p24_offs(col, c, aoff[q], amid_k);         // (host: packed operands are whole 128-column tiles of 256-wide tensors -- no column mask)
      }
      if (BM) {
#pragma unroll
        for (int tb = 0; tb < NBV; ++tb) boff[q][tb] = (i32 + 32 * tb < job.b_w) ? (c * (unsigned)job.ldb + (unsigned)(i32 + 32 * tb)) * 4u : OOB;
      } else {
        const unsigned col = (unsigned)(job.b_c0 + 4 * i32);
        boff[q][0] = (4 * i32 < job.b_w) ? (bblk ? ((col >> 4) * 512u + (col & 15u) + c * 16u) * 4u : (c * (unsigned)job.ldb + col) * 4u) : OOB;
      }
    }
    // scalar byte offset of stage s: 16 points further -- blocked: 32-point blocks of 8192 floats, 16 floats per point inside
    auto stage_off = [&](int s, bool blk, int ld) -> unsigned {
      return blk ? (unsigned)(((s >> 1) * 8192 + (s & 1) * 256) * 4) : (unsigned)(s * 16 * ld * 4);
    };
    // ... packed 24-bit records: 32-point blocks of P24_BLOCK floats; the second 16 points of a block 128 floats (upper halves) / 64 floats (mid bytes) further
    // (part 1: the mid-byte loads form their offset as (upper-half offset + stage offset) / 2 + mid_k, see load_stage: the scalar part is the other half
    // of the block offset -- the (s & 1) term halves exactly: 128 floats -> 64 floats)
    auto stage_off24 = [&](int s, int part) -> unsigned { return part == 0 ? (unsigned)(((s >> 1) * P24_BLOCK + (s & 1) * 128) * 4) : (unsigned)((s >> 1) * P24_BLOCK * 2); };
    const float relu_lo = t.relu_b != 0 ? 0.f : -3.0e38f;
    const float bias_w = (t.has_bias && jb == 0) ? 1.f : 0.f;
    float A0[8][TA], B0[8][TB], A1[8][TA], B1[8][TB];           // raw values of two stages: [k-slot q][tile]
    auto load_stage = [&](float (&A)[8][TA], float (&Bv)[8][TB], int s) __attribute__((always_inline)) {
      const unsigned sa = stage_off(s, ablk, job.lda), sb = stage_off(s, bblk, job.ldb);
      if (!AM && a24) {
        const unsigned sh = stage_off24(s, 0), sm = stage_off24(s, 1);      // sm: what (offset + sh) / 2 leaves of the mid part's stage offset
#pragma unroll
        for (int q = 0; q < 8; ++q) {
          // plain global loads (no column mask to range-check), as floats: `__builtin_bit_cast(float, v[1])` on an element of an unsigned ext-vector
          // read element 0 with hipcc 7.2 (values 2, 3 of every quad carried the upper halves of values 0, 1: profiles/r6_saves24.txt)
          const f32x2 hh = *reinterpret_cast<const f32x2*>(pa24 + (sh + aoff[q]));
          A[q][0] = hh[0]; A[q][1 % TA] = hh[1];
          // (the mid-byte offset is formed from the STAGE's upper-half offset, per load: as a loop invariant it would be hoisted into eight more live registers)
          A[q][2 % TA] = *reinterpret_cast<const float*>(pa24 + sm + (((aoff[q] + sh) >> 1) + amid_k));
        }
      }
#pragma unroll
      for (int q = 0; q < 8; ++q) {
        if (AM) A[q][0] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(ra, aoff[q], sa, WGN_AUX));
        else if (!a24) {
          const f32x4 x = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(ra, aoff[q], sa, WGN_AUX));
#pragma unroll
          for (int ta = 0; ta < TA; ++ta) A[q][ta] = x[ta];
        }
        if (BM) {
#pragma unroll
          for (int tb = 0; tb < TB; ++tb) Bv[q][tb] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rb, boff[q][tb < NBV ? tb : 0], sb, WGN_AUX));
        } else {
          const f32x4 x = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rb, boff[q][0], sb, WGN_AUX));
#pragma unroll
          for (int tb = 0; tb < TB; ++tb) Bv[q][tb] = x[tb];
        }
      }
    };
    // packed 24-bit rows: the three raw dwords of a k-slot -> its four values (masked columns were read as zeros: they unpack to zeros)
    auto unpack24 = [&](float (&X)[4]) __attribute__((always_inline)) {
      const unsigned h0 = __builtin_bit_cast(unsigned, X[0]), h1 = __builtin_bit_cast(unsigned, X[1]), mm = __builtin_bit_cast(unsigned, X[2]);
      // (x3.h p24_quad_value, written out: value tt = upper half (tt & 1) of h[tt >> 1] << 16 | mid byte tt << 8)
      X[0] = __builtin_bit_cast(float, __builtin_amdgcn_perm(h0, mm, 0x0504000cu));
      X[1] = __builtin_bit_cast(float, __builtin_amdgcn_perm(h0, mm, 0x0706010cu));
      X[2] = __builtin_bit_cast(float, __builtin_amdgcn_perm(h1, mm, 0x0504020cu));
      X[3] = __builtin_bit_cast(float, __builtin_amdgcn_perm(h1, mm, 0x0706030cu));
    };
    auto split = [&](const float (&X)[8], u32x4 (&pl)[NPL]) __attribute__((always_inline)) {
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        if (NPL == 2) { unsigned p0, p1; split2_pair(X[2 * i], X[2 * i + 1], p0, p1); pl[0][i] = p0; pl[1][i] = p1; }
        else { unsigned p0, p1, p2; split3_pair(X[2 * i], X[2 * i + 1], p0, p1, p2); pl[0][i] = p0; pl[1][i] = p1; pl[NPL - 1][i] = p2; }
      }
    };
    // nv = valid points of the stage (16, or fewer in the ragged last stage: those rows exist in the padded tensors but are not data)
    auto compute_stage = [&](float (&A)[8][TA], float (&Bv)[8][TB], int nv) __attribute__((always_inline)) {
      if constexpr (!AM) {
        if (a24) {
#pragma unroll
          for (int q = 0; q < 8; ++q) unpack24(A[q]);
        }
      }
      if (nv < 16) {                            // (wave-uniform: the ragged last stage of a job; selects, not per-lane branches)
#pragma unroll
        for (int q = 0; q < 8; ++q) {
          const bool ok = 2 * q + hi < nv;
#pragma unroll
          for (int ta = 0; ta < TA; ++ta) A[q][ta] = ok ? A[q][ta] : 0.f;
#pragma unroll
          for (int tb = 0; tb < TB; ++tb) Bv[q][tb] = ok ? Bv[q][tb] : 0.f;
        }
      }
      u32x4 pa[TA][NPL], pb[TB][NPL];
#pragma unroll
      for (int ta = 0; ta < TA; ++ta) {
        float x[8];
#pragma unroll
        for (int q = 0; q < 8; ++q) { x[q] = A[q][ta]; bsum[ta] = fmaf(x[q], bias_w, bsum[ta]); }
        split(x, pa[ta]);
      }
#pragma unroll
      for (int tb = 0; tb < TB; ++tb) {
        float x[8];
#pragma unroll
        for (int q = 0; q < 8; ++q) x[q] = fmaxf(Bv[q][tb], relu_lo);
        split(x, pb[tb]);
      }
      // products of combined order < NPL: a0b0, a0b1, a1b0 (+ a1b1, a0b2, a2b0 with three planes); consecutive MFMAs on different tiles
#pragma unroll
      for (int pa_i = 0; pa_i < NPL; ++pa_i)
#pragma unroll
        for (int pb_i = 0; pb_i < NPL; ++pb_i) {
          if (pa_i + pb_i >= NPL) continue;
#pragma unroll
          for (int ta = 0; ta < TA; ++ta)
#pragma unroll
            for (int tb = 0; tb < TB; ++tb) w3_mfma_valu(ta + 4 * tb, pa[ta][pa_i], pb[tb][pb_i]);
        }
    };
    // the look-ahead loads never leave the job's rows (these descriptors do no range checking for us): beyond the last stage they
    // re-read it, and those values are not used
    load_stage(A0, B0, 0);
    for (int s = 0; s < nst; s += 2) {
      load_stage(A1, B1, s + 1 < nst ? s + 1 : nst - 1);
      __builtin_amdgcn_sched_barrier(0);
      compute_stage(A0, B0, rows - 16 * s);
      __builtin_amdgcn_sched_barrier(0);
      load_stage(A0, B0, s + 2 < nst ? s + 2 : nst - 1);
      __builtin_amdgcn_sched_barrier(0);
      if (s + 1 < nst) compute_stage(A1, B1, rows - 16 * (s + 1));
      __builtin_amdgcn_sched_barrier(0);
    }
  }
  w3_mfma_drain();
}
template <int AM, int BM, int NPL>
__device__ __forceinline__ void wgrad_accumulate_x(const WgTask& t, int64_t lo, int64_t hi_cap, int lane,
                                                   f32x16 (&acc)[AM ? 1 : 4][BM ? BM : 4], float (&bsum)[AM ? 1 : 4]) {
  constexpr int TA = AM ? 1 : 4, TB = BM ? BM : 4;
  wgrad_accumulate_x_core<AM, BM, NPL>(t, lo, hi_cap, lane, bsum);
#pragma unroll
  for (int ta = 0; ta < TA; ++ta)
#pragma unroll
    for (int tb = 0; tb < TB; ++tb)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[ta][tb][r] = w3_acc_read(16 * (ta + 4 * tb) + r);
}

// one wave writes its tile (and, for the first column tile of a block, the column sums of A = the bias gradient) to the chunk's partials
template <int AM, int BM>
__device__ __forceinline__ void wgrad_store(const WgTask& t, float* __restrict__ out, int lane,
                                            const f32x16 (&acc)[AM ? 1 : 4][BM ? BM : 4], float (&bsum)[AM ? 1 : 4]) {
  constexpr int TA = AM ? 1 : 4, TB = BM ? BM : 4;
  const int i32 = lane & 31, hi = lane >> 5;
#pragma unroll
  for (int ta = 0; ta < TA; ++ta)
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int ri = (r & 3) + 8 * (r >> 2) + 4 * hi;
      const int n = AM ? ri : 4 * ri + ta;
      if (n < t.rows_store) {
        if (BM) {
#pragma unroll
          for (int tb = 0; tb < TB; ++tb)
            if (i32 + 32 * tb < t.cols_store) out[t.out_off + (int64_t)n * t.ldo + i32 + 32 * tb] = acc[ta][tb][r];
        } else if (4 * i32 < t.cols_store) {
          const f32x4 v = {acc[ta][0][r], acc[ta][1][r], acc[ta][2][r], acc[ta][3][r]};
          *reinterpret_cast<f32x4*>(out + t.out_off + (int64_t)n * t.ldo + 4 * i32) = v;
        }
      }
    }
  if (t.has_bias) {
#pragma unroll
    for (int ta = 0; ta < TA; ++ta) bsum[ta] += __shfl_xor(bsum[ta], 32);
    if (hi == 0) {
      if (AM) { if (i32 < t.rows_store) out[t.bias_off + i32] = bsum[0]; }
      else if (4 * i32 < t.rows_store) *reinterpret_cast<f32x4*>(out + t.bias_off + 4 * i32) = f32x4{bsum[0], bsum[TA > 1 ? 1 : 0], bsum[TA > 2 ? 2 : 0], bsum[TA > 3 ? 3 : 0]};
    }
  }
}

template <int AM, int BM>
__global__ __launch_bounds__(64) void wgrad_kernel(WgLaunch L) {
  constexpr int TA = AM ? 1 : 4, TB = BM ? BM : 4;
  const int lane = threadIdx.x & 63;
  const WgTask& t = L.t[blockIdx.y];
  const int64_t chunk = blockIdx.x + L.chunk0;
  f32x16 acc[TA][TB];
  float bsum[TA];
  wgrad_accumulate<AM, BM, PFW>(t, chunk * WG_CH, chunk * WG_CH + WG_CH, lane, acc, bsum);
  wgrad_store<AM, BM>(t, L.partials + chunk * L.chunk_stride, lane, acc, bsum);
}

// ---------------------------------------------------------------------------------------------------------------
// The narrow blocks (an operand at most 64 columns wide: positional-encoding columns, the sdf / rgb output rows) are few:
// 10 tiles x 100 chunks at synthetic.yml shapes, i.e. fewer single-wave tasks than the chip has SIMDs, each a long serial
// stream.  They run as ONE launch of 4-wave workgroups: a wave takes a quarter of the chunk's points, the four partial tiles
// are summed through LDS in a fixed order (deterministic) and wave 0 stores.  Three separate launches of single-wave
// workgroups took 0.62 ms per step.
// ---------------------------------------------------------------------------------------------------------------
constexpr int PFN = 8;               // 128 point pairs per wave = 8 groups of 16
constexpr int WGW_SLOT = (16 * 16 + 4) * 64;               // floats of one wave's partial 128x128 tile (+ 4 bias sums) in LDS
constexpr int WGN_LDS_BYTES = 2 * WGW_SLOT * 4;            // two such slots (wgrad_wide_body) > three narrow partials (3 * (8 * 16 + 4) * 64 * 4)

// NPL = 0: fp32-input MFMA (wgrad_accumulate); 2 / 3: bf16 split arithmetic with that many planes per operand (wgrad_accumulate_x)
template <int AM, int BM, int NPL, bool A24 = false>
__device__ __forceinline__ void wgrad_narrow_body(const WgLaunch& L, const WgTask& t, int64_t chunk, int wave, int lane, float* lds) {
  constexpr int TA = AM ? 1 : 4, TB = BM ? BM : 4, NW = TA * TB * 16 + TA;
  const int64_t lo = chunk * WG_CH + wave * (WG_CH / 4);
  if constexpr (NPL != 0 && BM != 0) {
    // split form with a narrow B operand: the tiles stay in a[...] and go through LDS / to memory ONE TILE AT A TIME (16 registers), as in
    // wgrad_wide_body.  Read out into acc[TA][TB] all at once, the 4 x 2-tile variant (PE columns of layer 0 / the skip layer against a
    // 128-row tile) held 128 registers of tiles next to the store loop's addresses and spilled 26 of them (108 B of scratch in the
    // shipped round-4 binary; tests/test_wgrad3p_isa.py now holds this kernel to zero).  Same values, same summation order ((w0 + w1) + w2) + w3.
    float bsum[TA];
    wgrad_accumulate_x_core<AM, BM, NPL == 0 ? 2 : NPL, A24>(t, lo, lo + WG_CH / 4, lane, bsum);
    if (wave > 0) {
      float* dst = lds + (wave - 1) * NW * 64 + lane;
#pragma unroll
      for (int a = 0; a < TA; ++a)
#pragma unroll
        for (int b = 0; b < TB; ++b) {
#pragma unroll
          for (int r = 0; r < 16; ++r) dst[((a * TB + b) * 16 + r) * 64] = w3_acc_read(16 * (a + 4 * b) + r);
          __builtin_amdgcn_sched_barrier(0);
        }
#pragma unroll
      for (int a = 0; a < TA; ++a) dst[(TA * TB * 16 + a) * 64] = bsum[a];
    }
    __syncthreads();
    if (wave == 0) {
      const int i32 = lane & 31, hi = lane >> 5;
      float* out = L.partials + chunk * L.chunk_stride;
#pragma unroll
      for (int a = 0; a < TA; ++a)
#pragma unroll
        for (int b = 0; b < TB; ++b) {
          float x[16];
#pragma unroll
          for (int r = 0; r < 16; ++r) x[r] = w3_acc_read(16 * (a + 4 * b) + r);
#pragma unroll
          for (int w = 0; w < 3; ++w)
#pragma unroll
            for (int r = 0; r < 16; ++r) x[r] += lds[(w * NW + (a * TB + b) * 16 + r) * 64 + lane];
          if (i32 + 32 * b < t.cols_store) {
#pragma unroll
            for (int r = 0; r < 16; ++r) {
              const int ri = (r & 3) + 8 * (r >> 2) + 4 * hi, n = AM ? ri : 4 * ri + a;          // as wgrad_store
              if (n < t.rows_store) out[t.out_off + (int64_t)n * t.ldo + i32 + 32 * b] = x[r];
            }
          }
          __builtin_amdgcn_sched_barrier(0);
        }
      if (t.has_bias) {
#pragma unroll
        for (int w = 0; w < 3; ++w)
#pragma unroll
          for (int a = 0; a < TA; ++a) bsum[a] += lds[(w * NW + TA * TB * 16 + a) * 64 + lane];
#pragma unroll
        for (int a = 0; a < TA; ++a) bsum[a] += __shfl_xor(bsum[a], 32);
        if (hi == 0) {
          if (AM) { if (i32 < t.rows_store) out[t.bias_off + i32] = bsum[0]; }
          else if (4 * i32 < t.rows_store) *reinterpret_cast<f32x4*>(out + t.bias_off + 4 * i32) = f32x4{bsum[0], bsum[TA > 1 ? 1 : 0], bsum[TA > 2 ? 2 : 0], bsum[TA > 3 ? 3 : 0]};
        }
      }
    }
    return;
  }
  f32x16 acc[TA][TB];
  float bsum[TA];
  if (NPL == 0) wgrad_accumulate<AM, BM, PFN>(t, lo, lo + WG_CH / 4, lane, acc, bsum);
  else wgrad_accumulate_x<AM, BM, NPL == 0 ? 2 : NPL>(t, lo, lo + WG_CH / 4, lane, acc, bsum);
  if (wave > 0) {
    float* dst = lds + (wave - 1) * NW * 64 + lane;
#pragma unroll
    for (int a = 0; a < TA; ++a)
#pragma unroll
      for (int b = 0; b < TB; ++b) {
#pragma unroll
        for (int r = 0; r < 16; ++r) dst[((a * TB + b) * 16 + r) * 64] = acc[a][b][r];
        __builtin_amdgcn_sched_barrier(0);
      }
#pragma unroll
    for (int a = 0; a < TA; ++a) dst[(TA * TB * 16 + a) * 64] = bsum[a];
  }
  __syncthreads();
  if (wave == 0) {
    // tile by tile (the scheduler would otherwise hoist all 3 x NW LDS reads in front of the adds and spill)
#pragma unroll
    for (int a = 0; a < TA; ++a)
#pragma unroll
      for (int b = 0; b < TB; ++b) {
#pragma unroll
        for (int w = 0; w < 3; ++w)
#pragma unroll
          for (int r = 0; r < 16; ++r) acc[a][b][r] += lds[(w * NW + (a * TB + b) * 16 + r) * 64 + lane];
        __builtin_amdgcn_sched_barrier(0);
      }
#pragma unroll
    for (int w = 0; w < 3; ++w)
#pragma unroll
      for (int a = 0; a < TA; ++a) bsum[a] += lds[(w * NW + TA * TB * 16 + a) * 64 + lane];
    wgrad_store<AM, BM>(t, L.partials + chunk * L.chunk_stride, lane, acc, bsum);
  }
}

// A 128 x 128 tile of a block that is not 256 x 256 (the light-mask head's 128 x 256 layer: A = G(a_0), B = relu(feature)) in the same
// 4-wave form and split arithmetic.  Sixteen accumulator tiles are the whole AGPR file, so they are never held in VGPRs all at once: the
// partial tiles go through LDS tile by tile straight from a[...] -- waves 2 and 3 write theirs, wave 1 adds its own onto wave 3's, wave 0
// sums (w0 + w2) + (w1 + w3) and stores (a fixed order: deterministic).  As single-wave fp32-input MFMA tasks (wgrad_kernel<0, 0>, 50 waves
// on the chip for half a batch) these two tiles were the LONGEST kernel of a cfg-3 step: 650 us per launch.
template <int NPL>
__device__ __forceinline__ void wgrad_wide_body(const WgLaunch& L, const WgTask& t, int64_t chunk, int wave, int lane, float* lds) {
  float bsum[4];
  const int64_t lo = chunk * WG_CH + wave * (WG_CH / 4);
  wgrad_accumulate_x_core<0, 0, NPL>(t, lo, lo + WG_CH / 4, lane, bsum);
  if (wave >= 2) {
    float* d = lds + (wave - 2) * WGW_SLOT + lane;
#pragma unroll
    for (int k = 0; k < 16; ++k)
#pragma unroll
      for (int r = 0; r < 16; ++r) d[(k * 16 + r) * 64] = w3_acc_read(16 * k + r);
#pragma unroll
    for (int a = 0; a < 4; ++a) d[(256 + a) * 64] = bsum[a];
  }
  __syncthreads();
  if (wave == 1) {
    float* d = lds + WGW_SLOT + lane;
#pragma unroll
    for (int k = 0; k < 16; ++k)
#pragma unroll
      for (int r = 0; r < 16; ++r) d[(k * 16 + r) * 64] += w3_acc_read(16 * k + r);
#pragma unroll
    for (int a = 0; a < 4; ++a) d[(256 + a) * 64] += bsum[a];
  }
  __syncthreads();
  if (wave == 0) {
    const float* s0 = lds + lane;
    const float* s1 = lds + WGW_SLOT + lane;
    const int i32 = lane & 31, hi = lane >> 5;
    float* out = L.partials + chunk * L.chunk_stride;
#pragma unroll
    for (int ta = 0; ta < 4; ++ta) {
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        f32x4 v;
#pragma unroll
        for (int tb = 0; tb < 4; ++tb) {
          const int k = ta + 4 * tb;
          v[tb] = (w3_acc_read(16 * (ta + 4 * tb) + r) + s0[(k * 16 + r) * 64]) + s1[(k * 16 + r) * 64];
        }
        const int ri = (r & 3) + 8 * (r >> 2) + 4 * hi, n = 4 * ri + ta;          // as wgrad_store<0, 0>
        if (n < t.rows_store && 4 * i32 < t.cols_store) *reinterpret_cast<f32x4*>(out + t.out_off + (int64_t)n * t.ldo + 4 * i32) = v;
      }
      __builtin_amdgcn_sched_barrier(0);
    }
    if (t.has_bias) {
#pragma unroll
      for (int a = 0; a < 4; ++a) {
        bsum[a] = (bsum[a] + s0[(256 + a) * 64]) + s1[(256 + a) * 64];
        bsum[a] += __shfl_xor(bsum[a], 32);
      }
      if (hi == 0 && 4 * i32 < t.rows_store) *reinterpret_cast<f32x4*>(out + t.bias_off + 4 * i32) = f32x4{bsum[0], bsum[1], bsum[2], bsum[3]};
    }
  }
}

template <int NPL>
__global__ __launch_bounds__(256) void wgrad_narrow_kernel(WgLaunch L) {
  extern __shared__ __attribute__((aligned(16))) float wgn_lds[];
  // wave index as a scalar: the buffer descriptors built from it must be wave-uniform (else every load becomes a waterfall loop)
  const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const WgTask& t = L.t[blockIdx.y];
  const int64_t chunk = blockIdx.x + L.chunk0;
  if (NPL != 0 && t.variant == 0) wgrad_wide_body<NPL == 0 ? 3 : NPL>(L, t, chunk, wave, lane, wgn_lds);      // (only the split form gets such tasks)
  else if (t.variant == 1) wgrad_narrow_body<1, 0, NPL>(L, t, chunk, wave, lane, wgn_lds);
  else if (t.variant == 2) { if (NPL != 0 && t.j[0].a_p24) wgrad_narrow_body<0, 1, NPL, NPL != 0>(L, t, chunk, wave, lane, wgn_lds); else wgrad_narrow_body<0, 1, NPL>(L, t, chunk, wave, lane, wgn_lds); }
  else { if (NPL != 0 && t.j[0].a_p24) wgrad_narrow_body<0, 2, NPL, NPL != 0>(L, t, chunk, wave, lane, wgn_lds); else wgrad_narrow_body<0, 2, NPL>(L, t, chunk, wave, lane, wgn_lds); }
}

constexpr int W3_PTS = 16;               // points per stage = one MFMA k-group of v_mfma_f32_32x32x16_bf16
// (Round 3 carried development knobs here -- wrong-result knock-outs for timing, a per-workgroup trace, a three-slot plane ring that
// measured slower: 660 vs 604 us per launch -- and compile-time alternatives of the MFMA / point interleave forms.  Round 4 removed them
// from the production source; git history and profiles/r3_wgrad3p_experiments.txt keep the measurements.)

// ---------------------------------------------------------------------------------------------------------------
// bf16x3 variant (x3.h) for full 256x256 blocks: one workgroup of 4 waves (2x2 tiles of 128x128) per block and chunk.
// 16 points form a stage.  Wave w owns one quarter of a stage (operand w>>1, column half w&1): it loads the raw fp32 rows
// straight into registers in MFMA operand layout (lane = column quad 4i..4i+3 x 8 consecutive points, one stage ahead), splits
// every value ONCE into three bf16 terms and writes the planes to LDS; consumers (two waves per quarter) read planes only and
// run the six leading partial products: 96 MFMAs of 32 cycles per stage and wave, where the fp32 kernel needs 128 of 64 cycles.
// History: every wave splitting both of its halves in registers (250 VGPRs, twice the VALU work) 1.75 ms; raw rows staged
// through LDS by LDS-DMA, scheduler-ordered stage 1.55 ms; this version (hand-placed stream, register loads) 1.35 ms, DESIGN.md.
//   LDS: plane ring 2 x 48 KB = 96 KB.
// ---------------------------------------------------------------------------------------------------------------
constexpr int W3P_PL = 4 * 4 * 3 * 256;                   // floats per plane stage: 4 quarters x 4 tiles x 3 planes x 1 KB (48 KB)
constexpr int W3P_SLOTS = 2;
constexpr int W3P_LDS_BYTES = W3P_SLOTS * W3P_PL * 4;      // 96 KB of the CU's 160 KB

// BLKA / BLKB: the A / B operand of this workgroup's chunk is in the blocked layout (compile-time: a runtime select inside the
// stage costs 5 % through a worse instruction stream).
// NPL = split planes per operand: 3 = bf16x3 (six partial products, error 2^-24: fp32-equivalent); 2 = bf16x2 (x = x0 + x1, the
// three products a0b0, a0b1, a1b0; per-product error <= 3 * 2^-18, I2SDF_OPT_WGRAD_BF16X2 -- see the note at the option).
// PLAIN = every job of this workgroup covers whole 16-point stages and no operand needs a ReLU: the row masks and the max are
// compiled out (all but the last chunk of a launch).
//
// The stage is a HAND-PLACED instruction stream.  One wave per SIMD issues in order, and an MFMA (32 cycles of pipe time) hides
// at most ~5 other instructions behind it (MI355X_MICROARCH.md).  The scheduler's own order ran the first half of a stage's MFMAs
// bare and packed the whole split between the MFMAs of the second half, eight VALU per MFMA (50 % MFMA-busy); sched_group_barrier
// pipelines changed the static order but not the time.  So the stage is cut into one UNIT per MFMA, fenced with
// sched_barrier(0): the MFMA, one sixth of a split item (2 values -> 3 planes, 11-17 VALU over 6 units; the 16 items of a
// quarter fill the 96 units exactly) and at most one LDS / global-memory instruction, each well ahead of its consumer.  (Build flags: -fno-slp-vectorize keeps the units' scalar ops from being merged into packed ops at one place,
// and packed f32 VALU is slow beside MFMAs anyway; the lifted pragma-unroll cap keeps the 96-unit loop unrolled.)
// The accumulator of wgrad3p is 16 tiles x 16 registers = all 256 AGPRs.  Left to itself the register allocator keeps three of the
// tiles in VGPRs across the stage loop's back edge and moves each into AGPRs at the head of an iteration and back at its end (48
// v_accvgpr_write + 48 v_accvgpr_read per two stages, in clusters of 16 to 48 that no MFMA can hide; many more around the last stage
// and the epilogue -- ISA of the round-2 kernel).  Neither an "a" constraint on the tile (the copies then feed the asm operand)
// nor naming the physical tuple in the constraint (the allocator spills the tiles to scratch between phases) changes that.  So with
// W3_ASM_MFMA the accumulator is STATE THE COMPILER DOES NOT SEE: tile (ta, tb) is a[16 k : 16 k + 15], k = ta + 4 tb, zero-filled,
// accumulated into and read out by inline asm that names the registers itself; every such statement clobbers all 256 AGPRs, so the
// compiler can keep nothing of its own in them across one (and its VGPR spills go to scratch, not to AGPRs).
// What the hazard recognizer does not see and this code provides: (1) a tile is the srcC of an MFMA again four MFMAs (128 cycles) after it
// was written -- more than the 8 passes of the instruction; (2) the epilogue's reads come behind w3_mfma_drain(); (3) the zero fill is
// separated from the first MFMA by the whole job prologue.  (s_waitcnt for the LDS-loaded A / B operands is the compiler's, as before.)
// LAY = layout of THIS WAVE'S operand quarter in this job and chunk: 0 point-major, 1 blocked fp32, 2 packed 24-bit records (x3.h P24).  A wave loads and
// splits only its own quarter (the consumers read planes from LDS), so the layout is a property of the wave's code path, chosen per job by a
// wave-uniform switch in wgrad3p_body: the four waves of a workgroup may run different instantiations (same barriers in the same order).
template <int LAY, int NPL, bool PLAIN>
__device__ __forceinline__ void w3p_job(const WgTask& t, const WgJob& job, int jb, int64_t chunk, float* plb, float (&bsum)[4], int w, int lane) {
  const int wa = w >> 1, wb = w & 1, i32 = lane & 31, kg = lane >> 5;
  const bool opB = (w >> 1) != 0;                          // this wave prepares a B quarter (else an A quarter)
  f32x16 acc_unused;                       // (the tiles are in a[0:255], see w3_mfma)
  {
    const int64_t m_lo = chunk * WG_CH;
    const int64_t m_hi = (m_lo + WG_CH < job.m_count) ? m_lo + WG_CH : job.m_count;
    const int rows = (int)(m_hi - m_lo);
    const int nst = (rows + W3_PTS - 1) / W3_PTS;
    const int dld = opB ? job.ldb : job.lda;
    constexpr bool blk = LAY != 0, P24 = LAY == 2;
    // Raw rows of stage s: this lane holds columns 4*i32 .. 4*i32+3 of its quarter for the 8 points 8*kg .. 8*kg+7, as four
    // point pairs (rlo[i], rhi[i]) = rows 8*kg+2i, 8*kg+2i+1 -- eight 16-byte loads per stage, address = scalar base + scalar
    // stage offset + a per-lane offset computed once per job (voff).  Rows of the last, partial stage that lie beyond `rows`
    // are read from the operand's padding (every operand has rows up to a multiple of 128, include/i2sdf.h) and masked when
    // they are split.
    //   point-major: the 32 lanes of one kg read 512 contiguous bytes of a row
    //   blocked (mlp_common.h): columns 4*i32.. of point r sit at r*16 + (i32>>2)*512 + 4*(i32&3): groups of 4 lanes read the 64
    //     bytes one point has in a k-chunk, and the pair's other point is the neighbouring 64 bytes (same 128 B line)
    //   packed 24-bit records (x3.h): the column quad 4*i32.. is quad q = i32 & 3 of k-chunk i32 >> 2: 8 B of upper halves at
    //     r*32 + (q & 1)*16 + (q >> 1)*8 and 4 mid bytes at 1024 + r*16 + (q & 1)*8 + (q >> 1)*4 of the k-chunk's 1536 B; a 16-point stage is 512 / 256 B further
    const float* ubase = P24 ? (opB ? job.B - job.b_c0 : job.A - job.a_c0) + m_lo * (P24_BLOCK / 32) + (8 * wb) * P24_KCS
                       : blk ? (opB ? job.B - job.b_c0 : job.A - job.a_c0) + m_lo * 256 + (8 * wb) * 512
                             : (opB ? job.B : job.A) + m_lo * dld + 128 * wb;
    unsigned voff[4];                           // bytes, row 8*kg + 2i; the pair's second row is `vnext` further
    unsigned voffm[P24 ? 4 : 1];                // P24: the mid-byte part (second row: vnext / 2 further)
    // the MFMA reduction index is the point, and any point <-> k-slot map will do as long as A and B use the same one: with the two
    // 32-lane halves taking the two points of a 128-B line (rows 4i + kg and 4i + 2 + kg) every load instruction moves whole lines
    // instead of half of each of twice as many
    const unsigned vnext = 2u * 4u * (unsigned)(P24 ? 8 : (blk ? 16 : dld));
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int r = 4 * i + kg;
      voff[i] = 4u * (unsigned)(P24 ? (i32 >> 2) * P24_KCS + r * 8 + (i32 & 1) * 4 + ((i32 >> 1) & 1) * 2
                                    : (blk ? r * 16 + (i32 >> 2) * 512 + 4 * (i32 & 3) : r * dld + 4 * i32));
      if (P24) voffm[i] = 4u * (unsigned)((i32 >> 2) * P24_KCS + P24_MID + r * 4 + (i32 & 1) * 2 + ((i32 >> 1) & 1));
    }
    const float relu_lo = (opB && t.relu_b != 0) ? 0.f : -3.0e38f;
    const float bias_w = (!opB && t.has_bias != 0 && jb == 0) ? 1.f : 0.f;
    constexpr int RS = W3_RAW_SETS;                     // register sets of raw rows (set of stage s: s % RS)
    f32x4 rlo[RS][P24 ? 1 : 4], rhi[RS][P24 ? 1 : 4];
    u32x2 hlo[RS][P24 ? 4 : 1], hhi[RS][P24 ? 4 : 1];           // P24: upper halves of the pair's two rows (values 0,1 | 2,3 of the quad) ...
    unsigned mlo[RS][P24 ? 4 : 1], mhi[RS][P24 ? 4 : 1];        // ... and their four mid bytes
    // (st = the register set, a constant at every call site: s % RS of the stage loaded)
    auto gload = [&](int st, int s, int i, int half) __attribute__((always_inline)) {
      const int sc = s < nst ? s : nst - 1;     // (the stream below prefetches ahead without a branch)
      if constexpr (P24) {
        // plain loads: the eight instructions of a stage touch every 128-B line of the quarter twice (8-B / 4-B pieces of 32-B / 16-B point records)
        const int64_t sblk = (int64_t)(sc >> 1) * P24_BLOCK;
        const char* sh = reinterpret_cast<const char*>(ubase + sblk + (sc & 1) * 128) + voff[i];
        const char* sm = reinterpret_cast<const char*>(ubase + sblk + (sc & 1) * 64) + voffm[i];
#if W3_P24_NT
        if (half == 0) { hlo[st][i] = __builtin_nontemporal_load(reinterpret_cast<const u32x2*>(sh)); mlo[st][i] = __builtin_nontemporal_load(reinterpret_cast<const unsigned*>(sm)); }
        else { hhi[st][i] = __builtin_nontemporal_load(reinterpret_cast<const u32x2*>(sh + vnext)); mhi[st][i] = __builtin_nontemporal_load(reinterpret_cast<const unsigned*>(sm + vnext / 2)); }
#else
        if (half == 0) { hlo[st][i] = *reinterpret_cast<const u32x2*>(sh); mlo[st][i] = *reinterpret_cast<const unsigned*>(sm); }
        else { hhi[st][i] = *reinterpret_cast<const u32x2*>(sh + vnext); mhi[st][i] = *reinterpret_cast<const unsigned*>(sm + vnext / 2); }
#endif
      } else {
      const int64_t soff = blk ? (int64_t)(sc >> 1) * 8192 + (sc & 1) * 256 : (int64_t)sc * W3_PTS * dld;
      const char* src = reinterpret_cast<const char*>(ubase + soff) + voff[i];
#if W3_NT_LD
      if (half == 0) rlo[st][i] = __builtin_nontemporal_load(reinterpret_cast<const f32x4*>(src));
      else rhi[st][i] = __builtin_nontemporal_load(reinterpret_cast<const f32x4*>(src + vnext));
#else
      if (half == 0) rlo[st][i] = *reinterpret_cast<const f32x4*>(src);
      else rhi[st][i] = *reinterpret_cast<const f32x4*>(src + vnext);
#endif
      }
    };
    // raw value of tile tt (a constant after unrolling) of the pair's first / second row
    auto raw_lo = [&](int st, int i, int tt) __attribute__((always_inline)) -> float {
      if constexpr (P24) return p24_quad_value(hlo[st][i], mlo[st][i], tt);
      else return rlo[st][i][tt];
    };
    auto raw_hi = [&](int st, int i, int tt) __attribute__((always_inline)) -> float {
      if constexpr (P24) return p24_quad_value(hhi[st][i], mhi[st][i], tt);
      else return rhi[st][i][tt];
    };
    unsigned pl[4][NPL][4];               // planes of the quarter being split: [tile][plane][point pair]
    // values of tile tt of point pair i (rows 2i, 2i+1 of this lane's 8) of stage s: mask / relu
    auto prep = [&](int s, int i, float lo, float hi_, float& x0, float& x1) __attribute__((always_inline)) {
      if (PLAIN) { x0 = lo; x1 = hi_; return; }
      const int p0 = s * W3_PTS + (4 * i + kg);
      x0 = p0 < rows ? fmaxf(lo, relu_lo) : 0.f;
      x1 = p0 + (2) < rows ? fmaxf(hi_, relu_lo) : 0.f;
    };
    // (the first argument of write_tile / plane is the slot's offset in floats: slot * W3P_PL)
    auto write_tile = [&](int so, int tt, int p) __attribute__((always_inline)) {
      float* dst = plb + so + w * (4 * 3 * 256) + lane * 4;
      *reinterpret_cast<u32x4*>(dst + (tt * 3 + p) * 256) = u32x4{pl[tt][p][0], pl[tt][p][1], pl[tt][p][2], pl[tt][p][3]};
    };
    auto plane = [&](int so, int quarter, int tile, int p) __attribute__((always_inline)) -> u32x4 {
      return *reinterpret_cast<const u32x4*>(plb + so + quarter * (4 * 3 * 256) + (tile * 3 + p) * 256 + lane * 4);
    };
    auto load_all = [&](int st, int s) __attribute__((always_inline)) {
#pragma unroll
      for (int i = 0; i < 4; ++i) { gload(st, s, i, 0); gload(st, s, i, 1); }
    };
    auto split_all = [&](int s) __attribute__((always_inline)) {          // rows of stage s (in rlo / rhi) -> pl, up front (job prologue)
#pragma unroll
      for (int i = 0; i < 4; ++i) {
#pragma unroll
        for (int tt = 0; tt < 4; ++tt) {
          float x0, x1;
          prep(s, i, raw_lo(0, i, tt), raw_hi(0, i, tt), x0, x1);       // (stage 0: set 0)
          bsum[tt] = fmaf(x0 + x1, bias_w, bsum[tt]);
          if (NPL == 3) split3_pair(x0, x1, pl[tt][0][i], pl[tt][1][i], pl[tt][NPL - 1][i]);
          else split2_pair(x0, x1, pl[tt][0][i], pl[tt][1][i]);
        }
      }
    };
    auto write_all = [&](int so) __attribute__((always_inline)) {
#pragma unroll
      for (int tt = 0; tt < 4; ++tt)
#pragma unroll
        for (int p = 0; p < NPL; ++p) write_tile(so, tt, p);
    };
    constexpr int NQ = (NPL == 3 ? 6 : 3), PH = 4 * NQ, NM = 4 * PH, GPI = NM / 16;     // MFMAs per phase / stage, units per item
    using BT = std::integral_constant<bool, true>; using BF = std::integral_constant<bool, false>;
    load_all(0, 0);
    __syncthreads();                       // the previous job is done with LDS
    split_all(0);                          // stage 0 is split up front
    load_all(1 % RS, 1);
    if (RS == 2) load_all(0, 2);           // (two sets: stage 2 into the set stage 0 has left)
    write_all(0);
    // ---- two plane slots.  MORE: stage s+1 exists: its raw rows (in rlo / rhi) are split here, and each point pair's registers are
    // refilled with the rows of stage s+2 as soon as the pair has been consumed (most of a stage ahead of their use)
    // (Pc = s & 1 and the loop below is written out for two stages: with ONE stage per loop iteration the rows loaded for the next stage
    // land in fresh registers and are copied into the loop-carried ones at the back edge -- eight v_mov_b64 behind an s_waitcnt vmcnt(1),
    // i.e. the loads are waited for in the stage that issues them: 658-695 vs 596-605 us per launch, 22-25 vs 11 % parked wave cycles.
    // Over two stages the registers line up without copies.)
    auto stage = [&](int s, auto Mc, auto Pc) __attribute__((always_inline)) {
      constexpr bool MORE = decltype(Mc)::value;
      constexpr int so0 = decltype(Pc)::value * W3P_PL, so1 = W3P_PL - so0;
      constexpr int NX = (decltype(Pc)::value + 1) % RS;      // register set of stage s + 1 (s has the parity Pc); refilled with stage s + 1 + RS
      __syncthreads();   // planes(s) written, everyone done with stage s-1
      u32x4 ap[4][NPL], bp[2][NPL];
      // the opening plane loads in the order of their first use (products (0,0) (0,1) (1,0) (0,2) (2,0) (1,1)): the first MFMA
      // waits for two loads, not for fifteen
#pragma unroll
      for (int p = 0; p < NPL; ++p) {
        bp[0][p] = plane(so0, 2 + wb, 0, p);
#pragma unroll
        for (int ta = 0; ta < 4; ++ta) ap[ta][p] = plane(so0, wa, ta, p);
      }
      float x0 = 0.f, x1 = 0.f, ra = 0.f, rb = 0.f;
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int g = 0; g < NM; ++g) {
        const int tb = g / PH, gg = g % PH, q = gg / 4, ta = g % 4;
        // (sa, sb): NPL 3: (0,0) (0,1) (1,0) (0,2) (2,0) (1,1);  NPL 2: (0,0) (0,1) (1,0)
        const int sa = (q == 2 || q == 5) ? 1 : (q == 4 ? NPL - 1 : 0);
        const int sb = (q == 1 || q == 5) ? 1 : (q == 3 ? NPL - 1 : 0);
        w3_mfma(ta + 4 * tb, acc_unused, ap[ta][sa], bp[tb & 1][sb]);
        if (MORE) {
          // split item `it` = (point pair i, tile tt): pair i is split during phase i from the raw rows fetched in phase i-1
          const int it = g / GPI, st = g % GPI, i = it / 4, tt = it % 4;
          if (st == 0) prep(s + 1, i, raw_lo(NX, i, tt), raw_hi(NX, i, tt), x0, x1);
          if (st == 1) { pl[tt][0][i] = pk_bf16(x0, x1); bsum[tt] = fmaf(x0 + x1, bias_w, bsum[tt]); }
          if (NPL == 3) {
            if (st == 2) { ra = x0 - bf16_lo(pl[tt][0][i]); rb = x1 - bf16_hi(pl[tt][0][i]); }
            if (st == 3) pl[tt][1][i] = pk_bf16(ra, rb);
            if (st == 4) { ra -= bf16_lo(pl[tt][1][i]); rb -= bf16_hi(pl[tt][1][i]); }
            if (st == 5) pl[tt][NPL - 1][i] = pk_bf16(ra, rb);
          } else {
            if (st == 2) pl[tt][1][i] = pk_bf16(x0 - bf16_lo(pl[tt][0][i]), x1 - bf16_hi(pl[tt][0][i]));
          }
          // the planes of tile tt are complete once pair 3 has been split: they are written in the units of the next item
          if (it >= 13 && st < NPL) write_tile(so1, tt - 1, st);
          if (tt == 3 && st == 1) gload(NX, s + 1 + RS, i, 0);          // the pair's last value was taken in the previous unit
          if (tt == 3 && st == 2) gload(NX, s + 1 + RS, i, 1);
        }
        // LDS reads of the next phase, one instruction per unit, half a phase ahead
        if (tb < 3 && gg >= PH / 2 && gg < PH / 2 + NPL) bp[(tb + 1) & 1][gg - PH / 2] = plane(so0, 2 + wb, tb + 1, gg - PH / 2);
        __builtin_amdgcn_sched_barrier(0);
      }
      if (MORE) {
#pragma unroll
        for (int p = 0; p < NPL; ++p) write_tile(so1, 3, p);
      }
    };
    using P0 = std::integral_constant<int, 0>; using P1 = std::integral_constant<int, 1>;
    {
      int s = 0;
      for (; s + 2 < nst; s += 2) { stage(s, BT{}, P0{}); stage(s + 1, BT{}, P1{}); }
      if (s + 1 < nst) { stage(s, BT{}, P0{}); stage(s + 1, BF{}, P1{}); }
      else stage(s, BF{}, P0{});
    }
  }
  (void)wa;
}

template <int NPL, bool PLAIN>
__device__ __forceinline__ void wgrad3p_body(const WgLaunch& L, float* lds) {
  float* plb = lds;
  const int tid = threadIdx.x, lane = tid & 63;
  const int w = __builtin_amdgcn_readfirstlane(tid >> 6);       // scalar: the load base addresses built from it stay in SGPRs
  const int wa = w >> 1, wb = w & 1, i32 = lane & 31, kg = lane >> 5;
  const WgTask& t = L.t[blockIdx.y];
  const int64_t chunk = blockIdx.x + L.chunk0;
  w3_acc_zero();
  float bsum[4] = {0.f, 0.f, 0.f, 0.f};
  const bool opB = (w >> 1) != 0;                          // this wave prepares a B quarter (else an A quarter)
  for (int jb = 0; jb < t.njobs; ++jb) {
    const WgJob& job = t.j[jb];
    const int64_t m_lo = chunk * WG_CH;
    const int64_t m_hi = (m_lo + WG_CH < job.m_count) ? m_lo + WG_CH : job.m_count;
    if (m_hi <= m_lo) continue;
    // this wave's operand in this job and chunk: point-major, blocked fp32 or packed 24-bit records (workgroup-uniform per operand, wave-uniform here)
    const bool blk = m_lo < (opB ? job.b_blk : job.a_blk);
    const int lay = __builtin_amdgcn_readfirstlane(!blk ? 0 : ((opB ? job.b_p24 : job.a_p24) != 0 ? 2 : 1));
    if (lay == 2) {
      if constexpr (NPL == 2) w3p_job<2, NPL, PLAIN>(t, job, jb, chunk, plb, bsum, w, lane);       // (the fp32-equivalent form keeps fp32 storage: mlp_common.h sdf_saves24)
    } else if (lay == 1) w3p_job<1, NPL, PLAIN>(t, job, jb, chunk, plb, bsum, w, lane);
    else w3p_job<0, NPL, PLAIN>(t, job, jb, chunk, plb, bsum, w, lane);
  }
  w3_mfma_drain();
  float* out = L.partials + chunk * L.chunk_stride;
  const int64_t toff = t.out_off + (int64_t)(wa * 128) * t.ldo + wb * 128;
#pragma unroll
  for (int ta = 0; ta < 4; ++ta)
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int ri = (r & 3) + 8 * (r >> 2) + 4 * kg;
      const f32x4 v = {w3_acc_read(16 * ta + r), w3_acc_read(16 * (ta + 4) + r), w3_acc_read(16 * (ta + 8) + r), w3_acc_read(16 * (ta + 12) + r)};
      *reinterpret_cast<f32x4*>(out + toff + (int64_t)(4 * ri + ta) * t.ldo + 4 * i32) = v;
    }
  if (t.has_bias != 0 && !opB) {           // waves 0 / 1 hold the column sums of A half 0 / 1
#pragma unroll
    for (int tt = 0; tt < 4; ++tt) bsum[tt] += __shfl_xor(bsum[tt], 32);
    if (kg == 0) *reinterpret_cast<f32x4*>(out + t.bias_off + (w & 1) * 128 + 4 * i32) = f32x4{bsum[0], bsum[1], bsum[2], bsum[3]};
  }
}

template <int NPL>
__global__ __launch_bounds__(256) void wgrad3p_kernel(WgLaunch L) {
  extern __shared__ __attribute__((aligned(16))) float lds[];
  const WgTask& t = L.t[blockIdx.y];
  const int64_t m_lo = (int64_t)(blockIdx.x + L.chunk0) * WG_CH;
  bool plain = t.relu_b == 0;
  for (int jb = 0; jb < t.njobs; ++jb) {
    const int64_t left = t.j[jb].m_count - m_lo;
    plain = plain && (left >= WG_CH || left <= 0 || left % W3_PTS == 0);
  }
  if (plain) wgrad3p_body<NPL, true>(L, lds);
  else wgrad3p_body<NPL, false>(L, lds);
}

// ---- ONE launch for every split-arithmetic task of a point range (round 6): the 256x256 blocks (wgrad3p_body) and the narrow / 128x128
// tasks (wgrad_narrow_body / wgrad_wide_body) as workgroups of the same grid, dispatched on the task's variant.  Two launches per range
// on one stream ran back to back -- the 250 narrow workgroups alone on their range's stream, not even one round of the chip, then the
// 475 block workgroups behind a kernel boundary; as one grid of 725 the short narrow workgroups fill the rounds of the long ones.  Both
// bodies are one workgroup per CU by their registers (460-512) and LDS either way.  Task order: longest first (see the launch site).
template <int NPL>
__global__ __launch_bounds__(256) void wgrad_all_kernel(WgLaunch L) {
  extern __shared__ __attribute__((aligned(16))) float lds[];
  const WgTask& t = L.t[blockIdx.y];
  if (t.variant == 4) {
    const int64_t m_lo = (int64_t)(blockIdx.x + L.chunk0) * WG_CH;
    bool plain = t.relu_b == 0;
    for (int jb = 0; jb < t.njobs; ++jb) {
      const int64_t left = t.j[jb].m_count - m_lo;
      plain = plain && (left >= WG_CH || left <= 0 || left % W3_PTS == 0);
    }
    if (plain) wgrad3p_body<NPL, true>(L, lds);
    else wgrad3p_body<NPL, false>(L, lds);
    return;
  }
  // (the narrow blocks always with three planes, as in wgrad_narrow_kernel<3>: bound by their operand reads)
  const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int64_t chunk = blockIdx.x + L.chunk0;
  if (t.variant == 0) wgrad_wide_body<3>(L, t, chunk, wave, lane, lds);
  else if (t.variant == 1) wgrad_narrow_body<1, 0, 3>(L, t, chunk, wave, lane, lds);
  else if (t.variant == 2) { if (t.j[0].a_p24) wgrad_narrow_body<0, 1, 3, true>(L, t, chunk, wave, lane, lds); else wgrad_narrow_body<0, 1, 3>(L, t, chunk, wave, lane, lds); }
  else { if (t.j[0].a_p24) wgrad_narrow_body<0, 2, 3, true>(L, t, chunk, wave, lane, lds); else wgrad_narrow_body<0, 2, 3>(L, t, chunk, wave, lane, lds); }
}

// ---- split-M reduction + weight-norm backward: one wave per weight row ------------------------------------
struct WnLayer {
  int64_t off_v, off_g, off_bias, blk_off, bias_blk_off;
  int32_t rows, cols, row0, ldo, rsplit, rbase, split, base1, valid0;
  float mult;
};
struct WnTab { WnLayer l[3 * I2SDF_MAX_LAYERS]; int32_t n; int32_t n_rows; };

__global__ __launch_bounds__(256) void wn_backward_kernel(WnTab tab, const float* __restrict__ params, const float* __restrict__ partials,
                                                           int n_chunks, int64_t chunk_stride, float* __restrict__ grad) {
  // one block per weight row; wave w sums chunks w, w+4, ... (4x the memory parallelism of one wave per row)
  __shared__ float s_dw[4][5 * 64];
  __shared__ float s_b[4];
  const int row = blockIdx.x;
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
  int e = 0;
  while (e + 1 < tab.n && row >= tab.l[e + 1].row0) ++e;
  const WnLayer& y = tab.l[e];
  const int r = row - y.row0;
  const int brow = r < y.rsplit ? r : r - y.rsplit + y.rbase;
  const float* v = params + y.off_v + (int64_t)r * y.cols;
  constexpr int MAXC = 5;      // cols <= 320
  // all columns of a chunk are loaded before any is added (5 independent loads in flight per lane, chunk loop unrolled)
  const float* src[MAXC];
  float sacc[MAXC];
#pragma unroll
  for (int q = 0; q < MAXC; ++q) {
    const int c = lane + 64 * q;
    const int kp = c < y.valid0 ? c : y.split + (c - y.base1);
    src[q] = (c < y.cols) ? partials + y.blk_off + (int64_t)brow * y.ldo + kp : nullptr;
    sacc[q] = 0.f;
  }
#pragma unroll 2
  for (int ch = wv; ch < n_chunks; ch += 4) {
    float x[MAXC];
#pragma unroll
    for (int q = 0; q < MAXC; ++q) x[q] = src[q] ? src[q][ch * chunk_stride] : 0.f;
#pragma unroll
    for (int q = 0; q < MAXC; ++q) sacc[q] += x[q];
  }
#pragma unroll
  for (int q = 0; q < MAXC; ++q) s_dw[wv][lane + 64 * q] = sacc[q];
  if (lane == 0) {
    float b = 0.f;
    const float* bs = partials + y.bias_blk_off + brow;
    for (int ch = wv; ch < n_chunks; ch += 4) b += bs[ch * chunk_stride];
    s_b[wv] = b;
  }
  __syncthreads();
  if (wv != 0) return;
  float dw[MAXC], vv[MAXC];
  float dot = 0.f, nrm2 = 0.f;
#pragma unroll
  for (int q = 0; q < MAXC; ++q) {
    const int c = lane + 64 * q;
    dw[q] = 0.f; vv[q] = 0.f;
    if (c < y.cols) {
      dw[q] = (s_dw[0][c] + s_dw[1][c] + s_dw[2][c] + s_dw[3][c]) * y.mult;
      vv[q] = v[c];
      dot = fmaf(dw[q], vv[q], dot);
      nrm2 = fmaf(vv[q], vv[q], nrm2);
    }
  }
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) { dot += __shfl_xor(dot, o); nrm2 += __shfl_xor(nrm2, o); }
  const float nrm = sqrtf(nrm2);
  const float g = params[y.off_g + r];
  const float gs = g / nrm, dn = dot / nrm2;
#pragma unroll
  for (int q = 0; q < MAXC; ++q) {
    const int c = lane + 64 * q;
    if (c < y.cols) grad[y.off_v + (int64_t)r * y.cols + c] = gs * (dw[q] - vv[q] * dn);
  }
  if (lane == 0) {
    grad[y.off_g + r] = dot / nrm;
    grad[y.off_bias + r] = s_b[0] + s_b[1] + s_b[2] + s_b[3];
  }
}

// ---------------------------------------------------------------------------------------------------------------
struct Src { const float* p; int ld; int w; int64_t blk = 0; int p24 = 0; };     // a [Mp][ld] matrix, the width used from it, blocked-prefix points, those as packed 24-bit records

struct TaskList {
  std::vector<WgTask> tasks;
  bool quads = false;        // emit one 256x256 task (bf16x3 kernel) where both operands are 256 wide
  // Emit the tiles of one weight block.  rows: segments of A (one per source, stacked in block rows at row0[i]);
  // cols: segments of B.  job 0 = (A0[i], B0[k]) over m0 points, job 1 = (A1[i], B1[k]) over m1 points (optional).
  void add_block(int64_t blk_off, int ldo, int64_t bias_off, const std::vector<Src>& A0, const std::vector<Src>& A1,
                 const std::vector<int>& row0, const std::vector<int64_t>& mA0, const std::vector<Src>& B0, const std::vector<Src>& B1,
                 int64_t m1, bool relu_b) {
    int col0 = 0;
    for (size_t k = 0; k < B0.size(); ++k) {
      if (quads && B0[k].w == 256) {
        for (size_t i = 0; i < A0.size(); ++i) {
          if (A0[i].w != 256) continue;
          WgTask t{};
          t.j[0] = WgJob{A0[i].p, B0[k].p, A0[i].ld, B0[k].ld, 256, 256, mA0[i], A0[i].blk, B0[k].blk, 0, 0, A0[i].p24, B0[k].p24};
          t.njobs = 1;
          if (!A1.empty() && A1[i].p != nullptr && !B1.empty()) {
            t.j[1] = WgJob{A1[i].p, B1[k].p, A1[i].ld, B1[k].ld, 256, 256, m1, A1[i].blk, B1[k].blk, 0, 0, A1[i].p24, B1[k].p24};
            t.njobs = 2;
          }
          t.relu_b = relu_b ? 1 : 0;
          t.has_bias = (k == 0) ? 1 : 0;
          t.rows_store = 256; t.cols_store = 256; t.ldo = ldo;
          t.out_off = blk_off + (int64_t)row0[i] * ldo + col0;
          t.bias_off = bias_off + row0[i];
          tasks.push_back(t);
        }
      }
      for (int ct = 0; ct * 128 < B0[k].w; ++ct) {
        for (size_t i = 0; i < A0.size(); ++i) {
          if (quads && B0[k].w == 256 && A0[i].w == 256) continue;      // covered by the quad task above
          for (int rt = 0; rt * 128 < A0[i].w; ++rt) {
            WgTask t{};
            const int aw = std::min(128, A0[i].w - rt * 128), bw = std::min(128, B0[k].w - ct * 128);
            t.j[0] = WgJob{A0[i].p + rt * 128, B0[k].p + ct * 128, A0[i].ld, B0[k].ld, aw, bw, mA0[i], A0[i].blk, B0[k].blk, rt * 128, ct * 128, A0[i].p24, B0[k].p24};
            t.njobs = 1;
            if (!A1.empty() && A1[i].p != nullptr && !B1.empty()) {
              t.j[1] = WgJob{A1[i].p + rt * 128, B1[k].p + ct * 128, A1[i].ld, B1[k].ld, aw, bw, m1, A1[i].blk, B1[k].blk, rt * 128, ct * 128, A1[i].p24, B1[k].p24};
              t.njobs = 2;
            }
            t.relu_b = relu_b ? 1 : 0;
            t.has_bias = (k == 0 && ct == 0) ? 1 : 0;
            t.rows_store = aw; t.cols_store = bw; t.ldo = ldo;
            t.out_off = blk_off + (int64_t)(row0[i] + rt * 128) * ldo + col0 + ct * 128;
            t.bias_off = bias_off + row0[i] + rt * 128;
            tasks.push_back(t);
          }
        }
      }
      col0 += B0[k].w;
    }
  }
};

void fill_wn(WnTab& tab, const NetPlan& np, int kind, int& row0) {
  const i2sdf_mlp_desc& d = np.d;
  const int PEC8 = (d.multires > 0 ? cdiv(d.d_in + 2 * d.d_in * d.multires, 8) : 0) * 8;
  const int PED = d.multires > 0 ? d.d_in + 2 * d.d_in * d.multires : d.d_in;
  for (int l = 0; l < d.n_lin; ++l) {
    WnLayer& y = tab.l[tab.n++];
    y.off_v = d.off_v[l]; y.off_g = d.off_g[l]; y.off_bias = d.off_bias[l];
    y.rows = d.out_dim[l]; y.cols = d.in_dim[l]; y.row0 = row0; row0 += y.rows;
    y.ldo = np.wg_cols[l];
    y.blk_off = np.wgrad_off[l];
    y.bias_blk_off = np.wgrad_off[l] + (int64_t)np.wg_rows[l] * np.wg_cols[l];
    y.rsplit = 1 << 30; y.rbase = 0; y.split = 0; y.base1 = 0; y.valid0 = y.cols; y.mult = 1.0f;
    const bool last = l == d.n_lin - 1;
    if (kind == 0) {
      if (l == d.skip_layer) { y.valid0 = y.cols - PED; y.split = d.hidden; y.base1 = y.cols - PED; y.mult = 0.70710678118654752440f; }
      if (last) { y.rsplit = 1; y.rbase = 32; }
    } else if (kind == 1) {
      if (l == 0) { y.valid0 = PED; y.split = PEC8; y.base1 = PED; }
    }
  }
}

}  // namespace

extern "C" int64_t i2sdf_wgrad_chunk_points(void) { return WG_CH; }

extern "C" int i2sdf_weight_grads(const i2sdf_plan* p, const i2sdf_train_buffers* tb, const float* params, float* partials,
                                  int64_t n_chunks_cap, float* grad_flat, void* stream) {
  if (!p || !tb || !params || !partials || !grad_flat) return I2SDF_EINVAL;
  const int64_t Mp = tb->Mp, Ms = tb->M_sdf, Mm = tb->M_main;
  if (Ms <= 0 || Mm < 0 || Mm > Ms || Mp < Ms || Mp % 128 != 0) return I2SDF_EINVAL;    // (the kernels read whole 16-point stages: into the padding rows)
  const int n_chunks = (int)((Ms + WG_CH - 1) / WG_CH);
  if (n_chunks > n_chunks_cap) return I2SDF_EWORKSPACE;
  hipStream_t st = (hipStream_t)stream;
  TaskList tl;
  tl.quads = p->wgrad_bf16x3 != 0;
  {  // ---- SDF net
    const NetPlan& np = p->sdf;
    const i2sdf_mlp_desc& d = np.d;
    const int L = d.n_lin, H = d.hidden, F = p->F, PEC8 = cdiv(d.in0, 8) * 8;
    const int64_t ls = Mp * H;
    const int64_t bs = (H == 256) ? sdf_blocked_points(p, Ms, Mp) : 0;      // leading points of hs / abars / gus / gas in the blocked layout
    const int s24 = (H == 256 && sdf_saves24(p) && bs == Mp) ? 1 : 0;       // abars / gus / gas as packed 24-bit records (hs stays fp32)
    for (int l = 0; l < L - 1; ++l) {
      std::vector<Src> B0, B1;
      if (l == 0) { B0 = {{tb->pe, PEC8, PEC8}}; B1 = {{tb->gpbar, PEC8, PEC8}}; }
      else {
        B0 = {{tb->hs + (l - 1) * ls, H, H, bs}}; B1 = {{tb->gus + l * ls, H, H, bs, s24}};
        if (l == d.skip_layer) { B0.push_back({tb->pe, PEC8, PEC8}); B1.push_back({tb->gpbar, PEC8, PEC8}); }
      }
      tl.add_block(np.wgrad_off[l], np.wg_cols[l], np.wgrad_off[l] + (int64_t)np.wg_rows[l] * np.wg_cols[l],
                   {{tb->gas + l * ls, H, H, bs, s24}}, {{tb->abars + l * ls, H, H, bs, s24}}, {0}, {Ms}, B0, B1, Ms, false);
    }
    {
      const int l = L - 1;
      std::vector<Src> A0 = {{tb->ga_last4, 4, 4}}, A1 = {{tb->ones4, 4, 4}};
      std::vector<int> row0 = {0};
      std::vector<int64_t> mA0 = {Ms};
      if (F > 0 && Mm > 0 && tb->fbar) { A0.push_back({tb->fbar, F, F}); A1.push_back({nullptr, 0, 0}); row0.push_back(32); mA0.push_back(Mm); }
      tl.add_block(np.wgrad_off[l], np.wg_cols[l], np.wgrad_off[l] + (int64_t)np.wg_rows[l] * np.wg_cols[l], A0, A1, row0, mA0,
                   {{tb->hs + (L - 2) * ls, H, H, bs}}, {{tb->gus + (L - 1) * ls, H, H, bs}}, Ms, false);       // (gus[L-1] stays fp32: x3.h X3Sweep2Src)
    }
  }
  if (Mm > 0 && tb->gar) {  // ---- radiance net
    const NetPlan& np = p->rgb;
    const i2sdf_mlp_desc& d = np.d;
    const int L = d.n_lin, H = d.hidden, F = p->F, PV8 = cdiv(d.in0 - F, 8) * 8;
    const int64_t ls = Mp * H;
    const int64_t br = (H == 256) ? rgb_blocked_points(p, Mm, Mp) : 0;     // leading points of rs / gar in the blocked layout
    for (int l = 0; l < L; ++l) {
      std::vector<Src> B0;
      if (l == 0) B0 = {{tb->pev, PV8, PV8}, {tb->feat, F, F}};
      else B0 = {{tb->rs + (l - 1) * ls, H, H, br}};
      std::vector<Src> A0;
      if (l < L - 1) A0 = {{tb->gar + l * ls, H, H, br}};
      else A0 = {{tb->ga_last_rgb, 4, 4}};
      tl.add_block(np.wgrad_off[l], np.wg_cols[l], np.wgrad_off[l] + (int64_t)np.wg_rows[l] * np.wg_cols[l], A0, {}, {0}, {Mm}, B0, {}, 0,
                   false);
    }
  }
  const bool has_light = p->light.d.n_lin > 0 && Mm > 0 && tb->gal0;
  size_t light_first = tl.tasks.size();
  if (has_light) {  // ---- light head: l=0: A = G(a_0), B = relu(feature) ; l=1: A = G(a_1), B = softplus acts
    const NetPlan& np = p->light;
    const int H = np.d.hidden, F = p->F;
    tl.add_block(np.wgrad_off[0], np.wg_cols[0], np.wgrad_off[0] + (int64_t)np.wg_rows[0] * np.wg_cols[0], {{tb->gal0, H, H}}, {}, {0}, {Mm},
                 {{tb->feat, F, F}}, {}, 0, true);
    tl.add_block(np.wgrad_off[1], np.wg_cols[1], np.wgrad_off[1] + (int64_t)np.wg_rows[1] * np.wg_cols[1], {{tb->gal_last, 4, 4}}, {}, {0},
                 {Mm}, {{tb->hl, H, H}}, {}, 0, false);
  }
  (void)light_first;
  // group the tiles by operand shape (narrow operands run 4x / 2x fewer MFMAs per point pair), longest first inside a group
  auto variant = [](const WgTask& x) {
    return x.j[0].a_w == 256 ? 4 : (x.j[0].a_w <= 32 ? 1 : (x.j[0].b_w <= 32 ? 2 : (x.j[0].b_w <= 64 ? 3 : 0)));
  };
  for (WgTask& x : tl.tasks) x.variant = variant(x);
  hipStream_t st_main = st;
  int c_lo = 0, c_hi = n_chunks;             // chunk range of the launches below (point ranges: one range at a time)
  auto launch = [&](const std::vector<WgTask>& sel, int var) {
    for (size_t off = 0; off < sel.size(); off += MAX_TASKS) {
      WgLaunch L{};
      L.n = (int32_t)std::min<size_t>(MAX_TASKS, sel.size() - off);
      for (int i = 0; i < L.n; ++i) L.t[i] = sel[off + i];
      L.chunk_stride = p->wgrad_floats; L.partials = partials;
      L.chunk0 = c_lo;
      dim3 grid((unsigned)(c_hi - c_lo), (unsigned)L.n);
      if (var == 5) {          // split arithmetic: every task of the range in one grid (wgrad_all_kernel)
        constexpr int LB = W3P_LDS_BYTES > WGN_LDS_BYTES ? W3P_LDS_BYTES : WGN_LDS_BYTES;
        if (p->wgrad_bf16x2) {
          (void)hipFuncSetAttribute((const void*)wgrad_all_kernel<2>, hipFuncAttributeMaxDynamicSharedMemorySize, LB);
          wgrad_all_kernel<2><<<grid, 256, LB, st>>>(L);
        } else {
          (void)hipFuncSetAttribute((const void*)wgrad_all_kernel<3>, hipFuncAttributeMaxDynamicSharedMemorySize, LB);
          wgrad_all_kernel<3><<<grid, 256, LB, st>>>(L);
        }
      } else if (var == 4) {
        if (p->wgrad_bf16x2) {
          (void)hipFuncSetAttribute((const void*)wgrad3p_kernel<2>, hipFuncAttributeMaxDynamicSharedMemorySize, W3P_LDS_BYTES);
          wgrad3p_kernel<2><<<grid, 256, W3P_LDS_BYTES, st>>>(L);
        } else {
          (void)hipFuncSetAttribute((const void*)wgrad3p_kernel<3>, hipFuncAttributeMaxDynamicSharedMemorySize, W3P_LDS_BYTES);
          wgrad3p_kernel<3><<<grid, 256, W3P_LDS_BYTES, st>>>(L);
        }
      } else if (var == 0) {
        wgrad_kernel<0, 0><<<grid, 64, 0, st>>>(L);
      } else {
        // the narrow blocks (PE columns of layer 0 / the skip layer, the output rows): bf16 split with THREE planes whenever the 256x256
        // blocks run in split arithmetic, else fp32-input MFMA.  Also under I2SDF_OPT_WGRAD_BF16X2: the kernel is bound by its operand
        // reads (4.0-4.7 TB/s, matrix pipe 8-16 % busy), so the third plane costs little and these few blocks keep fp32-equivalent
        // gradients (measured: 107 us per launch with two planes, 127 us with three: +0.03 ms per step).
        if (p->wgrad_bf16x3) {
          (void)hipFuncSetAttribute((const void*)wgrad_narrow_kernel<3>, hipFuncAttributeMaxDynamicSharedMemorySize, WGN_LDS_BYTES);
          wgrad_narrow_kernel<3><<<grid, 256, WGN_LDS_BYTES, st>>>(L);
        } else {
          (void)hipFuncSetAttribute((const void*)wgrad_narrow_kernel<0>, hipFuncAttributeMaxDynamicSharedMemorySize, WGN_LDS_BYTES);
          wgrad_narrow_kernel<0><<<grid, 256, WGN_LDS_BYTES, st>>>(L);
        }
      }
    }
  };
  std::vector<WgTask> sel_narrow, sel_blocks[2];
  // in split arithmetic the 128 x 128 tiles of blocks that are not 256 x 256 (variant 0: the light-mask head) join the 4-wave launch,
  // in front (the longest tasks): wgrad_wide_body
  const bool wide_in_narrow = p->wgrad_bf16x3 != 0;
  if (wide_in_narrow)
    for (const WgTask& x : tl.tasks) if (x.variant == 0) sel_narrow.push_back(x);
  for (int var : {3, 1, 2})      // all narrow tiles in one launch, the longest (most MFMAs per point pair, two jobs) first
    for (int nj = 2; nj >= 1; --nj)
      for (const WgTask& x : tl.tasks) if (x.variant == var && x.njobs == nj) sel_narrow.push_back(x);
  for (int v = 0; v < 2; ++v) {
    for (const WgTask& x : tl.tasks) if (x.variant == (v == 0 ? 4 : 0) && !(v == 1 && wide_in_narrow)) sel_blocks[v].push_back(x);
    std::stable_sort(sel_blocks[v].begin(), sel_blocks[v].end(), [](const WgTask& x, const WgTask& y) { return x.njobs > y.njobs; });
  }
  // split arithmetic: narrow tasks + 256x256 blocks as ONE grid per range (I2SDF_WGRAD_MERGED=0: the two launches of rounds 4-5, for A/B runs)
  static const bool merged_env = [] { const char* e = getenv("I2SDF_WGRAD_MERGED"); return !(e && e[0] == '0'); }();
  const bool merged = merged_env && p->wgrad_bf16x3 != 0 && sel_narrow.size() + sel_blocks[0].size() <= (size_t)MAX_TASKS && !sel_blocks[0].empty();
  std::vector<WgTask> sel_all;
  // order inside the grid: workgroups are dispatched task by task.  LONGEST FIRST -- the two-job 256x256 tasks, the one-job ones, then the short
  // narrow tasks, which fill the last round: i2sdf_weight_grads 0.95-0.96 -> 0.91-0.92 ms against narrow-first (three interleaved repetitions in
  // one GPU call, profiles/r6_launch_chain_ab.txt); I2SDF_WGRAD_BLOCKS_FIRST=0 restores narrow-first for A/B runs
  static const bool blocks_first = [] { const char* e = getenv("I2SDF_WGRAD_BLOCKS_FIRST"); return !(e && e[0] == '0'); }();
  if (merged) {
    if (blocks_first) { sel_all = sel_blocks[0]; sel_all.insert(sel_all.end(), sel_narrow.begin(), sel_narrow.end()); }
    else { sel_all = sel_narrow; sel_all.insert(sel_all.end(), sel_blocks[0].begin(), sel_blocks[0].end()); }
  }
  PartRun pr;
  if (i2sdf_parts_on(p) && i2sdf_parts_begin(p, st_main, Ms, &pr)) {
    // point ranges (plan.h: PartRun): the GEMMs of a range's chunks on the range's stream, behind that range's backward sweeps
    for (int q = 0; q < pr.n; ++q) {
      if (pr.hi[q] <= pr.lo[q]) continue;
      c_lo = (int)(pr.lo[q] / WG_CH); c_hi = (int)((pr.hi[q] + WG_CH - 1) / WG_CH);
      st = pr.st[q];
      if (merged) launch(sel_all, 5);
      else {
        launch(sel_narrow, 1);               // (the narrow blocks behind the 256x256 ones instead: +0.02 ms, measured)
        launch(sel_blocks[0], 4);
      }
      launch(sel_blocks[1], 0);
    }
    st = st_main; c_lo = 0; c_hi = n_chunks;
    i2sdf_parts_join_all(p, st_main);                   // the reduction below reads every chunk's partials (also ends a chain's ranges)
  } else {
    hipStream_t side = i2sdf_tail_fork(p, st_main);
    st = side;
    launch(sel_narrow, 1);
    st = st_main;
    launch(sel_blocks[0], 4);
    launch(sel_blocks[1], 0);
    i2sdf_tail_join(p, st_main, side);
  }
  WnTab tab{};
  int row0 = 0;
  fill_wn(tab, p->sdf, 0, row0);
  if (Mm > 0 && tb->gar) fill_wn(tab, p->rgb, 1, row0);
  if (has_light) fill_wn(tab, p->light, 2, row0);
  tab.n_rows = row0;
  wn_backward_kernel<<<row0, 256, 0, st>>>(tab, params, partials, n_chunks, p->wgrad_floats, grad_flat);
  return i2sdf_hip_check(hipGetLastError(), "weight_grads launch");
}

