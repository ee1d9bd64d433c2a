// Host-side plan: where every packed weight stream lives and how it is produced from the flat parameter
// buffer.  The kernels walk the streams linearly; the order of ops inside a stream is fixed by the
// kernel's own static structure (see mlp_fwd.hip), and build_*_stream() below must emit the same order.
#pragma once
#include <stdint.h>
#include <vector>
#include "../../include/i2sdf.h"
#include "x3h.h"

namespace i2sdf {

enum SegType : int32_t { SEG_ZERO = 0, SEG_BIAS = 1, SEG_WFWD = 2, SEG_WBWD = 3, SEG_ROWVEC = 4, SEG_SCALAR = 5,
                         SEG_WFWD3 = 6 /* bf16x3 split planes, K-outer order (x3.h) */,
                         SEG_WBWD3 = 7 /* transposed weights, bf16x3 split planes, K-outer order */,
                         // the 16-point-wave family (x3h.h): 16-row tiles, 32-wide k-chunks
                         SEG_BIAS_H = 8, SEG_WFWD3H = 9, SEG_WBWD3H = 10, SEG_ROWVEC_H = 11,
                         SEG_WFWD2H = 12 /* as SEG_WFWD3H with the two leading split planes only (the sampler's bf16x2 passes) */ };

// Column map from a padded register-space index to a source column of weight_v:
//   kp <  split : kp < valid0 ? base0 + kp : none
//   kp >= split : (kp-split) < valid1 ? base1 + (kp-split) : none
struct ColMap { int32_t split, base0, valid0, base1, valid1; };

struct Seg {
  int64_t chunk0;        // first chunk of the segment inside the pack buffer
  int32_t nchunks;       // chunks covered (including its zero padding)
  int32_t type;
  int64_t off_v, off_bias;
  int32_t scale_off;     // index of row 0 of this layer in the row-scale array
  int32_t rows, cols;    // logical shape of weight_v
  int32_t row_off;       // first logical row this segment covers (e.g. 1 for the feature rows)
  int32_t nrows;         // rows covered from row_off (rows beyond are zero)
  int32_t NT, KC;        // tile geometry: WFWD chunk (nt,kc); WBWD chunk (kt,nc) with NT=#kt, KC=#nc
  int32_t used;          // chunks that carry data (rest is zero padding)
  ColMap cm;
  float mult;
};

struct LayerGeom {       // per net, per layer: padded geometry shared by pack + kernels
  int32_t NT;            // output tiles (32 rows each)
  int32_t KC;            // input chunks (8 cols each)
};

struct NetPlan {
  i2sdf_mlp_desc d;
  int32_t scale_off[I2SDF_MAX_LAYERS];
  int64_t fwd_chunk0 = 0, fwd_chunks = 0;      // forward stream
  int64_t rev_chunk0 = 0, rev_chunks = 0;      // transposed / reverse stream
  int64_t rev_wsdf_chunk = 0;                  // where the igrad chain starts inside rev (sdf net)
  int64_t fwd3_chunk0 = 0, fwd3_chunks = 0;    // bf16x3 forward stream: hidden layers, sdf row, feature rows (sdf net, x3.h)
  int64_t rev3_chunk0 = 0, rev3_chunks = 0;    // bf16x3 reverse stream: [w_sdf][W_feat^T][w_sdf][W_{L-2}^T] ... [W_0^T]
  int64_t rev3_wsdf_chunk = 0;                 // where the d sdf/dx chain starts inside it
  int64_t fwd3h_chunk0 = 0, fwd3h_chunks = 0;  // bf16x3 streams of the 16-point-wave kernels (x3h.h): sdf net forward [hidden layers, sdf row, feature rows];
  int64_t rev3h_chunk0 = 0, rev3h_chunks = 0;  // radiance net forward and reverse
  int64_t fwd2h_chunk0 = 0, fwd2h_chunks = 0;  // sdf net, hidden layers + sdf row with TWO split planes (I2SDF_OPT_SAMPLER_BF16X2)
  int64_t wgrad_off[I2SDF_MAX_LAYERS];         // offset (floats) of layer l's [rowsP x colsP] block in the wgrad buffer
  int32_t wg_rows[I2SDF_MAX_LAYERS], wg_cols[I2SDF_MAX_LAYERS];   // padded shape of that block
};

}  // namespace i2sdf

struct i2sdf_plan {
  i2sdf_net_desc desc;
  i2sdf::NetPlan sdf, rgb, light;
  std::vector<i2sdf::Seg> segs;
  i2sdf::Seg* d_segs = nullptr;      // device copy
  int32_t n_segs = 0;
  int32_t n_scale = 0;               // total rows over all layers of all nets
  int64_t scale_floats = 0;          // region [0, scale_floats) of the pack buffer holds g/||v||
  int64_t total_chunks = 0;          // chunks after the scale region (+1 stage of slack for the DMA look-ahead)
  int64_t wgrad_floats = 0;
  int32_t H = 0, F = 0;              // sdf hidden width / feature size
  int32_t n_cu = 256;                // compute units of the device the plan was created on (hipDeviceProp; 256 on MI355X)
  int32_t rgb_bf16x3 = 0;            // I2SDF_OPT_RGB_BF16X3: radiance forward / backward (full workgroups) in bf16x3 split arithmetic
  int32_t sdf_bwd_bf16x3 = 0;        // I2SDF_OPT_SDF_BWD_BF16X3: SDF backward sweeps (full workgroups) in bf16x3 split arithmetic
  int32_t train_fwd_bf16x3 = 0;      // I2SDF_OPT_TRAIN_FWD_BF16X3: SDF forward + d sdf/dx kernel in bf16x3 split arithmetic
  int32_t wgrad_bf16x3 = 0;          // I2SDF_OPT_WGRAD_BF16X3: full 256x256 weight-gradient blocks in bf16x3 split arithmetic
  int32_t sdf_fwd_bf16x3 = 0;        // i2sdf_plan_set_option(I2SDF_OPT_SDF_FWD_BF16X3): sdf-only forward in bf16x3 split arithmetic
  int32_t sampler_bf16x2 = 0;        // I2SDF_OPT_SAMPLER_BF16X2: the sampler's sdf-only passes with two split planes / three products
  i2sdf_exchange exchange{nullptr, nullptr};   // i2sdf_plan_set_exchange: small data-parallel exchanges (copied)
  int32_t dp_flags = 0;              // I2SDF_DP_GLOBAL_SAMPLER
  int32_t wgrad_bf16x2 = 0;          // I2SDF_OPT_WGRAD_BF16X2: 256x256 weight-gradient blocks with two split planes / three products
  int32_t blocked_saves = 0;         // I2SDF_OPT_BLOCKED_SAVES: saved tensors of the bf16x3 full workgroups in the blocked layout (mlp_common.h)
  int32_t saves24 = 0;               // I2SDF_OPT_SAVES24: abars / gus / gas as packed 24-bit records (mlp_common.h: sdf_saves24 says when it takes effect)
  int32_t tail_overlap = 0;          // I2SDF_OPT_TAIL_OVERLAP: split-K tail workgroups on a side stream, concurrent with the full ones
  // side stream + fork/join events of the tail overlap, created on first use (entry points take a const plan)
  mutable hipStream_t side = nullptr;
  mutable hipEvent_t ev_fork = nullptr, ev_join = nullptr;
  // I2SDF_OPT_PARTS (see PartRun below): number of point ranges the per-point entry points are cut into (0 / 1 = off)
  int32_t parts = 0;
  mutable hipStream_t part_st[I2SDF_MAX_PARTS - 1] = {nullptr, nullptr, nullptr};
  mutable hipEvent_t part_ev[I2SDF_MAX_PARTS] = {nullptr, nullptr, nullptr, nullptr};   // [0] fork event, [q] join event of part q
  mutable int32_t chain_active = 0;          // between i2sdf_chain_begin and i2sdf_chain_end
  mutable int64_t chain_M = 0;               // the point count the active chain's ranges were cut from
};

// ---- point ranges ("parts") ---------------------------------------------------------------------------------------
// A launch over M points of 128-point workgroups on 256 CUs runs ceil(M/128/256) rounds, the last one partly empty, and the
// next kernel of the chain (forward -> d sdf/dx -> radiance -> ... ) cannot start before it has drained.  With I2SDF_OPT_PARTS
// = n the point batch is cut into n ranges at multiples of the weight-gradient chunk (1024 points); range q runs its whole
// chain of kernels on its own stream (range 0 on the caller's, the others on streams owned by the plan), so the ranges drift
// apart and the partly empty last round of one kernel is filled by workgroups of another range's next kernel.  No split-K tail
// workgroups in this mode: every point goes through the bf16x3 kernels, every saved tensor is blocked throughout.
//   parts_begin : fills `pr` with the ranges of a batch of M points and the stream of each.  Outside a chain it forks the side
//                 streams from `st` (they see everything enqueued on `st` so far); inside a chain (i2sdf_chain_begin) the streams
//                 are already forked and stay un-joined between entry points.
//   parts_end   : outside a chain joins the side streams back into `st` (the entry point returns stream-ordered on `st`).
struct PartRun {
  int n = 1;
  hipStream_t st[I2SDF_MAX_PARTS];
  int64_t lo[I2SDF_MAX_PARTS], hi[I2SDF_MAX_PARTS];     // point range of part q (lo a multiple of 1024; hi capped by M)
  bool own = false;                                      // this call forked, this call joins
};
bool i2sdf_parts_begin(const i2sdf_plan* p, hipStream_t st, int64_t M, PartRun* pr);
void i2sdf_parts_end(const i2sdf_plan* p, hipStream_t st, PartRun* pr);
void i2sdf_parts_join_all(const i2sdf_plan* p, hipStream_t st);        // `st` waits for every side stream (also inside a chain)
void i2sdf_parts_fence(const i2sdf_plan* p, hipStream_t st);           // every side stream waits for what is enqueued on `st`
// points per split-M chunk of the weight-gradient GEMMs == alignment of the point ranges (a chunk never straddles two ranges)
#ifndef I2SDF_WG_CH
#define I2SDF_WG_CH 2048      // (round 4: 2048 instead of 1024 halves the per-chunk partial sums -- wn_backward reads 220 instead of 440 MB -- step -0.04 ms)
#endif
inline bool i2sdf_parts_on(const i2sdf_plan* p) { return p->parts >= 2; }
// An entry point that does NOT cut its batch into ranges (its kernel family is not on the ranged path, e.g. rgb_bf16x3 off while the SDF
// flags are on) called inside a chain: the side streams still hold un-joined work of the previous entry point, and the next one will run
// on them again.  The guard joins every range into `st` before the entry point's launches and fences the side streams behind them.
struct ChainGuard {
  const i2sdf_plan* p; hipStream_t st; bool on;
  ChainGuard(const i2sdf_plan* p_, hipStream_t st_, bool ranged) : p(p_), st(st_), on(!ranged && p_->chain_active != 0) {
    if (on) i2sdf_parts_join_all(p, st);
  }
  ~ChainGuard() { if (on) i2sdf_parts_fence(p, st); }
};

// Fork: returns the stream the split-K tail of an entry point should be launched on -- the plan's side stream, ordered after
// everything already enqueued on `st`, or `st` itself when the overlap is off.  Join: `st` waits for the side stream.
// Launch order at the call sites: fork, tail (few long workgroups), full workgroups on `st`, join.
hipStream_t i2sdf_tail_fork(const i2sdf_plan* p, hipStream_t st);
void i2sdf_tail_join(const i2sdf_plan* p, hipStream_t st, hipStream_t side);
