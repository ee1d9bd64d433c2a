// Argument blocks of the SDF training kernels (shared by the fp32 MFMA kernels and their bf16x3 twins in mlp_x3.hip).
#pragma once
#include "mlp_common.h"

struct SdfTrainFwdArgs {
  const float* fwd; int n_fwd;
  const float* rev; int n_rev;          // starts at the w_sdf row vector
  int L, skip;
  i2sdf::PointSpec pts;
  int64_t M, Mp;
  float* sdf;                           // (M)
  float* feat;                          // (Mp, F)
  float* grad;                          // (M,3) or nullptr
  float* hs;                            // (L-1, Mp, H)  h_1..h_{L-1}   or nullptr (no saves: eval)
  float* abars;                         // (L-1, Mp, H)  abar_0..abar_{L-2} or nullptr
  float* pe_save;                       // (Mp, PEC*8) PE(x) for the weight-gradient GEMMs, or nullptr
  int kcs = 16;                         // layout of hs / abars for the points of THIS launch: 16 point-major, 512 blocked (mlp_common.h)
  int p24 = 0;                          // abars as packed 24-bit records (mlp_common.h: sdf_saves24; kcs is then 512 and applies to hs)
  int wg0 = 0;                          // first 128-point workgroup of this launch (point ranges, plan.h: PartRun)
  int64_t ldf = 0;                      // row stride of `feat` in floats (0 = F: the training workspaces; i2sdf_sdf_forward passes the caller's ld_feat)
};

struct SdfBwdArgs {
  const float* fwd; int n_fwd;          // forward stream up to (excluding) the last layer
  const float* rev; int n_rev;          // reverse stream from W_feat^T down to W_1^T
  int L, skip;
  i2sdf::PointSpec pts;
  int64_t M, Mp;
  const float* hs; const float* abars;  // from sdf_train_fwd
  const float* sbar;                    // (M) d loss / d sdf            (nullptr = 0)
  const float* fbar; int64_t m_fbar;    // (Mp,F) d loss / d feature, rows >= m_fbar are zero (nullptr = 0)
  const float* nbar;                    // (M,3) d loss / d grad         (nullptr = 0)
  float* gus;                           // (L, Mp, H)  G(hbar_l): slot l = h-part of G(ubar_l), l = 1..L-1 (slot 0 unused)
  float* gpbar;                         // (Mp, PEC*8) G(pbar)
  float* gas;                           // (L-1, Mp, H) G(a_l), l = 0..L-2 (the fp32 / split-K kernels of mlp_bwd.hip park G2(a_l) here between their sweeps; the bf16x3 sweeps do not: x3.h)
  float* ga_last4;                      // (Mp,4) {sbar,0,0,0}: A operand of the last layer's sdf-row weight gradient
  float* ones4;                         // (Mp,4) {1,0,0,0}
  int kcs = 16;                         // layout of hs / abars / gus / gas for the points of this launch
  int p24 = 0;                          // abars / gus / gas as packed 24-bit records (kcs is then 512 and applies to hs)
  int wg0 = 0;                          // first 128-point workgroup of this launch
};

struct RgbFwdArgs {
  const float* fwd; int n_fwd; int L;
  const float* dirs; int n_per_ray;     // view dir of point m = dirs[m / n_per_ray]
  const float* feat;                    // (Mp, F)
  int64_t M, Mp;
  float* rgb;                           // (M,3)
  float* rs;                            // (L-1, Mp, H) post-ReLU activations r_1..r_{L-1}, or nullptr
  float* pev_save;                      // (Mp, PECV*8) PE(view dir), or nullptr
  int kcs = 16;                         // layout of rs for the points of this launch
  int wg0 = 0;                          // first 128-point workgroup of this launch
};

struct RgbBwdArgs {
  const float* rev; int n_rev; int L;
  int64_t M, Mp;
  const float* rgb;         // (M,3) forward output
  const float* rgb_bar;     // (M,3)
  const float* rs;          // (L-1, Mp, H)
  float* gar;               // (L-1, Mp, H)  G(a_l), l = 0..L-2
  float* ga_last;           // (Mp, 4)       G(a_{L-1}) (3 used)
  float* fbar;              // (Mp, F)
  int kcs = 16;             // layout of rs / gar for the points of this launch
  int wg0 = 0;              // first 128-point workgroup of this launch
};

// bf16x3 kernels: launch over `grid` workgroups of 128 points -- four 32-point waves (mlp_x3.hip: d sdf/dx chain, sweeps) or eight 16-point
// waves (mlp_x3h.hip: forward with saves, radiance net)
void i2sdf_launch_igrad3(const SdfTrainFwdArgs& a, unsigned grid, hipStream_t st);
void i2sdf_launch_sdf_bwd3(const SdfBwdArgs& a, unsigned grid, hipStream_t st);
void i2sdf_launch_train_fwd3h(const SdfTrainFwdArgs& a, unsigned grid, hipStream_t st);
void i2sdf_launch_rgb_fwd3h(const RgbFwdArgs& a, unsigned grid, hipStream_t st);
void i2sdf_launch_rgb_bwd3h(const RgbBwdArgs& a, unsigned grid, hipStream_t st);
// light-mask head forward on 16-point waves (mlp_x3h.hip): lm = sigmoid(W1 softplus100(W0 relu(feature) + b0) + b1), hl = the softplus activations
struct LightFwd3hArgs {
  const float* fwd; int n_fwd;
  const float* feat;                    // (Mp, F)
  int64_t M;
  float* lm;                            // (M)
  float* hl;                            // (Mp, HL) point-major, or nullptr
};
void i2sdf_launch_light_fwd3h(const LightFwd3hArgs& a, unsigned grid, hipStream_t st);
