// SDF network forward without gradient: ImplicitNetwork.forward / get_sdf_vals
// (model/network/mlp.py:84-105,145-151).  MFMA-bound: 2*524544 FLOP per point at synthetic.yml shapes.
#include "epi.h"
#include "mlp_args.h"

using namespace i2sdf;

int i2sdf_hip_check(hipError_t e, const char* what);
// bf16x3 variant of the sdf-only forward: sdf_fwd3_kernel lives in mlp_x3.hip (the translation unit with the lifted unroll cap)
void i2sdf_launch_sdf_fwd3(int H, const float* stream, int n_stages, int L, int skip, const PointSpec& ps, const int* skip_flag, int64_t M,
                           float* sdf_out, unsigned grid, hipStream_t st);

void i2sdf_launch_sdf_fwd3h(const float* stream, int n_stages, int L, int skip, const PointSpec& ps, const int* skip_flag, int64_t M, float* sdf_out,
                            int planes, hipStream_t st);

namespace {

template <int H, int F, int LF, bool FULL>
__global__ __launch_bounds__(256) void sdf_fwd_kernel(const float* __restrict__ stream, int n_stages, int L, int skip, PointSpec ps,
                                                       const int* __restrict__ skip_flag, int64_t M, float* __restrict__ sdf_out,
                                                       float* __restrict__ feat_out, int64_t ld_feat) {
  constexpr int NT = H / 32, KC = H / 8, PEC = PE<LF>::PEC;
  extern __shared__ __attribute__((aligned(16))) float lds[];
  if (skip_flag != nullptr && skip_flag[0] != 0) return;      // batch already converged (sampler)
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, hi = lane >> 5;
  const int64_t m = ((int64_t)blockIdx.x * 4 + wave) * 32 + (lane & 31);
  const bool valid = m < M;
  const int64_t mc = valid ? m : M - 1;
  float px, py, pz;
  fetch_point(ps, mc, px, py, pz);
  float pe[PEC * 4];
  {
    float full[PEC * 8];
    pe_full<LF>(px, py, pz, full);
    to_b_layout<PEC>(full, pe, hi);
  }
  WStream ws;
  ws.begin(stream, lds, n_stages, tid);
  f32x16 acc[NT];
  float h[NT * 16];
  SoftplusEpi sp{nullptr, hi, valid};
  dense_op_epi<NT, PEC, NT * 4, 0, 0, SoftplusEpi>(ws, pe, acc, sp, tid);
  commit_tiles<NT>(acc, h);
  for (int l = 1; l < L - 1; ++l) {
    if (l == skip) {
      constexpr float rs2 = RS2;
      float u[(KC + PEC) * 4];
#pragma unroll
      for (int i = 0; i < KC * 4; ++i) u[i] = h[i] * rs2;
#pragma unroll
      for (int i = 0; i < PEC * 4; ++i) u[KC * 4 + i] = pe[i] * rs2;
      dense_op_epi<NT, KC + PEC, NT * 4, 0, 0, SoftplusEpi>(ws, u, acc, sp, tid);
    } else {
      dense_op_epi<NT, KC, NT * 4, 0, 0, SoftplusEpi>(ws, h, acc, sp, tid);
    }
    commit_tiles<NT>(acc, h);
  }
  float s[1];
  rowvec_op<1, KC>(ws, h, s, tid);
  if (sdf_out != nullptr && valid && hi == 0) sdf_out[m] = s[0];
  if (FULL) {
    constexpr int FT = F / 32;
    f32x16 fa[FT];
    dense_op<FT, KC, 0>(ws, h, fa, tid);
    store_tile<FT>(feat_out + mc * ld_feat, hi, valid, fa);
  }
}

template <int H, int F, int LF>
int launch_sdf_fwd(const i2sdf_plan* p, const float* packed, PointSpec points, const int* skip_flag, int64_t M, float* sdf_out,
                   float* feat_out, int64_t ld_feat, hipStream_t st, bool sampler_pass = false) {
  const i2sdf_mlp_desc& d = p->sdf.d;
  const float* stream = packed + p->scale_floats + p->sdf.fwd_chunk0 * CHUNK_FLOATS;
  const bool full = feat_out != nullptr;
  if (!full && sdf_out != nullptr && p->sdf_fwd_bf16x3 && H == 256 && p->sdf.fwd3h_chunks > 0) {
    // 256-wide nets: 16-point waves, two per SIMD (x3h.h); the sampler's passes with two split planes under I2SDF_OPT_SAMPLER_BF16X2
    const int PL = (sampler_pass && p->sampler_bf16x2 && p->sdf.fwd2h_chunks > 0) ? 2 : 3;
    const float* s3 = packed + p->scale_floats + (PL == 2 ? p->sdf.fwd2h_chunk0 : p->sdf.fwd3h_chunk0) * CHUNK_FLOATS;
    const int ns3 = sdf_fwd3h_stages(H, PE<LF>::DIM, d.n_lin, d.skip_layer > 0, PL);
    i2sdf_launch_sdf_fwd3h(s3, ns3, d.n_lin, d.skip_layer, points, skip_flag, M, sdf_out, PL, st);
    return i2sdf_hip_check(hipGetLastError(), "sdf_forward (bf16x3, 16-point waves) launch");
  }
  if (!full && sdf_out != nullptr && p->sdf_fwd_bf16x3 && H == 64 && p->sdf.fwd3_chunks > 0) {
    const float* s3 = packed + p->scale_floats + p->sdf.fwd3_chunk0 * CHUNK_FLOATS;
    const int ns3 = sdf_fwd3_stages(H, PE<LF>::DIM, d.n_lin, d.skip_layer > 0);
    i2sdf_launch_sdf_fwd3(H, s3, ns3, d.n_lin, d.skip_layer, points, skip_flag, M, sdf_out, (unsigned)((M + PTS_PER_WG - 1) / PTS_PER_WG), st);
    return i2sdf_hip_check(hipGetLastError(), "sdf_forward (bf16x3) launch");
  }
  if (full && p->sdf_fwd_bf16x3 && H == 256 && F == 256 && p->sdf.fwd3h_chunks > 0) {
    // [sdf | feature] rows (ImplicitNetwork.forward: the meshing callers, model/eval/recon.py:51,90, utils/plots.py:52): the 16-point-wave
    // forward of the training path without its saves -- the same kernel, so the 257 columns are the values a training step computes
    SdfTrainFwdArgs a{};
    a.fwd = packed + p->scale_floats + p->sdf.fwd3h_chunk0 * CHUNK_FLOATS;
    a.n_fwd = sdf_fwd3h_train_stages(256, 256, PE<LF>::DIM, d.n_lin, d.skip_layer > 0, true);
    a.L = d.n_lin; a.skip = d.skip_layer; a.pts = points; a.M = M; a.Mp = M; a.sdf = sdf_out; a.feat = feat_out; a.ldf = ld_feat;
    if (skip_flag != nullptr) return I2SDF_EINVAL;      // (the sampler never asks for features)
    i2sdf_launch_train_fwd3h(a, (unsigned)((M + PTS_PER_WG - 1) / PTS_PER_WG), st);
    return i2sdf_hip_check(hipGetLastError(), "sdf_forward (features, bf16x3, 16-point waves) launch");
  }
  const int ns = sdf_fwd_stages(H, F, PE<LF>::PEC, d.n_lin, d.skip_layer > 0, full);
  const unsigned grid = (unsigned)((M + PTS_PER_WG - 1) / PTS_PER_WG);
  if (full)
    launch_lds(sdf_fwd_kernel<H, F, LF, true>, grid, st, stream, ns, d.n_lin, d.skip_layer, points, skip_flag, M, sdf_out, feat_out, ld_feat);
  else
    launch_lds(sdf_fwd_kernel<H, F, LF, false>, grid, st, stream, ns, d.n_lin, d.skip_layer, points, skip_flag, M, sdf_out, nullptr, 0);
  return i2sdf_hip_check(hipGetLastError(), "sdf_forward launch");
}

}  // namespace

extern "C" int i2sdf_sdf_forward(const i2sdf_plan* p, const float* packed, const float* points, int64_t M, float* sdf_out,
                                 float* feat_out, int64_t ld_feat, void* stream) {
  if (M == 0) return I2SDF_OK;                 // empty batch: nothing to validate, nothing to launch
  if (!p || !packed || !points || M < 0) return I2SDF_EINVAL;
  if (feat_out && (ld_feat < p->F || ld_feat % 4)) return I2SDF_EINVAL;
  hipStream_t st = (hipStream_t)stream;
  if (p->sdf.d.multires != 6) return I2SDF_EINVAL;
  const PointSpec ps{points, nullptr, nullptr, nullptr, 0, 0, 1};
  if (p->H == 256 && p->F == 256) return launch_sdf_fwd<256, 256, 6>(p, packed, ps, nullptr, M, sdf_out, feat_out, ld_feat, st);
  if (p->H == 64 && p->F == 64) return launch_sdf_fwd<64, 64, 6>(p, packed, ps, nullptr, M, sdf_out, feat_out, ld_feat, st);
  return I2SDF_EINVAL;
}

// internal (sampler): sdf at x = cam[r] + z[r*ldz + j]*dirs[r] for j < n_per_ray; whole launch is a no-op when *skip_flag != 0
int i2sdf_sdf_forward_rays_flagged(const i2sdf_plan* p, const float* packed, const float* cam, const float* dirs, const float* z, int64_t ldz,
                                   int32_t n_per_ray, int64_t B, float* sdf_out, const int* skip_flag, void* stream) {
  if (p->sdf.d.multires != 6) return I2SDF_EINVAL;
  const int64_t M = B * n_per_ray;
  const PointSpec ps{nullptr, cam, dirs, z, ldz, M, n_per_ray};
  hipStream_t st = (hipStream_t)stream;
  if (p->H == 256 && p->F == 256) return launch_sdf_fwd<256, 256, 6>(p, packed, ps, skip_flag, M, sdf_out, nullptr, 0, st, true);
  if (p->H == 64 && p->F == 64) return launch_sdf_fwd<64, 64, 6>(p, packed, ps, skip_flag, M, sdf_out, nullptr, 0, st, true);
  return I2SDF_EINVAL;
}
