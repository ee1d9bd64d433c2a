/*
 * i2sdf.h -- C ABI of the MI355X-native volume-rendering core for I2-SDF.
 *
 * The upstream reference (jingsenzhu/i2-sdf) has no FFI / plugin interface: its hot path is entered
 * through one Python call form, `I2SDFNetwork.forward(input, predict_only)` (model/network/__init__.py:80),
 * plus bare `implicit_network(x)` queries (model/eval/recon.py:51,90).  This header is the boundary a
 * maintainer binds UNDER that Python module (ctypes stub in INTEGRATION.md): every entry point replaces a
 * span of stock-torch ops of the reference, cited per function as file:line relative to the upstream root.
 *
 * Conventions
 *   - plain C: raw device pointers, explicit sizes, no torch types;
 *   - all tensors fp32, row-major, contiguous unless a leading dimension is given;
 *   - stream-ordered on the `hipStream_t` passed last (as void*), re-entrant, no allocation inside,
 *     no hidden global state: everything a call needs travels in `i2sdf_net` + caller-owned buffers;
 *   - return 0 on success, a negative I2SDF_E* code otherwise (never exit(), unlike
 *     utils/rend_util.py:220-222); `i2sdf_strerror` maps codes to text.
 */
#ifndef I2SDF_H
#define I2SDF_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define I2SDF_VERSION 100          /* major*10000 + minor*100 + patch */
#define I2SDF_MAX_LAYERS 12
#define I2SDF_MAX_PARTS 4           /* I2SDF_OPT_PARTS: point ranges per batch (one HIP stream each) */

#define I2SDF_OK 0
#define I2SDF_EINVAL (-1)          /* bad argument / unsupported shape */
#define I2SDF_EHIP (-2)            /* HIP runtime error (see i2sdf_last_hip_error) */
#define I2SDF_ESPHERE (-3)         /* ray does not hit the bounding sphere (rend_util.py:220 `exit()`) */
#define I2SDF_EWORKSPACE (-4)      /* caller workspace too small */
#define I2SDF_ECOMM (-5)           /* RCCL error or RCCL unavailable, see i2sdf_last_comm_error() */

/* ------------------------------------------------------------------------------------------------
 * Network description.  Parameters live in ONE flat fp32 device buffer (`params`), laid out exactly in
 * the reference's state_dict order: for each net, for each layer l: bias[out], weight_g[out],
 * weight_v[out*in] (model/network/mlp.py:45-74, 196-203); then density.beta (density.py:6-8).
 * Offsets are in floats.
 * ---------------------------------------------------------------------------------------------- */
typedef struct i2sdf_mlp_desc {
  int32_t n_lin;                          /* number of nn.Linear layers                              */
  int32_t hidden;                         /* hidden width H (multiple of 32, <= 256)                 */
  int32_t d_in;                           /* raw input dims of layer 0 before encoding               */
  int32_t in0;                            /* columns of lin0.weight_v                                */
  int32_t d_out;                          /* rows of the last layer                                  */
  int32_t multires;                       /* positional-encoding frequencies L (embedder.py:138-152) */
  int32_t skip_layer;                     /* l with `l in skip_in` (mlp.py:94), -1 if none           */
  int32_t reserved;
  int32_t out_dim[I2SDF_MAX_LAYERS];      /* rows of weight_v per layer                              */
  int32_t in_dim[I2SDF_MAX_LAYERS];       /* cols of weight_v per layer                              */
  int64_t off_bias[I2SDF_MAX_LAYERS];
  int64_t off_g[I2SDF_MAX_LAYERS];
  int64_t off_v[I2SDF_MAX_LAYERS];
} i2sdf_mlp_desc;

typedef struct i2sdf_net_desc {
  i2sdf_mlp_desc sdf;                     /* ImplicitNetwork: PE(x) -> softplus100 stack -> [sdf | feature]   */
  i2sdf_mlp_desc rgb;                     /* RenderingNetwork 'nerf': [PE(view) | feature] -> ReLU -> sigmoid */
  i2sdf_mlp_desc light;                   /* light-mask head (n_lin == 0 when absent)                         */
  int64_t off_beta;                       /* density.beta                                                     */
  int64_t n_params;                       /* total floats in `params`                                         */
  float beta_min;                         /* density.py:17                                                    */
  float scene_bounding_sphere;            /* model/network/__init__.py:23                                     */
} i2sdf_net_desc;

/* Opaque, caller-owned plan: host copy of the derived layouts + a small device table.
 * i2sdf_plan_create allocates it (the only allocation in the library, one-time), i2sdf_plan_destroy frees. */
typedef struct i2sdf_plan i2sdf_plan;

int i2sdf_version(void);
const char* i2sdf_strerror(int code);
const char* i2sdf_last_hip_error(void);    /* thread-local text of the last HIP failure */

int i2sdf_plan_create(const i2sdf_net_desc* desc, i2sdf_plan** out);
void i2sdf_plan_destroy(i2sdf_plan* plan);
/* Plan options (host side, takes effect on the next launch).
 * Defaults (round 6): a plan of a 256-wide configuration is created with the five *_BF16X3 options, I2SDF_OPT_BLOCKED_SAVES and
 * I2SDF_OPT_TAIL_OVERLAP ON (fp32-equivalent results from the kernels the parity tests and profiles cover), I2SDF_OPT_WGRAD_BF16X2,
 * I2SDF_OPT_SAMPLER_BF16X2 and I2SDF_OPT_PARTS off; a 64-wide plan with everything off.  Value 0 selects the fp32-input MFMA form.
 *   I2SDF_OPT_SDF_FWD_BF16X3: evaluate the sdf-only forward (i2sdf_sdf_forward without features, and the SDF passes inside
 *   i2sdf_sample_rays) in bf16x3 split arithmetic: every fp32 operand is split into three bf16 terms and the six leading
 *   partial products are accumulated in fp32 on the bf16 matrix pipe -- results agree with the fp32 path to fp32 rounding
 *   level (same 1e-4 parity bar) at 3/8 of the matrix-pipe cycles.  (0 = plain fp32 MFMA.) */
#define I2SDF_OPT_SDF_FWD_BF16X3 1
/*   I2SDF_OPT_WGRAD_BF16X3: the 256x256 blocks of i2sdf_weight_grads in the same split arithmetic (both operands are split
 *   on the fly); the narrower blocks run the split form with three planes then, the fp32 MFMA kernel otherwise. */
#define I2SDF_OPT_WGRAD_BF16X3 2
/*   I2SDF_OPT_TRAIN_FWD_BF16X3: the full workgroups of i2sdf_sdf_forward_grad (256-wide nets) in the same arithmetic. */
#define I2SDF_OPT_TRAIN_FWD_BF16X3 4
/*   I2SDF_OPT_SDF_BWD_BF16X3: the full workgroups of i2sdf_sdf_backward (both sweeps, 256-wide nets). */
#define I2SDF_OPT_SDF_BWD_BF16X3 8
/*   I2SDF_OPT_RGB_BF16X3: the full workgroups of i2sdf_rgb_forward / i2sdf_rgb_backward (256-wide nets), and i2sdf_light_forward of the
 *   128-unit light-mask head on 256 features. */
#define I2SDF_OPT_RGB_BF16X3 16
/*   I2SDF_OPT_TAIL_OVERLAP: the split-K tail workgroups of i2sdf_sdf_forward_grad and i2sdf_sdf_backward (the partial last
 *   round of a launch) and the narrow blocks of i2sdf_weight_grads run on a side stream owned by the plan,
 *   concurrently with the full workgroups / the 256x256 blocks; the entry point still returns stream-ordered on the
 *   caller's stream (fork/join with events; capturable in a hipGraph). */
#define I2SDF_OPT_TAIL_OVERLAP 32
/*   (64 was I2SDF_OPT_SRC_RING: saved-tensor reads through a per-wave LDS DMA ring -- built and measured in rounds 2-3, no gain, removed) */
/*   I2SDF_OPT_BLOCKED_SAVES: the 256-wide tensors saved for the backward (hs, abars, gus, gas, rs, gar) are stored in a blocked
 *   layout [Mp/32][16 k-chunks][32 points][16 floats] for the points handled by the bf16x3 full workgroups (point-major for the
 *   rest): a wave instruction of the K-outer kernels then moves one contiguous 2 KB run instead of touching 32 rows 1 KB apart.
 *   Set it before the first i2sdf_sdf_forward_grad of a step and leave it unchanged through i2sdf_weight_grads; the tensors are
 *   opaque to the caller (i2sdf_blocked_points below gives the address formula for inspecting a copy). */
#define I2SDF_OPT_BLOCKED_SAVES 128
/*   I2SDF_OPT_WGRAD_BF16X2 (needs I2SDF_OPT_WGRAD_BF16X3): the 256x256 weight-gradient blocks split every operand into TWO bf16
 *   terms and accumulate the three leading products a0b0 + a0b1 + a1b0 in fp32: per-product error <= 3 * 2^-18 (1.1e-5), i.e.
 *   16+ mantissa bits per operand -- more than the TF32 / bf16 internals the reference itself runs its matmuls with
 *   (main_recon.py:61: torch.set_float32_matmul_precision('medium')), less than fp32.  Only the weight gradients qualify: they are
 *   terminal sums over ~1e5 points (rounding errors of the terms average out and propagate nowhere), measured 1e-5-level
 *   max-norm error of every parameter gradient against fp64 (parity bar 1e-4).  Half the MFMAs and 2/3 of the split work of the
 *   bf16x3 form.  Default 0; every other kernel keeps the fp32-equivalent bf16x3 / fp32 arithmetic. */
#define I2SDF_OPT_WGRAD_BF16X2 256
/*   I2SDF_OPT_PARTS (value n = 2..I2SDF_MAX_PARTS, 0 / 1 = off; 256-wide nets with the bf16x3 options on): the per-point entry points
 *   (i2sdf_sdf_forward_grad, i2sdf_rgb_forward, i2sdf_rgb_backward, i2sdf_sdf_backward, the GEMMs of i2sdf_weight_grads) cut their
 *   point batch into n ranges at multiples of i2sdf_wgrad_chunk_points(); range 0 runs on the caller's stream, the others on streams
 *   owned by the plan.  On its own an entry point forks and joins (it returns stream-ordered on the caller's stream, as without the
 *   option).  Between i2sdf_chain_begin and i2sdf_chain_end the ranges stay un-joined ACROSS entry points: every range runs its own
 *   chain of kernels, the ranges drift apart, and the partly empty last round of one kernel (M/128 workgroups on 256 CUs) is filled
 *   by another range's next kernel instead of idling -- which replaces the split-K tail workgroups (none in this mode: every point
 *   goes through the full-workgroup kernels and every saved tensor is blocked throughout).  Results do not depend on n.  Default 0. */
#define I2SDF_OPT_PARTS 512
/*   I2SDF_OPT_SAMPLER_BF16X2 (256-wide nets with I2SDF_OPT_SDF_FWD_BF16X3): the sdf-only passes INSIDE i2sdf_sample_rays and
 *   i2sdf_render_image -- the evaluations the error-bounded sampler chooses its depths from (ray_sampler.py:83-95, under no_grad) --
 *   split every operand into TWO bf16 terms and accumulate the three leading products (per-product error <= 3 * 2^-18): half the MFMAs
 *   of the bf16x3 form.  No returned value is computed from these passes (sdf, colours, gradients at the chosen depths come from the
 *   fp32-equivalent kernels); what changes is WHERE the samples sit, by a perturbation of the sdf of ~1e-5 relative -- well inside what
 *   the reference itself computes them with on its GPU (torch.set_float32_matmul_precision('medium'), main_recon.py:61).
 *   i2sdf_sdf_forward and i2sdf_sdf_grid (values that are returned) keep three planes.  Default 0. */
#define I2SDF_OPT_SAMPLER_BF16X2 1024
/*   I2SDF_OPT_SAVES24 (256-wide nets; takes effect only together with I2SDF_OPT_PARTS, I2SDF_OPT_BLOCKED_SAVES, the *_BF16X3 options and
 *   I2SDF_OPT_WGRAD_BF16X2 -- otherwise the tensors keep their fp32 form): abars, gus and gas -- the three saved SDF tensors whose consumers are
 *   the weight-gradient GEMMs (which in the two-plane form keep 16 significant bits of every operand) and, for abars / gus, the second-order
 *   injection of sweep 2 -- are stored with 16 significant bits (round to nearest, relative error <= 2^-16 per element) as 3 bytes per value
 *   inside their fp32 allocation (layout: mlp_common.h P24).  hs stays fp32.  8 of the 13 saved-tensor passes per layer move 3/4 of their bytes.
 *   Measured: every parameter gradient stays within 7e-7 (max-norm relative) of the fp32-storage result at 1024 rays; the parity bar is 1e-4
 *   against fp64.  The tensors are opaque to the caller in this form (i2sdf_amd.engine.saved_to_point_major decodes a copy).  Default 0. */
#define I2SDF_OPT_SAVES24 2048
/* number of leading points (a multiple of 32) of a batch whose saved tensors are blocked under the current options: which = 0
 * hs / abars / gus / gas of an i2sdf_sdf_forward_grad batch of M points (has_feat: feat != NULL in that call), which = 1 rs / gar
 * of an i2sdf_rgb_forward batch.  Element (point m < that count, column c) of a blocked (Mp,256) tensor lives at float offset
 * (m/32)*8192 + (c/16)*512 + (m%32)*16 + c%16; points behind the count are ordinary rows m*256 + c.
 * which = 2: the leading points whose abars / gus / gas are packed 24-bit records under the current options (I2SDF_OPT_SAVES24: Mp -- every point --
 * or 0).  A packed layer holds, per 32-point block (6144 floats, dense) and 16-column k-chunk (384 floats): 32 x 2 x 4 dwords of upper halves
 * (lane hi = 0, 1 of point p at dword p*8 + hi*4; value u of that lane = column 4 hi + u for u < 4, 8 + 4 hi + u - 4 for u >= 4; two values per dword)
 * and, from dword 256 on, 32 x 2 x 2 dwords of mid bytes (four per dword); value = upper half << 16 | mid byte << 8.  The top layer of gus stays fp32. */
int64_t i2sdf_blocked_points(const i2sdf_plan* plan, int32_t which, int64_t M, int64_t Mp, int32_t has_feat);
int i2sdf_plan_set_option(i2sdf_plan* plan, int32_t option, int32_t value);
/* A chain (I2SDF_OPT_PARTS): the per-point entry points called between begin and end leave their point ranges un-joined.
 *   i2sdf_chain_begin(plan, M, stream): the ranges are cut from a batch of M points (the LARGEST batch of the chain: entry points
 *     with fewer points clip them); the side streams wait for everything enqueued on `stream` so far.
 *   i2sdf_chain_fence(plan, stream): the side streams additionally wait for what was enqueued on `stream` since (call it after
 *     enqueueing, on `stream`, an input that a later entry point of the chain reads).
 *   i2sdf_chain_end(plan, stream): `stream` waits for every range; afterwards everything the chain wrote is visible on `stream`.
 * Inside a chain only the entry points named at I2SDF_OPT_PARTS may be called, with the same `stream`; i2sdf_weight_grads joins the
 * ranges itself before its final reduction.  Without I2SDF_OPT_PARTS all three calls do nothing. */
int i2sdf_chain_begin(const i2sdf_plan* plan, int64_t M, void* stream);
int i2sdf_chain_fence(const i2sdf_plan* plan, void* stream);
int i2sdf_chain_end(const i2sdf_plan* plan, void* stream);
/* floats of device memory the packed weight streams need (pass to i2sdf_pack_weights) */
int64_t i2sdf_plan_pack_floats(const i2sdf_plan* plan);
/* floats of ONE split-M chunk of effective-weight gradient partial sums: i2sdf_weight_grads takes n_chunks x this many floats of
 * workspace (`partials`) and reduces them, fused with the weight-norm backward, into the flat gradient */
int64_t i2sdf_plan_wgrad_floats(const i2sdf_plan* plan);

/* ------------------------------------------------------------------------------------------------
 * Weight-norm reparametrisation + packing.  Replaces `W = g * v/||v||` executed inside every `lin(x)`
 * (mlp.py:71-72,97,222) for all layers at once and lays W out in the order the MFMA kernels stream it.
 * Must be called after every parameter update, before any kernel below.
 * ---------------------------------------------------------------------------------------------- */
int i2sdf_pack_weights(const i2sdf_plan* plan, const float* params, float* packed, void* stream);

/* ------------------------------------------------------------------------------------------------
 * SDF network forward without gradient -- ImplicitNetwork.forward / get_sdf_vals (mlp.py:84-105,145-151),
 * the call the sampler (ray_sampler.py:88-89), marching cubes (model/eval/recon.py:51,90) and the bubble
 * loss (model/network/__init__.py:200) make.
 *   points   (M,3)
 *   sdf_out  (M)            or NULL
 *   feat_out (M, ld_feat)   or NULL : the feature columns out[:,1:]   (ld_feat >= feature size, multiple of 4)
 * ---------------------------------------------------------------------------------------------- */
int i2sdf_sdf_forward(const i2sdf_plan* plan, const float* packed, const float* points, int64_t M,
                      float* sdf_out, float* feat_out, int64_t ld_feat, void* stream);


/* ------------------------------------------------------------------------------------------------
 * Point batches.  The training kernels take a batch laid out as
 *     [ n_ray_pts points generated from rays | (M - n_ray_pts) explicit points ]
 * ray part:  x[m] = cam[r] + z[r*ldz + j]*dirs[r], r = m / n_per_ray, j = m % n_per_ray (model/network/__init__.py:103)
 * explicit:  x[m] = points[m - n_ray_pts]        (eikonal / neighbour / bubble points, :178-201)
 * Every per-point workspace below has Mp rows (Mp multiple of 128, >= M); rows >= M are never read (kernels may WRITE them: the saved
 * tensors' stores of padding points are unconditional instructions, which lets the stage waits count them -- csrc/x3.h).
 *
 * SDF network forward WITH d sdf/dx and saved activations -- ImplicitNetwork.get_outputs / .gradient
 * (mlp.py:107-143) as called by the main render pass (model/network/__init__.py:113) and the eikonal pass
 * (:188).  The reference obtains d sdf/dx from torch.autograd.grad(create_graph=True); here the reverse
 * chain is explicit and fused into the same kernel.
 *   sdf (M) ; feat (Mp,F) or NULL ; grad (M,3) or NULL
 *   hs      (L-1, Mp, H)  h_l = softplus100(a_{l-1}), l = 1..L-1   (required when grad != NULL)
 *   abars   (L-1, Mp, H)  d sdf / d a_l, l = 0..L-2                (NULL if no backward will follow)
 *   pe_save (Mp, 40)      PE(x), an operand of the weight-gradient GEMMs (NULL if no backward will follow)
 * ---------------------------------------------------------------------------------------------- */
int i2sdf_sdf_forward_grad(const i2sdf_plan* plan, const float* packed, const float* points, const float* cam,
                           const float* dirs, const float* z, int64_t ldz, int32_t n_per_ray, int64_t n_ray_pts, int64_t M,
                           int64_t Mp, float* sdf, float* feat, float* grad, float* hs, float* abars, float* pe_save,
                           void* stream);

/* ------------------------------------------------------------------------------------------------
 * Radiance network forward, 'nerf' mode -- RenderingNetwork.forward (mlp.py:208-229):
 * rgb = sigmoid(MLP([PE4(view_dir) | feature])).  dirs (B,3) unit view directions, point m uses dirs[m / n_per_ray].
 *   rgb (M,3) ; rs (L-1, Mp, H) post-ReLU activations and pev_save (Mp,32) PE(view) (NULL if no backward follows)
 * ---------------------------------------------------------------------------------------------- */
int i2sdf_rgb_forward(const i2sdf_plan* plan, const float* packed, const float* dirs, int32_t n_per_ray, const float* feat,
                      int64_t M, int64_t Mp, float* rgb, float* rs, float* pev_save, void* stream);

/* ------------------------------------------------------------------------------------------------
 * Backward of the radiance network (autograd through mlp.py:208-229).
 *   rgb (M,3) forward output, rgb_bar (M,3) upstream, rs from the forward
 *   -> gar (L-1, Mp, H) G(a_l) l=0..L-2 ; ga_last (Mp,4) G(a_{L-1}) ; fbar (Mp,F) d loss / d feature
 * ---------------------------------------------------------------------------------------------- */
int i2sdf_rgb_backward(const i2sdf_plan* plan, const float* packed, const float* rgb, const float* rgb_bar, const float* rs,
                       int64_t M, int64_t Mp, float* gar, float* ga_last, float* fbar, void* stream);

/* ------------------------------------------------------------------------------------------------
 * Backward of the SDF network including the double backward through d sdf/dx (what loss.backward() does through
 * mlp.py:84-143 with create_graph=True): sweep 1 = adjoint of the d sdf/dx chain, sweep 2 =
 * ordinary backward.  Upstream: sbar (M) d/d sdf, fbar (Mp,F) d/d feature (rows >= m_fbar are zero), nbar (M,3) d/d grad;
 * any of them may be NULL (= 0).  Emits the operands of the weight-gradient GEMMs:
 *   gus (L, Mp, H)  G(hbar_l) for l = 1..L-1 (slot 0 unused) ; gpbar (Mp,40) G(pbar) ;
 *   gas (L-1, Mp, H) G(a_l), l = 0..L-2 ; ga_last4 (Mp,4) {sbar,0,0,0} ; ones4 (Mp,4) {1,0,0,0}
 * ---------------------------------------------------------------------------------------------- */
int i2sdf_sdf_backward(const i2sdf_plan* plan, const float* packed, const float* points, const float* cam, const float* dirs,
                       const float* z, int64_t ldz, int32_t n_per_ray, int64_t n_ray_pts, int64_t M, int64_t Mp,
                       const float* hs, const float* abars, const float* sbar, const float* fbar, int64_t m_fbar,
                       const float* nbar, float* gus, float* gpbar, float* gas, float* ga_last4, float* ones4, void* stream);

/* ------------------------------------------------------------------------------------------------
 * Ray set-up -- utils/rend_util.py:92-147 (get_camera_params + lift, pose-matrix form) and
 * model/network/__init__.py:88-93 (per-pixel cam_loc, ||d||, F.normalize).
 *   uv (batch, pixels, 2) pixel coords ; pose (batch,4,4) cam->world ; intrinsics (batch,4,4)
 *   -> cam_loc (N,3), dirs (N,3) unit, dnorm (N) with N = batch*pixels
 * ---------------------------------------------------------------------------------------------- */
int i2sdf_ray_setup(const float* uv, const float* pose, const float* intrinsics, int64_t batch, int32_t pixels,
                    float* cam_loc, float* dirs, float* dnorm, void* stream);
/* Same with the pose given as (batch,7) = [qr qi qj qk tx ty tz] when pose_is_quat != 0 (rend_util.py:93-98, quat_to_rot :150-167). */
int i2sdf_ray_setup_ex(const float* uv, const float* pose, int32_t pose_is_quat, const float* intrinsics, int64_t batch,
                       int32_t pixels, float* cam_loc, float* dirs, float* dnorm, void* stream);

/* ------------------------------------------------------------------------------------------------
 * Ray batcher -- replaces ReconDataset.__getitem__ + collate_fn (dataset/train_dataset.py:169-209),
 * which stack one 4x4 K and one 4x4 pose PER RAY on the host, and the get_camera_params call that consumes them.
 * The camera tables and (optionally) the ground-truth images stay resident in HBM; a batch is a list of global pixel
 * indices tidx = image * (H*W) + pixel (what the DataLoader's sampler yields).  One pass writes the rays and gathers
 * the ground truth.  uv follows dataset/train_dataset.py:67-70: uv = (column, row) as floats.
 * Any table / output pointer except intrinsics, pose, cam_loc, dirs, dnorm may be NULL (skipped).
 * Dtypes follow the dataset: mask / light_mask are float images (:82,:96), depth_mask / normal_mask are bool bytes (:123,:163).
 * ---------------------------------------------------------------------------------------------- */
typedef struct i2sdf_ray_tables {
  const float* intrinsics;         /* (n_images,4,4) */
  const float* pose;               /* (n_images,4,4) cam->world, or (n_images,7) when pose_is_quat */
  int32_t pose_is_quat;
  int32_t n_images, height, width;
  const float* rgb;                /* (n_images, H*W, 3) */
  const float* depth;              /* (n_images, H*W)    */
  const float* normal;             /* (n_images, H*W, 3) */
  const float* mask;               /* (n_images, H*W, 1) */
  const float* light_mask;         /* (n_images, H*W, 1) */
  const uint8_t* depth_mask;       /* (n_images, H*W)    */
  const uint8_t* normal_mask;      /* (n_images, H*W)    */
} i2sdf_ray_tables;

typedef struct i2sdf_ray_batch_out {
  int64_t* image_idx;              /* (n_rays)   tidx / (H*W) */
  float* uv;                       /* (n_rays,2) */
  float* cam_loc;                  /* (n_rays,3) */
  float* dirs;                     /* (n_rays,3) unit */
  float* dnorm;                    /* (n_rays)   */
  float* rgb;                      /* (n_rays,3) */
  float* depth;                    /* (n_rays)   */
  float* normal;                   /* (n_rays,3) */
  float* mask;                     /* (n_rays,1) */
  float* light_mask;               /* (n_rays,1) */
  uint8_t* depth_mask;             /* (n_rays)   */
  uint8_t* normal_mask;            /* (n_rays)   */
  int32_t* n_bad;                  /* NULL or device counter (caller-zeroed): += number of tidx outside [0, n_images*H*W); such rays
                                      are produced from the clamped index, nothing is read out of bounds */
} i2sdf_ray_batch_out;

int i2sdf_ray_batch(const i2sdf_ray_tables* tables, const int64_t* tidx, int64_t n_rays, const i2sdf_ray_batch_out* out, void* stream);

/* ------------------------------------------------------------------------------------------------
 * Ray / bounding-sphere intersection -- utils/rend_util.py:211-227 (reached only with a background network).
 *   -> t_near_far (n_rays,2) = clamp(+-sqrt((d.o)^2 - (|o|^2 - r^2)) - d.o, min 0).  Where the reference prints and exit()s
 *   (non-positive discriminant) this entry writes (0,0) for the ray and increments *n_miss (device int32, caller-zeroed).
 * ---------------------------------------------------------------------------------------------- */
int i2sdf_sphere_intersections(const float* cam_loc, const float* dirs, int64_t n_rays, float radius, float* t_near_far,
                               int32_t* n_miss, void* stream);

/* ------------------------------------------------------------------------------------------------
 * Laplace density + log-space alpha compositing -- LaplaceDensity (density.py:21-30),
 * I2SDFNetwork.volume_rendering (model/network/__init__.py:223-240) and the weighted sums (:120-125,
 * :169, :204-219).  beta = |*beta_param| + beta_min is read on the device (no host sync).
 *   z (B, ldz): n sample depths followed by z_max in column n ; sdf (B*n) ; rgb (B*n,3) ;
 *   grad (B*n,3) raw d sdf/dx (needed iff o_normal) ; lmask (B*n) (needed iff o_lmask) ; dnorm (B)
 *   -> o_rgb (B,3), o_depth (B), o_wsum (B), o_normal (B,3)|NULL, o_lmask (B)|NULL,
 *      w_save (B,n)|NULL weights, nsum_save (B,3)|NULL un-normalised normal sum (for the backward)
 * ---------------------------------------------------------------------------------------------- */
int i2sdf_composite_forward(const float* beta_param, float beta_min, const float* z, int64_t ldz, const float* sdf,
                            const float* rgb, const float* grad, const float* lmask, const float* dnorm, int64_t B, int32_t n,
                            float* o_rgb, float* o_depth, float* o_wsum, float* o_normal, float* o_lmask, float* w_save,
                            float* nsum_save, void* stream);

/* The same and i2sdf_eikonal_outputs_forward (below) in ONE launch -- the two per-ray launches of a training forward between the radiance
 * net and the loss: grad_all (3B,3) = d sdf / d x of the extra points [uniform | near | neighbour] -> grad_theta (2B,3), diff_norm (B). */
int i2sdf_composite_forward_eik(const float* beta_param, float beta_min, const float* z, int64_t ldz, const float* sdf,
                                const float* rgb, const float* grad, const float* lmask, const float* dnorm, int64_t B, int32_t n,
                                float* o_rgb, float* o_depth, float* o_wsum, float* o_normal, float* o_lmask, float* w_save,
                                float* nsum_save, const float* grad_all, float* grad_theta, float* diff_norm, void* stream);

/* Backward of the above (what loss.backward() does through :118-125,:169,:204-209).
 * Upstream: g_rgb (B,3), g_depth (B)|NULL, g_wsum (B)|NULL, g_normal (B,3)|NULL (w.r.t. normal_values),
 * g_lmask (B)|NULL.  The normal / light composites use w.detach() as the reference does in training.
 *   -> sdf_bar (B*n), rgb_bar (B*n,3), grad_bar (B*n,3)|NULL, lmask_bar (B*n)|NULL,
 *      beta_partial (B) scratch; if beta_grad_accum != NULL: *beta_grad_accum += d loss / d density.beta */
int i2sdf_composite_backward(const float* beta_param, float beta_min, const float* z, int64_t ldz, const float* sdf,
                             const float* rgb, const float* grad, const float* dnorm, const float* nsum_save, int64_t B, int32_t n,
                             const float* g_rgb, const float* g_depth, const float* g_wsum, const float* g_normal,
                             const float* g_lmask, float* sdf_bar, float* rgb_bar, float* grad_bar, float* lmask_bar,
                             float* beta_partial, float* beta_grad_accum, void* stream);


/* ------------------------------------------------------------------------------------------------
 * Weight gradients: the dW = G^T U GEMMs of every nn.Linear (autograd's mm/addmm backward nodes for mlp.py:97,222),
 * reduced over all points, then the weight-norm backward (d/dg, d/dv of W = g v/||v||) and bias gradients, written
 * into `grad_flat` (same layout as `params`; entries of nets that took no part are left untouched).
 * All pointers are the per-point workspaces the kernels above produced ([Mp][ld], see their comments).
 * Every one of them has Mp rows, Mp a multiple of 128 (as the producing kernels require): the bf16x3 kernel reads whole 16-point
 * stages, i.e. up to 15 rows past M_sdf / M_main, and masks them -- the padding rows may hold anything but must exist.
 * ---------------------------------------------------------------------------------------------- */
typedef struct i2sdf_train_buffers {
  int64_t M_sdf;    /* points that went through the SDF network (render + eikonal + bubble points) */
  int64_t M_main;   /* leading points that also went through the radiance (and light) network        */
  int64_t Mp;
  const float *pe, *hs, *abars, *gus, *gpbar, *gas, *ga_last4, *ones4, *fbar;   /* SDF network   */
  const float *pev, *feat, *rs, *gar, *ga_last_rgb;                             /* radiance net  */
  const float *hl, *gal0, *gal_last;                                            /* light head (NULL if absent) */
} i2sdf_train_buffers;

int64_t i2sdf_wgrad_chunk_points(void);   /* points per split-M chunk: partials needs ceil(M_sdf/chunk) * wgrad_floats floats */
int i2sdf_weight_grads(const i2sdf_plan* plan, const i2sdf_train_buffers* bufs, const float* params, float* partials,
                       int64_t n_chunks_cap, float* grad_flat, void* stream);


/* ------------------------------------------------------------------------------------------------
 * Data parallelism (the reference itself is single-GPU: main_recon.py:111-112).  One process per GPU; rays
 * are sharded, parameters replicated, and the only exchange of a training step is the parameter-gradient mean over the flat
 * buffer -- RCCL over xGMI.  RCCL is bound at run time (librccl.so.1); without it these calls return I2SDF_ECOMM.
 *   i2sdf_comm_unique_id : rank 0 makes the 128-byte id, the caller distributes it (any side channel)
 *   i2sdf_comm_init_rank : collective over all ranks; the calling thread's current HIP device is the rank's GPU
 *   i2sdf_allreduce_grads: in place, stream ordered, every rank ends with the MEAN over ranks (DDP convention)
 *   i2sdf_allreduce_max_i32 / i2sdf_broadcast: the small exchanges of 1-GPU-equivalent mode (below)
 * 1-GPU-equivalent mode: besides the gradient mean, a sharded step needs two SMALL exchanges to equal the single-GPU step on
 * the concatenated batch.  They go through an `i2sdf_exchange` hook (an in-place all-reduce of a few device words, enqueued on
 * the stream, never synchronising the host): i2sdf_comm_as_exchange() gives the RCCL implementation; a caller may supply its
 * own (i2sdf_amd/dist.py routes it through torch.distributed for gloo tests).
 *   i2sdf_plan_set_exchange(plan, ex, I2SDF_DP_GLOBAL_SAMPLER): the sampler's convergence test becomes the batch-global OR over
 *     ALL ranks' rays (ray_sampler.py:151: while any ray of the batch is unconverged every ray is up-sampled) -- one 4-byte MAX
 *     all-reduce per iteration;
 *   i2sdf_loss_cfg.exchange: every loss denominator (B, n_pc, masked-mean counts, model/network/__init__.py:320-336) becomes its
 *     mean over the ranks, so that the mean over ranks of the per-rank gradients is the gradient on the concatenated batch.
 * ---------------------------------------------------------------------------------------------- */
typedef struct i2sdf_comm i2sdf_comm;
#define I2SDF_COMM_UNIQUE_ID_BYTES 128
#define I2SDF_DP_GLOBAL_SAMPLER 1
#define I2SDF_XCHG_F32 0
#define I2SDF_XCHG_I32 1
#define I2SDF_XCHG_SUM 0
#define I2SDF_XCHG_AVG 1
#define I2SDF_XCHG_MAX 2
typedef struct i2sdf_exchange {
  /* in-place all-reduce of n elements at device pointer buf, ordered on `stream`; returns 0 or a negative error code */
  int (*allreduce)(void* ctx, void* buf, int64_t n, int32_t dtype, int32_t op, void* stream);
  void* ctx;
} i2sdf_exchange;
int32_t i2sdf_comm_available(void);       /* 1 if RCCL could be bound in this process (creates nothing): agree on it across ranks before i2sdf_comm_init_rank */
int i2sdf_comm_unique_id(void* out, int64_t out_bytes);
int i2sdf_comm_init_rank(const void* unique_id, int32_t nranks, int32_t rank, i2sdf_comm** out);
void i2sdf_comm_destroy(i2sdf_comm* comm);
int32_t i2sdf_comm_size(const i2sdf_comm* comm);
int32_t i2sdf_comm_rank(const i2sdf_comm* comm);
const char* i2sdf_last_comm_error(void);
int i2sdf_allreduce_grads(float* flat, int64_t n, const i2sdf_comm* comm, void* stream);
int i2sdf_allreduce_max_i32(int32_t* flags, int64_t n, const i2sdf_comm* comm, void* stream);
int i2sdf_broadcast(void* buf, int64_t bytes, int32_t root, const i2sdf_comm* comm, void* stream);
int i2sdf_comm_as_exchange(const i2sdf_comm* comm, i2sdf_exchange* out);
int i2sdf_plan_set_exchange(i2sdf_plan* plan, const i2sdf_exchange* ex, int32_t flags);   /* ex = NULL detaches; *ex is copied */

/* ------------------------------------------------------------------------------------------------
 * Optimizer step over the flat parameter buffer -- torch.optim.Adam(model.get_param_groups(lr), eps=1e-15) of
 * model/trainer/recon.py:201-203 (same update rule and operation order, bias corrections in double on the host) as ONE
 * launch.  step = 1 for the first update.  grad_scale multiplies the gradient first (1/world for a summed all-reduce).
 *   params, grads, exp_avg, exp_avg_sq: (n) fp32, updated in place (grads is read-only)
 * ---------------------------------------------------------------------------------------------- */
int i2sdf_adam_step(float* params, const float* grads, float* exp_avg, float* exp_avg_sq, int64_t n, double lr, double beta1,
                    double beta2, double eps, double weight_decay, int64_t step, double grad_scale, void* stream);

/* ------------------------------------------------------------------------------------------------
 * Error-bounded ray sampler -- ErrorBoundSampler.get_z_vals incl. UniformSampler and get_error_bound
 * (model/network/ray_sampler.py:22-43,67-251), bg disabled.  Runs the whole Algorithm-1 loop on the device:
 * all max_total_iters iterations are enqueued, a device flag turns the remaining ones into no-ops once the
 * batch-global test `beta.max() > beta0` (ray_sampler.py:151) fails, so there is no host synchronisation.
 * Random draws are inputs (training): strat_u (B,N_eval) [:39], u_final (B,N_samples) [:190],
 * extra_idx (max_total_iters, N_extra): row it-1 = randperm(N_eval*it)[:N_extra], the draw for a loop that ran `it`
 * iterations [:223] (the row length is only known on the device), eik_idx (B) [:233].  Deterministic tables (eval and the
 * error-proportional up-sampling): t_lin = linspace(0,1,N_eval) [:30], u_more = linspace(0,1,N_eval) [:188],
 * u_final = linspace(0,1,N_samples) with ldu_final = 0 [:188], extra_tab (max_total_iters, N_extra) =
 * linspace(0, N_eval*(it+1)-1, N_extra).long() per possible row length [:225].
 *   force_iters > 0 replaces the data-dependent test by "exactly force_iters iterations" (fixed-work benchmarks).
 *   -> z_out (B, ldz): N_samples + N_extra + 2 sorted depths (last = far) ; z_eik (B)|NULL ; iters_out device int|NULL
 * ---------------------------------------------------------------------------------------------- */
typedef struct i2sdf_sampler_cfg {
  float near, eps, add_tiny;
  int32_t N_samples, N_samples_eval, N_samples_extra, beta_iters, max_total_iters;
} i2sdf_sampler_cfg;

/* ErrorBoundSampler.get_error_bound (model/network/ray_sampler.py:243-251) and the Theorem-1 distance bound d* of
 * get_z_vals (:99-114) on caller-provided rows -- the device functions the Algorithm-1 loop uses, as their own entry point.
 *   z, sdf (B,n) sorted depths / sdf values, 2 <= n <= 640 ; beta: one value (ldbeta = 0) or one per ray (ldbeta = 1)
 *   d_star_in (B,n-1)|NULL: use these bounds instead of computing them ; d_star_out (B,n-1)|NULL ; bound (B)|NULL
 *   -> bound[r] = max_i (min(exp(cumsum err)_i, 1e6) - 1) * exp(-integral_i)                                       */
int i2sdf_error_bound(const float* z, const float* sdf, int64_t B, int32_t n, const float* beta, int64_t ldbeta,
                      const float* d_star_in, float* d_star_out, float* bound, void* stream);

int64_t i2sdf_sampler_workspace_floats(int64_t B);
int i2sdf_sample_rays(const i2sdf_plan* plan, const float* packed, const float* params, const i2sdf_sampler_cfg* cfg,
                      const float* cam, const float* dirs, int64_t B, int32_t training, const float* t_lin, const float* u_more,
                      const float* u_final, int64_t ldu_final, const int32_t* extra_tab, const float* strat_u,
                      const int32_t* extra_idx, const int32_t* eik_idx, int32_t force_iters, float* workspace, float* z_out,
                      int64_t ldz, float* z_eik, int32_t* iters_out, void* stream);


/* ------------------------------------------------------------------------------------------------
 * Light-mask head -- model/network/__init__.py:29-32,162-170:
 * lm = sigmoid(W1 softplus100(W0 relu(feature).detach() + b0) + b1).  Backward stops at the head's parameters
 * (detach_light_feature = True, the reference default).
 *   forward : feat (Mp,F) -> lm (M), hl (Mp,HL) softplus activations (NULL if no backward follows)
 *   backward: lm_bar (M) -> gal0 (Mp,HL) G(a_0), gal_last (Mp,4) {G(a_1),0,0,0}   (weight-gradient operands)
 * ---------------------------------------------------------------------------------------------- */
int i2sdf_light_forward(const i2sdf_plan* plan, const float* packed, const float* feat, int64_t M, int64_t Mp, float* lm, float* hl,
                        void* stream);
int i2sdf_light_backward(const i2sdf_plan* plan, const float* packed, const float* lm, const float* lm_bar, const float* hl, int64_t M,
                         int64_t Mp, float* gal0, float* gal_last, void* stream);


/* ------------------------------------------------------------------------------------------------
 * I2SDFLoss -- model/network/__init__.py:289-406 -- value AND gradient w.r.t. every render output in one call
 * (SURVEY.md "next" row N1).  Per-ray tensors as returned by the module: rgb (B,3), depth (B), wsum (B),
 * normal (B,3)|NULL, grad_theta (2B,3)|NULL, diff_norm (B)|NULL, surface (n_pc)|NULL, lmask (B)|NULL; ground truth
 * gt_rgb (B,3), gt_depth (B)+depth_mask (B bytes)|NULL, gt_normal (B,3)+normal_mask|NULL, gt_mask (B)|NULL,
 * gt_lmask (B)|NULL.  `smooth_on` = the reference's `smooth_iter is None or step > smooth_iter` (:347).
 *   -> losses[10] = {loss, rgb, eikonal, smooth, mask, depth, normal, angular, bubble, light_mask} (device), loss_value (1)|NULL = the
 *      total once more as a tensor of its own (a scalar for autograd without a select on the vector),
 *      g_* = d loss / d (same-named input); scratch: i2sdf_loss_scratch_floats() floats of workspace, contents irrelevant on entry
 *      (nothing persists in it between calls: no counter, nothing to zero; calls in flight on DIFFERENT streams need different
 *      scratch buffers, calls on one stream can share one).
 *   Two launches: per-block partial sums; then the reported values + every gradient, where every workgroup first adds the block
 *   partials up itself, in block order (deterministic, no atomics).  With an exchange hook a one-workgroup reduction and the hook's
 *   collective sit between the two.
 * ---------------------------------------------------------------------------------------------- */
typedef struct i2sdf_loss_cfg {
  float eikonal_w, smooth_w, mask_w, depth_w, normal_w, angular_w, bubble_w, light_w;
  int32_t smooth_on;
  int32_t reserved;
  const i2sdf_exchange* exchange;  /* NULL, or: every denominator (B, n_pc, mask counts) becomes its mean over the ranks, so that the
                                      mean over ranks of the per-rank gradients is the gradient on the concatenated batch */
} i2sdf_loss_cfg;

int64_t i2sdf_loss_scratch_floats(void);
int i2sdf_loss_forward_backward(const i2sdf_loss_cfg* cfg, int64_t B, int64_t n_pc, const float* rgb, const float* depth,
                                const float* wsum, const float* normal, const float* grad_theta, const float* diff_norm,
                                const float* surface, const float* lmask, const float* gt_rgb, const float* gt_depth,
                                const uint8_t* depth_mask, const float* gt_normal, const uint8_t* normal_mask, const float* gt_mask,
                                const float* gt_lmask, float* scratch, float* losses, float* loss_value, float* g_rgb, float* g_depth, float* g_wsum,
                                float* g_normal, float* g_grad_theta, float* g_diff_norm, float* g_surface, float* g_lmask,
                                void* stream);

/* ------------------------------------------------------------------------------------------------
 * Eikonal / smoothness outputs -- model/network/__init__.py:188-193.  grad_all (3B,3) = d sdf/dx at the extra points
 * [B uniform | B near-surface | B neighbours] (the tail of i2sdf_sdf_forward_grad's `grad`):
 *   forward : grad_theta (2B,3)|NULL = rows [0,2B) (:189);  diff_norm (B) = ||normalize(g[B+i]) - normalize(g[2B+i])||,
 *             normalize = F.normalize(dim=1, eps=1e-6) (:190-192)
 *   backward: grad_theta_bar (2B,3)|NULL, diff_norm_bar (B)|NULL -> grad_all_bar (3B,3), every row written
 *             (torch's conventions: d||x|| = 0 at x = 0, no gradient through a norm below eps).
 * ---------------------------------------------------------------------------------------------- */
int i2sdf_eikonal_outputs_forward(const float* grad_all, int64_t B, float* grad_theta, float* diff_norm, void* stream);
int i2sdf_eikonal_outputs_backward(const float* grad_all, const float* grad_theta_bar, const float* diff_norm_bar, int64_t B,
                                   float* grad_all_bar, void* stream);

/* ------------------------------------------------------------------------------------------------
 * The extra points of a training step -- model/network/__init__.py:175-186 -- in one launch instead of a multiply, two adds and a
 * concatenation:  out (3B,3) = [ eik_pts (B,3) | cam + z_eik * dirs (B,3) | the same + nbr_off (B,3) ],  z_eik (B) the per-ray depth
 * i2sdf_sample_rays returned, the sum rounded as the reference's two fp32 operations (product, then sum; no fused multiply-add).
 * ---------------------------------------------------------------------------------------------- */
int i2sdf_extra_points(const float* cam, const float* dirs, const float* z_eik, const float* eik_pts, const float* nbr_off, int64_t B,
                       float* out, void* stream);

/* ------------------------------------------------------------------------------------------------
 * Seeds of the backward pass (what autograd's accumulation does with a handful of fills and copies), one launch:
 *   beta_grad[0, n_beta) = 0                                   (i2sdf_composite_backward accumulates d loss / d beta into it)
 *   sdf_bar (M_sdf)    : rows [M_main, M_sdf) = 0, except rows [M_main + n_eik, M_main + n_eik + n_pc) = g_surf (n_pc)|NULL
 *   grad_bar (M_sdf,3) : rows [M_main, M_sdf) = 0, except rows [M_main, M_main + n_eik) = g_eik (n_eik,3)|NULL;
 *                        rows [0, M_main) = 0 too iff zero_main_grad (no normal output: the compositing backward does not write them)
 * The rays' rows [0, M_main) of sdf_bar (and of grad_bar otherwise) are left to i2sdf_composite_backward, which writes all of them.
 * ---------------------------------------------------------------------------------------------- */
int i2sdf_backward_seeds(float* beta_grad, int64_t n_beta, float* sdf_bar, float* grad_bar, int64_t M_main, int64_t M_sdf,
                         const float* g_eik, int64_t n_eik, const float* g_surf, int64_t n_pc, int32_t zero_main_grad, void* stream);

/* ------------------------------------------------------------------------------------------------
 * Loss + render backward fused (round 6): I2SDFLoss -- model/network/__init__.py:289-406 -- evaluated on the outputs of a TRAINING render,
 * together with the backward of everything between those outputs and the per-sample gradients the MLP backward starts from: the
 * compositing backward (:223-240,:120-125,:169,:204-219), the backward of the eikonal / smoothness outputs (:188-193) and the seeds of the
 * extra points.  Replaces i2sdf_loss_forward_backward + i2sdf_eikonal_outputs_backward + i2sdf_backward_seeds + i2sdf_composite_backward
 * (seven launches and, in between, two of autograd's own) by two launches, for an upstream gradient of 1; i2sdf_scale_seeds multiplies the
 * results by the gradient autograd delivers.  No data-parallel exchange hook here (cfg->exchange must be NULL: use the separate entry points).
 *   batch layout as in i2sdf_backward_seeds: M_main = B n ray samples, then n_eik (= 3 B or 0) extra points [uniform | near | neighbour],
 *     then n_pc bubble points (iff surface), M_sdf rows in all
 *   compositing inputs as in i2sdf_composite_backward: beta_param, z (B, ldz), sdf (M_sdf), rgb_pts (M_main,3), grad_pts (M_sdf,3)|NULL
 *     (d sdf / d x of every point: needed iff normal_term or n_eik), dnorm (B), nsum_save (B,3)|NULL (iff normal_term)
 *   render outputs / ground truth as in i2sdf_loss_forward_backward (grad_theta (2B,3) and diff_norm (B) iff n_eik; surface (n_pc)|NULL)
 *   -> losses (10), loss_value (1)|NULL, the output seeds g_* (as i2sdf_loss_forward_backward writes them),
 *      sdf_bar (M_sdf), rgb_bar (M_main,3), grad_bar (M_sdf,3) -- every row written; normal_term = 0: the ray samples' rows are zeros --,
 *      lmask_bar (M_main)|NULL, beta_grad (1) = d loss / d beta_param;  scratch: i2sdf_render_loss_scratch_floats(B) floats
 * ---------------------------------------------------------------------------------------------- */
int64_t i2sdf_render_loss_scratch_floats(int64_t B);
int i2sdf_render_loss_backward(const i2sdf_loss_cfg* cfg, int64_t B, int32_t n, int64_t n_pc, int64_t M_main, int64_t M_sdf, int64_t n_eik,
                               const float* beta_param, float beta_min, const float* z, int64_t ldz, const float* sdf, const float* rgb_pts,
                               const float* grad_pts, const float* dnorm, const float* nsum_save,
                               const float* rgb, const float* depth, const float* wsum, const float* normal, const float* grad_theta,
                               const float* diff_norm, const float* surface, const float* lmask,
                               const float* gt_rgb, const float* gt_depth, const uint8_t* depth_mask, const float* gt_normal,
                               const uint8_t* normal_mask, const float* gt_mask, const float* gt_lmask,
                               float* scratch, float* losses, float* loss_value,
                               float* g_rgb, float* g_depth, float* g_wsum, float* g_normal, float* g_grad_theta, float* g_diff_norm,
                               float* g_surface, float* g_lmask,
                               float* sdf_bar, float* rgb_bar, float* grad_bar, int32_t normal_term, float* lmask_bar, float* beta_grad,
                               void* stream);
/* x *= g[0] for the four per-sample gradient tensors (n_* = their float counts; a pointer may be NULL with a count of 0) and
 * beta_out[0] = beta_in[0] * g[0] (beta_out may be NULL); g = a device scalar (the gradient of the loss value autograd delivers). */
int i2sdf_scale_seeds(const float* g, float* sdf_bar, int64_t n_sdf, float* grad_bar, int64_t n_grad, float* rgb_bar, int64_t n_rgb,
                      float* lmask_bar, int64_t n_lmask, const float* beta_in, float* beta_out, void* stream);

/* ------------------------------------------------------------------------------------------------
 * Full-image inference in one call -- the chunk loop of utils.split_input / model(chunk) /
 * utils.merge_output (utils/__init__.py:35-84) as used by model/eval/recon.py:161-182 and the plotting callbacks: eval mode,
 * `chunk` = split_n_pixels rays at a time, every chunk rendered exactly as the reference renders it (the sampler's convergence
 * test is per chunk), outputs written straight into the (P, C) image-order tensors.  Enqueues all chunks on `stream` without
 * allocating or synchronising; one chunk-sized workspace (i2sdf_render_image_workspace_floats) is reused by every chunk.
 *   uv (P,2) pixel coordinates of ONE view; pose (4,4) or (7); intrinsics (4,4); tables as for i2sdf_sample_rays (eval)
 *   -> o_rgb (P,3), o_depth (P), o_wsum (P), o_normal (P,3)|NULL = normal_map, o_lmask (P) (required iff the plan has a light
 *      head), o_z (P, N_samples+N_extra+2)|NULL the depths used, o_iters (ceil(P/chunk)) device ints|NULL sampler iterations
 * ---------------------------------------------------------------------------------------------- */
int64_t i2sdf_render_image_workspace_floats(const i2sdf_plan* plan, const i2sdf_sampler_cfg* cfg, int64_t chunk);
int i2sdf_render_image(const i2sdf_plan* plan, const float* packed, const float* params, const i2sdf_sampler_cfg* cfg,
                       const float* uv, const float* pose, int32_t pose_is_quat, const float* intrinsics, int64_t P, int64_t chunk,
                       const float* t_lin, const float* u_more, const float* u_final, const int32_t* extra_tab, float* workspace,
                       float* o_rgb, float* o_depth, float* o_wsum, float* o_normal, float* o_lmask, float* o_z, int32_t* o_iters,
                       void* stream);

/* ------------------------------------------------------------------------------------------------
 * SDF volume for marching cubes -- model/eval/recon.py:46-51 (coarse uniform grid) and :75-103 (the
 * PCA-aligned fine grid evaluated through GridDataset / a 32-worker DataLoader), utils/plots.py:440-489 (get_grid_uniform,
 * get_grid: np.meshgrid(x, y, z) flattened to an (n,3) host tensor).  The axis vectors are the whole input: grid points are
 * generated on the device one chunk ahead of the SDF kernel and never exist on the host.
 *   x (nx), y (ny), z (nz)   device fp32 axis coordinates (the reference casts its float64 linspace / arange to float)
 *   order   I2SDF_GRID_ORDER_MESHGRID: output index ((iy*nx)+ix)*nz+iz = np.meshgrid(x,y,z).ravel() order, the order of the
 *           reference's flat `z` before its reshape(ny,nx,nz).transpose([1,0,2]);  I2SDF_GRID_ORDER_VOLUME: ((ix*ny)+iy)*nz+iz,
 *           i.e. that transposed (nx,ny,nz) volume itself, ready for measure.marching_cubes
 *   rot (9, row-major, HOST) | NULL, trans (3, HOST) | NULL : evaluated point = rot * (x,y,z) + trans; the aligned grid of
 *           model/eval/recon.py:82-85 passes rot = vecs^T, trans = s_mean
 *   [first, first+count) the output indices to evaluate (a rank's slab; count = nx*ny*nz for everything)
 *   sdf_out (count); workspace i2sdf_sdf_grid_workspace_floats(chunk_points) floats; chunk_points > 0
 * Enqueues ceil(count/chunk_points) generator + SDF launches on `stream`; no allocation, no synchronisation.
 * ---------------------------------------------------------------------------------------------- */
#define I2SDF_GRID_ORDER_MESHGRID 0
#define I2SDF_GRID_ORDER_VOLUME 1
int64_t i2sdf_sdf_grid_workspace_floats(int64_t chunk_points);
int i2sdf_sdf_grid(const i2sdf_plan* plan, const float* packed, const float* x, const float* y, const float* z, int32_t nx,
                   int32_t ny, int32_t nz, int32_t order, const float* rot, const float* trans, int64_t first, int64_t count,
                   float* sdf_out, float* workspace, int64_t chunk_points, void* stream);

/* ------------------------------------------------------------------------------------------------
 * Bubble-PDF update (row N4) -- VolumeRenderSystem.update_pdf fused with the error it is fed (model/trainer/recon.py:142-152,
 * :195-199 in the initial sweep over all images, :246-252 every training step):
 *   channels == 1: v = |pred - target|                         (criterion DEPTH: depth_values vs depth image)
 *   channels == 3: v = mean_c |clamp(pred,0,1) - clamp(target,0,1)|   (criterion RGB)
 *   v = min(v, pdf_max) unless pdf_max is NaN / inf ("None");  v = 0 where v < pdf_prune;
 *   link = pointlinks[pixel];  pdf[link] = v where link != -1
 *   pixel_idx (n) global pixel indices, or NULL for the run first_pixel .. first_pixel+n-1 (a split of one image)
 *   n_bad: device counter of out-of-range pixels / links (skipped), or NULL.
 * ---------------------------------------------------------------------------------------------- */
int i2sdf_pdf_update(const float* pred, const float* target, int32_t channels, const int64_t* pixel_idx, int64_t first_pixel,
                     int64_t n, const int64_t* pointlinks, int64_t n_links, double pdf_max, double pdf_prune, float* pdf,
                     int64_t n_pdf, int32_t* n_bad, void* stream);

/* ------------------------------------------------------------------------------------------------
 * All random draws of one training forward in one launch (Philox4x32-10 keyed by `seed`; a different stream, not torch's) --
 * the torch.rand / randperm / randint / uniform_ calls of ray_sampler.py:60-66,176-177,223,234 and
 * model/network/__init__.py:177,184.  Every output may be NULL (not drawn):
 *   strat_u (B, n_eval) in [0,1)      stratified jitter            cdf_u (B, n_samples) in [0,1)   inverse-CDF draws
 *   extra_idx (max_iters, n_extra)    row it = n_extra distinct columns of [0, n_eval*(it+1)) in random order
 *                                     (= randperm(n)[:n_extra] for the row length the sampler has after `it` iterations)
 *   eik_idx (B) in [0, n_z)           eik_pts (B,3) in [-eik_radius, eik_radius)    nbr_off (B,3) in [-w, w), w = nbr_half_width
 * ---------------------------------------------------------------------------------------------- */
int i2sdf_training_draws(uint64_t seed, int64_t B, int32_t n_eval, int32_t n_samples, int32_t n_extra, int32_t max_iters,
                         int32_t n_z, float eik_radius, float nbr_half_width, float* strat_u, float* cdf_u,
                         int32_t* extra_idx, int32_t* eik_idx, float* eik_pts, float* nbr_off, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* I2SDF_H */
