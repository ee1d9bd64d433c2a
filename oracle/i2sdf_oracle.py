"""CPU oracle for the I2-SDF volume-rendering hot path.

TEST INFRASTRUCTURE ONLY.  This file is a functional restatement (plain torch on CPU, any
float dtype) of the algorithm the reference implements in Python/PyTorch.  It is imported
only by tests/, by __graft_entry__.smoke() and by bench.py's `cpu_baseline` leg -- always as
the checker / timed baseline, never by the product path (i2sdf_amd/ fails loudly when its
HIP library is missing; it has no CPU fallback).

Parity pinning: the reference has no tests/golden vectors of its own, so this
oracle is pinned against outputs of the reference itself, imported in the build container by
tests/golden/gen_golden.py -> tests/golden/*.npz (committed), and checked live against the
reference by tests/test_oracle_vs_reference.py whenever /root/reference is present.

Differences in *form* from the reference (not in arithmetic):
  * purely functional: parameters arrive as a state_dict with the reference's key names
    (`implicit_network.lin3.weight_v`, `density.beta`, ...), configuration as `NetCfg`;
  * every random draw the reference makes internally is an explicit input (`Draws`);
  * derivatives are available twice: through torch.autograd (what the reference does) and
    as explicit analytic sweeps (`sdf_analytic_*`, `composite_backward`) that mirror what the HIP
    kernels compute, so that kernel intermediates can be checked one by one.

Reference citations are `file:line` relative to the upstream repo root.
"""
from __future__ import annotations

import math
from dataclasses import dataclass, field
from typing import Dict, List, Optional, Sequence, Tuple

import torch

Tensor = torch.Tensor
SOFTPLUS_BETA = 100.0  # model/network/mlp.py:76
SOFTPLUS_THRESHOLD = 20.0  # torch.nn.Softplus default, used by mlp.py:76


# --------------------------------------------------------------------------------------
# configuration
# --------------------------------------------------------------------------------------
@dataclass
class SdfCfg:
    """model/network/mlp.py:11-31 (ImplicitNetwork ctor arguments that shape the arithmetic)."""
    dims: List[int]                 # hidden widths
    feature_size: int               # feature_vector_size
    skip_in: Tuple[int, ...] = ()
    multires: int = 6
    d_in: int = 3
    d_out: int = 1
    bias: float = 0.6

    @property
    def n_lin(self) -> int:
        return len(self.dims) + 1

    @property
    def pe_dim(self) -> int:
        return self.d_in + 2 * self.d_in * self.multires if self.multires > 0 else self.d_in

    def layer_shapes(self) -> List[Tuple[int, int]]:
        """(out, in) of every nn.Linear, mlp.py:45-53."""
        full = [self.pe_dim] + list(self.dims) + [self.d_out + self.feature_size]
        shapes = []
        for l in range(len(full) - 1):
            out = full[l + 1] - full[0] if (l + 1) in self.skip_in else full[l + 1]
            shapes.append((out, full[l]))
        return shapes


@dataclass
class RgbCfg:
    """model/network/mlp.py:160-206, 'nerf' mode only (mode 'idr' is commented out of the configs)."""
    dims: List[int]
    feature_size: int
    multires_view: int = 4
    d_out: int = 3

    @property
    def n_lin(self) -> int:
        return len(self.dims) + 1

    @property
    def pe_dim(self) -> int:
        return 3 + 6 * self.multires_view

    def layer_shapes(self) -> List[Tuple[int, int]]:
        full = [self.pe_dim + self.feature_size] + list(self.dims) + [self.d_out]
        return [(full[l + 1], full[l]) for l in range(len(full) - 1)]


@dataclass
class LightCfg:
    """model/network/__init__.py:29-32: ImplicitNetwork(0, 0, d_in=fvs, d_out=1, dims, no PE, sigmoid)."""
    dims: List[int]
    feature_size: int

    def layer_shapes(self) -> List[Tuple[int, int]]:
        full = [self.feature_size] + list(self.dims) + [1]
        return [(full[l + 1], full[l]) for l in range(len(full) - 1)]


@dataclass
class SamplerCfg:
    """model/network/ray_sampler.py:47-61."""
    near: float = 0.0
    N_samples: int = 64
    N_samples_eval: int = 128
    N_samples_extra: int = 32
    eps: float = 0.1
    beta_iters: int = 10
    max_total_iters: int = 5
    add_tiny: float = 1.0e-6


@dataclass
class NetCfg:
    sdf: SdfCfg
    rgb: RgbCfg
    sampler: SamplerCfg
    scene_bounding_sphere: float = 3.0
    beta_min: float = 1.0e-4
    light: Optional[LightCfg] = None
    use_normal: bool = True

    @staticmethod
    def from_conf(conf) -> "NetCfg":
        """Read the same keys model/network/__init__.py:20-47 reads from the yaml `model:` node."""
        g = lambda node, k, d=None: (node[k] if k in node else d)
        fvs = int(conf["feature_vector_size"])
        inet, rnet = conf["implicit_network"], conf["rendering_network"]
        sdf = SdfCfg(dims=list(inet["dims"]), feature_size=fvs, skip_in=tuple(g(inet, "skip_in", ())),
                     multires=int(g(inet, "multires", 0)), d_in=int(inet["d_in"]), d_out=int(inet["d_out"]),
                     bias=float(g(inet, "bias", 1.0)))
        rgb = RgbCfg(dims=list(rnet["dims"]), feature_size=fvs, multires_view=int(g(rnet, "multires", 0)),
                     d_out=int(rnet["d_out"]))
        rs = conf["ray_sampler"]
        sam = SamplerCfg(near=float(rs["near"]), N_samples=int(rs["N_samples"]), N_samples_eval=int(rs["N_samples_eval"]),
                         N_samples_extra=int(rs["N_samples_extra"]), eps=float(rs["eps"]), beta_iters=int(rs["beta_iters"]),
                         max_total_iters=int(rs["max_total_iters"]), add_tiny=float(g(rs, "add_tiny", 0.0)))
        light = None
        if "light_network" in conf:
            light = LightCfg(dims=list(conf["light_network"]["dims"]), feature_size=fvs)
        dens = conf["density"]
        return NetCfg(sdf=sdf, rgb=rgb, sampler=sam, scene_bounding_sphere=float(g(conf, "scene_bounding_sphere", 1.0)),
                      beta_min=float(g(dens, "beta_min", 1e-4)), light=light, use_normal=bool(g(conf, "use_normal", False)))


def synthetic_cfg(light: bool = False) -> NetCfg:
    """config/synthetic.yml:32-74 (and synthetic_light_mask.yml deltas) as literals."""
    if light:
        sdf = SdfCfg(dims=[256] * 6, feature_size=256, skip_in=(3,), multires=6)
        rgb = RgbCfg(dims=[256] * 3, feature_size=256, multires_view=4)
        return NetCfg(sdf=sdf, rgb=rgb, sampler=SamplerCfg(), light=LightCfg(dims=[128], feature_size=256))
    sdf = SdfCfg(dims=[256] * 8, feature_size=256, skip_in=(4,), multires=6)
    rgb = RgbCfg(dims=[256] * 4, feature_size=256, multires_view=4)
    return NetCfg(sdf=sdf, rgb=rgb, sampler=SamplerCfg())


def plumbing_cfg(skip: bool = False, light: bool = False) -> NetCfg:
    """BASELINE.json configs[0]: 2-layer x 64 SDF MLP, 16 samples/ray (cfg 1)."""
    sdf = SdfCfg(dims=[64, 64, 64] if skip else [64, 64], feature_size=64, skip_in=(2,) if skip else (), multires=6)
    rgb = RgbCfg(dims=[64, 64], feature_size=64, multires_view=4)
    sam = SamplerCfg(N_samples=16, N_samples_eval=32, N_samples_extra=8)
    return NetCfg(sdf=sdf, rgb=rgb, sampler=sam, light=LightCfg(dims=[32], feature_size=64) if light else None)


# --------------------------------------------------------------------------------------
# parameter initialisation (reference scheme) -- used for synthetic random-weight networks
# --------------------------------------------------------------------------------------
def init_params(cfg: NetCfg, seed: int = 0, dtype=torch.float32) -> Dict[str, Tensor]:
    """Random-weight state_dict with the reference's init distributions and key names.

    SDF net: geometric init, model/network/mlp.py:55-69.  Radiance / light nets: nn.Linear default
    (kaiming-uniform a=sqrt(5) == U(-1/sqrt(in), 1/sqrt(in)) for weight and bias).  weight_norm
    (mlp.py:71-72) stores v = W and g = row norms of W.  density.beta = 0.1 (config/synthetic.yml:60-61).
    Same distributions as the reference, not the same random stream.
    """
    gen = torch.Generator().manual_seed(seed)
    sd: Dict[str, Tensor] = {}

    def put(prefix, W, b):
        sd[prefix + ".bias"] = b.to(dtype)
        sd[prefix + ".weight_g"] = W.norm(dim=1, keepdim=True).to(dtype)
        sd[prefix + ".weight_v"] = W.to(dtype)

    shapes = cfg.sdf.layer_shapes()
    pe = cfg.sdf.pe_dim
    for l, (out, inn) in enumerate(shapes):
        if l == len(shapes) - 1:
            W = torch.randn(out, inn, generator=gen, dtype=torch.float64) * 1e-4 + math.sqrt(math.pi) / math.sqrt(inn)
            b = torch.full((out,), -cfg.sdf.bias, dtype=torch.float64)
        else:
            W = torch.randn(out, inn, generator=gen, dtype=torch.float64) * (math.sqrt(2) / math.sqrt(out))
            b = torch.zeros(out, dtype=torch.float64)
            if cfg.sdf.multires > 0 and l == 0:
                W[:, 3:] = 0.0
            elif cfg.sdf.multires > 0 and l in cfg.sdf.skip_in:
                W[:, -(pe - 3):] = 0.0
        put(f"implicit_network.lin{l}", W, b)

    def default_linear(prefix, out, inn):
        bound = 1.0 / math.sqrt(inn)
        W = (torch.rand(out, inn, generator=gen, dtype=torch.float64) * 2 - 1) * bound
        b = (torch.rand(out, generator=gen, dtype=torch.float64) * 2 - 1) * bound
        put(prefix, W, b)

    for l, (out, inn) in enumerate(cfg.rgb.layer_shapes()):
        default_linear(f"rendering_network.lin{l}", out, inn)
    if cfg.light is not None:
        for l, (out, inn) in enumerate(cfg.light.layer_shapes()):
            default_linear(f"light_network.lin{l}", out, inn)
    sd["density.beta"] = torch.tensor(0.1, dtype=dtype)
    return sd


def perturb_params(sd: Dict[str, Tensor], scale: float = 0.05, seed: int = 1) -> Dict[str, Tensor]:
    """Break the special structure of the geometric init (zero columns, zero biases) for stronger tests."""
    gen = torch.Generator().manual_seed(seed)
    out = {}
    for k, v in sd.items():
        if k == "density.beta":
            out[k] = v.clone()
            continue
        ref = v.abs().mean().clamp_min(1e-3).item()
        out[k] = v + torch.randn(v.shape, generator=gen, dtype=torch.float64).to(v.dtype) * scale * ref
    return out


# --------------------------------------------------------------------------------------
# M1  positional encoding -- model/network/embedder.py:6-38,138-152
# --------------------------------------------------------------------------------------
def positional_encode(x: Tensor, n_freqs: int) -> Tensor:
    """[x, sin(2^0 x), cos(2^0 x), ..., sin(2^{L-1} x), cos(2^{L-1} x)], each block 3 wide."""
    if n_freqs <= 0:
        return x
    parts = [x]
    for k in range(n_freqs):
        f = float(2 ** k)  # embedder.py:24 -- 2**linspace(0, L-1, L): exact powers of two
        parts.append(torch.sin(x * f))
        parts.append(torch.cos(x * f))
    return torch.cat(parts, dim=-1)


def effective_weight(sd: Dict[str, Tensor], prefix: str) -> Tensor:
    """torch.nn.utils.weight_norm with dim=0: W = g * v / ||v||_row  (mlp.py:71-72).
    Evaluated with the same torch primitive nn.utils.weight_norm dispatches to, so that the oracle's
    weights are bit-identical to the reference's (a hand-written v*(g/||v||) differs by 1 ulp, which
    the ill-conditioned inverse-CDF sampler amplifies to ~1e-3 in individual z values)."""
    v, g = sd[prefix + ".weight_v"], sd[prefix + ".weight_g"]
    return torch._weight_norm(v, g, 0)


def softplus100(a: Tensor) -> Tensor:
    """nn.Softplus(beta=100): log1p(exp(100 a))/100, identity where 100 a > 20 (mlp.py:76)."""
    return torch.nn.functional.softplus(a, beta=SOFTPLUS_BETA, threshold=SOFTPLUS_THRESHOLD)


# --------------------------------------------------------------------------------------
# M3/M4  SDF network -- model/network/mlp.py:84-151
# --------------------------------------------------------------------------------------
def sdf_forward(sd: Dict[str, Tensor], cfg: SdfCfg, x: Tensor, prefix: str = "implicit_network") -> Tensor:
    """ImplicitNetwork.forward (mlp.py:84-105): (M,3) -> (M, 1+feature)."""
    p = positional_encode(x, cfg.multires)
    h = p
    n = cfg.n_lin
    for l in range(n):
        if l in cfg.skip_in:
            h = torch.cat([h, p], dim=1) / math.sqrt(2)
        h = torch.nn.functional.linear(h, effective_weight(sd, f"{prefix}.lin{l}"), sd[f"{prefix}.lin{l}.bias"])
        if l < n - 1:
            h = softplus100(h)
    return h


def sdf_outputs(sd, cfg: SdfCfg, x: Tensor, create_graph: bool = False):
    """ImplicitNetwork.get_outputs (mlp.py:123-143) with sdf_bounding_sphere == 0 (network/__init__.py:26):
    returns sdf (M,1), feature (M,F), gradient d sdf / d x (M,3)."""
    x = x.detach().clone().requires_grad_(True)
    with torch.enable_grad():
        out = sdf_forward(sd, cfg, x)
        sdf = out[:, :1]
        grad = torch.autograd.grad(sdf, x, torch.ones_like(sdf), create_graph=create_graph, retain_graph=True)[0]
    return sdf, out[:, 1:], grad


def sdf_gradient(sd, cfg: SdfCfg, x: Tensor, create_graph: bool = False) -> Tensor:
    """ImplicitNetwork.gradient (mlp.py:107-118)."""
    return sdf_outputs(sd, cfg, x, create_graph)[2]


# --------------------------------------------------------------------------------------
# M5  radiance network ('nerf' mode) -- model/network/mlp.py:208-229
# --------------------------------------------------------------------------------------
# Test hook: `with relu_hook(fn):` routes every ReLU of the radiance net through fn(pre_activation, layer) -> activation.
# Used by the parity tests to (a) record the pre-activations and (b) force the backward mask of units whose pre-activation is
# zero within fp32 rounding -- there the reference's own mask is decided by rounding noise (tests/helpers.py: relu_flip_analysis).
_RELU_HOOK = None


class relu_hook:
    def __init__(self, fn):
        self.fn = fn

    def __enter__(self):
        global _RELU_HOOK
        self.prev, _RELU_HOOK = _RELU_HOOK, self.fn
        return self

    def __exit__(self, *exc):
        global _RELU_HOOK
        _RELU_HOOK = self.prev


def rgb_forward(sd, cfg: RgbCfg, view_dirs: Tensor, feat: Tensor, prefix: str = "rendering_network") -> Tensor:
    h = torch.cat([positional_encode(view_dirs, cfg.multires_view), feat], dim=-1)
    for l in range(cfg.n_lin):
        h = torch.nn.functional.linear(h, effective_weight(sd, f"{prefix}.lin{l}"), sd[f"{prefix}.lin{l}.bias"])
        if l < cfg.n_lin - 1:
            h = torch.relu(h) if _RELU_HOOK is None else _RELU_HOOK(h, l)
    return torch.sigmoid(h)


# --------------------------------------------------------------------------------------
# M6  light-mask head -- model/network/__init__.py:29-32,162-170
# --------------------------------------------------------------------------------------
def light_forward(sd, cfg: LightCfg, feat: Tensor, prefix: str = "light_network") -> Tensor:
    """sigmoid(W1 softplus100(W0 relu(feat).detach() + b0) + b1): an ImplicitNetwork without PE."""
    h = torch.relu(feat).detach()
    n = len(cfg.dims) + 1
    for l in range(n):
        h = torch.nn.functional.linear(h, effective_weight(sd, f"{prefix}.lin{l}"), sd[f"{prefix}.lin{l}.bias"])
        if l < n - 1:
            h = softplus100(h)
    return torch.sigmoid(h)


# --------------------------------------------------------------------------------------
# D  Laplace density -- model/network/density.py:16-30
# --------------------------------------------------------------------------------------
def get_beta(sd, cfg: NetCfg) -> Tensor:
    return sd["density.beta"].abs() + cfg.beta_min


def laplace_density(sdf: Tensor, beta) -> Tensor:
    """(1/beta) * (0.5 + 0.5 * sign(s) * expm1(-|s|/beta))."""
    return (1.0 / beta) * (0.5 + 0.5 * sdf.sign() * torch.expm1(-sdf.abs() / beta))


# --------------------------------------------------------------------------------------
# R1/R2  rays -- utils/rend_util.py:92-147 ; model/network/__init__.py:88-93
# --------------------------------------------------------------------------------------
def quat_to_rot(q: Tensor) -> Tensor:
    """utils/rend_util.py:150-167; q = (qr, qi, qj, qk), normalised first."""
    q = torch.nn.functional.normalize(q, dim=1)
    qr, qi, qj, qk = q[:, 0], q[:, 1], q[:, 2], q[:, 3]
    R = torch.stack([
        1 - 2 * (qj ** 2 + qk ** 2), 2 * (qj * qi - qk * qr), 2 * (qi * qk + qr * qj),
        2 * (qj * qi + qk * qr), 1 - 2 * (qi ** 2 + qk ** 2), 2 * (qj * qk - qi * qr),
        2 * (qk * qi - qj * qr), 2 * (qj * qk + qi * qr), 1 - 2 * (qi ** 2 + qj ** 2)], dim=1)
    return R.reshape(-1, 3, 3)


def get_camera_params(uv: Tensor, pose: Tensor, intrinsics: Tensor) -> Tuple[Tensor, Tensor]:
    """uv (B,P,2), pose (B,4,4) cam->world or (B,7) [quaternion, translation], K (B,4,4)
    -> un-normalised ray dirs (B,P,3), cam_loc (B,3).  utils/rend_util.py:92-120."""
    if pose.dim() == 2 and pose.shape[1] == 7:                 # rend_util.py:93-98
        p = torch.eye(4, dtype=pose.dtype, device=pose.device).repeat(pose.shape[0], 1, 1)
        p[:, :3, :3] = quat_to_rot(pose[:, :4])
        p[:, :3, 3] = pose[:, 4:]
        pose = p
    fx, fy = intrinsics[:, 0, 0:1], intrinsics[:, 1, 1:2]
    cx, cy = intrinsics[:, 0, 2:3], intrinsics[:, 1, 2:3]
    sk = intrinsics[:, 0, 1:2]
    x, y = uv[:, :, 0], uv[:, :, 1]
    z = torch.ones_like(x)
    x_l = (x - cx + cy * sk / fy - sk * y / fy) / fx * z      # rend_util.py:143
    y_l = (y - cy) / fy * z                                    # rend_util.py:144
    pts = torch.stack([x_l, y_l, z, torch.ones_like(z)], dim=-1)      # (B,P,4)
    world = torch.bmm(pose, pts.permute(0, 2, 1)).permute(0, 2, 1)[:, :, :3]  # rend_util.py:116
    cam_loc = pose[:, :3, 3]
    return world - cam_loc[:, None, :], cam_loc


def pixel_uv(height: int, width: int) -> Tensor:
    """dataset/train_dataset.py:67-70: flipped mgrid -> (H*W, 2) with uv = (column, row)."""
    rows, cols = torch.meshgrid(torch.arange(height), torch.arange(width), indexing="ij")
    return torch.stack([cols.reshape(-1), rows.reshape(-1)], dim=1).float()


def ray_batch(tables: Dict[str, Tensor], img_res, tidx: Tensor):
    """ReconDataset.__getitem__ + collate_fn (dataset/train_dataset.py:169-209) for a list of global pixel indices.
    tables: intrinsics_all, pose_all, rgb_images and optionally mask_images, lightmask_images, depth_images, depth_masks,
    normal_images, normal_masks (the dataset's attribute names / layouts).
    -> (tidx, image_idx, sample{uv (B,1,2), intrinsics (B,4,4), pose}, ground_truth{...})."""
    hw = img_res[0] * img_res[1]
    uv = pixel_uv(img_res[0], img_res[1])
    pidx, idx = tidx % hw, tidx // hw
    sample = {"uv": uv[pidx].unsqueeze(1), "intrinsics": tables["intrinsics_all"][idx], "pose": tables["pose_all"][idx]}
    gt = {"rgb": tables["rgb_images"][idx, pidx]}
    if "mask_images" in tables:
        gt["mask"] = tables["mask_images"][idx, pidx]
    if "lightmask_images" in tables:
        gt["light_mask"] = tables["lightmask_images"][idx, pidx]
    if "depth_images" in tables:
        gt["depth"], gt["depth_mask"] = tables["depth_images"][idx, pidx], tables["depth_masks"][idx, pidx]
    if "normal_images" in tables:
        gt["normal"], gt["normal_mask"] = tables["normal_images"][idx, pidx], tables["normal_masks"][idx, pidx]
    return tidx, idx, sample, gt


def prepare_rays(uv, pose, intrinsics):
    """network/__init__.py:86-93 -> cam_loc (N,3), unit dirs (N,3), ||raw dir|| (N,)."""
    dirs, cam = get_camera_params(uv, pose, intrinsics)
    B, P, _ = dirs.shape
    cam = cam.unsqueeze(1).repeat(1, P, 1).reshape(-1, 3)
    dirs = dirs.reshape(-1, 3)
    norm = torch.linalg.vector_norm(dirs, dim=1)
    dirs = torch.nn.functional.normalize(dirs, dim=1)
    return cam, dirs, norm


def get_sphere_intersections(cam_loc: Tensor, dirs: Tensor, r: float) -> Tensor:
    """utils/rend_util.py:211-227 (only reached when a bg network exists; kept for completeness).
    Raises instead of exit()."""
    dot = (dirs * cam_loc).sum(-1, keepdim=True)
    under = dot ** 2 - (cam_loc.norm(2, 1, keepdim=True) ** 2 - r ** 2)
    if (under <= 0).any():
        raise ValueError("BOUNDING SPHERE PROBLEM")
    t = torch.sqrt(under) * torch.tensor([-1.0, 1.0], dtype=dirs.dtype, device=dirs.device) - dot
    return t.clamp_min(0.0)


# --------------------------------------------------------------------------------------
# S0/S/S1  sampler -- model/network/ray_sampler.py
# --------------------------------------------------------------------------------------
@dataclass
class Draws:
    """Every random draw of one training forward, in the reference's call order.
    strat_u   (B, N_samples_eval)  ray_sampler.py:39   torch.rand
    cdf_u     (B, N_samples)       ray_sampler.py:190  torch.rand
    extra_idx (N_samples_extra,)   ray_sampler.py:223  torch.randperm(n)[:k]  (CPU generator, shared by all rays)
    eik_idx   (B,)                 ray_sampler.py:233  torch.randint
    eik_pts   (B,3) in [-R,R]      network/__init__.py:178  uniform_
    nbr_off   (B,3) in +-0.005     network/__init__.py:186  uniform_
    """
    strat_u: Optional[Tensor] = None
    cdf_u: Optional[Tensor] = None
    extra_idx: Optional[Tensor] = None
    eik_idx: Optional[Tensor] = None
    eik_pts: Optional[Tensor] = None
    nbr_off: Optional[Tensor] = None


def uniform_z_vals(n_rays: int, cfg: NetCfg, training: bool, strat_u: Optional[Tensor], dtype, device=None) -> Tensor:
    """UniformSampler.get_z_vals (ray_sampler.py:22-43), take_sphere_intersection=False."""
    near, far = cfg.sampler.near, 2.0 * cfg.scene_bounding_sphere
    t = torch.linspace(0.0, 1.0, steps=cfg.sampler.N_samples_eval, dtype=dtype, device=device)
    z = (near * (1.0 - t) + far * t).unsqueeze(0).repeat(n_rays, 1)
    if training:
        mids = 0.5 * (z[:, 1:] + z[:, :-1])
        upper = torch.cat([mids, z[:, -1:]], -1)
        lower = torch.cat([z[:, :1], mids], -1)
        z = lower + (upper - lower) * strat_u
    return z


def d_star_bound(z_vals: Tensor, sdf_rows: Tensor) -> Tensor:
    """Theorem-1 bound per interval (ray_sampler.py:99-114)."""
    d = sdf_rows
    a = z_vals[:, 1:] - z_vals[:, :-1]
    b, c = d[:, :-1].abs(), d[:, 1:].abs()
    first = a.pow(2) + b.pow(2) <= c.pow(2)
    second = a.pow(2) + c.pow(2) <= b.pow(2)
    s = (a + b + c) / 2.0
    area = s * (s - a) * (s - b) * (s - c)
    mask = ~first & ~second & (b + c - a > 0)
    first = first & ~second
    ds = first * b + second * c + torch.nan_to_num((2.0 * torch.sqrt(area)) / a) * mask
    return (d[:, 1:].sign() * d[:, :-1].sign() == 1) * ds


def error_bound(beta, sdf_rows: Tensor, dists: Tensor, d_star: Tensor) -> Tensor:
    """ErrorBoundSampler.get_error_bound (ray_sampler.py:243-251) -> (B,)."""
    dens = laplace_density(sdf_rows, beta)
    sfe = torch.cat([torch.zeros_like(dists[:, :1]), dists * dens[:, :-1]], dim=-1)
    integral = torch.cumsum(sfe, dim=-1)
    eps_sec = torch.exp(-d_star / beta) * (dists ** 2.0) / (4 * beta ** 2)
    eint = torch.cumsum(eps_sec, dim=-1)
    bo = (torch.clamp(torch.exp(eint), max=1.0e6) - 1.0) * torch.exp(-integral[:, :-1])
    return bo.max(-1)[0]


def inverse_cdf(bins: Tensor, cdf: Tensor, u: Tensor) -> Tensor:
    """ray_sampler.py:193-207: searchsorted(right=True), clamp, lerp with denom<1e-5 -> 1."""
    inds = torch.searchsorted(cdf, u.contiguous(), right=True)
    below = (inds - 1).clamp_min(0)
    above = inds.clamp_max(cdf.shape[-1] - 1)
    c0, c1 = torch.gather(cdf, 1, below), torch.gather(cdf, 1, above)
    b0, b1 = torch.gather(bins, 1, below), torch.gather(bins, 1, above)
    denom = c1 - c0
    denom = torch.where(denom < 1e-5, torch.ones_like(denom), denom)
    return b0 + (u - c0) / denom * (b1 - b0)


@dataclass
class SamplerTrace:
    iters: int = 0
    betas: List[Tensor] = field(default_factory=list)        # per iteration, after bisection (B,)
    z_rows: List[Tensor] = field(default_factory=list)       # per iteration, the row the bound was evaluated on
    sdf_rows: List[Tensor] = field(default_factory=list)
    new_samples: List[Tensor] = field(default_factory=list)  # per iteration, inverse-CDF output


def sample_z_vals(sd, cfg: NetCfg, dirs: Tensor, cam_loc: Tensor, training: bool, draws: Optional[Draws] = None,
                  force_iters: Optional[int] = None, trace: Optional[SamplerTrace] = None):
    """ErrorBoundSampler.get_z_vals (ray_sampler.py:67-241), inverse_sphere_bg=False.

    Returns z_vals (B, N_samples + N_samples_extra + 2) sorted (last column = far) and z_samples_eik (B,1).
    `force_iters=k` replaces the data-dependent, batch-global `beta.max() > beta0` test (ray_sampler.py:151)
    by "exactly k evaluations of the loop body" -- used for fixed-work throughput runs only.
    """
    sc = cfg.sampler
    B, dtype, dev = dirs.shape[0], dirs.dtype, dirs.device
    draws = draws or Draws()
    with torch.no_grad():
        beta0 = get_beta(sd, cfg).detach().to(dtype)
        z_vals = uniform_z_vals(B, cfg, training, draws.strat_u, dtype, dirs.device)
        samples, samples_idx = z_vals, None
        dists = z_vals[:, 1:] - z_vals[:, :-1]
        bound = (1.0 / (4.0 * torch.log(torch.tensor(sc.eps + 1.0)))) * (dists ** 2.0).sum(-1)  # ray_sampler.py:76
        beta = torch.sqrt(bound)
        total_iters, not_converge, sdf = 0, True, None
        while not_converge and total_iters < sc.max_total_iters:
            pts = (cam_loc.unsqueeze(1) + samples.unsqueeze(2) * dirs.unsqueeze(1)).reshape(-1, 3)
            s_new = sdf_forward(sd, cfg.sdf, pts)[:, :1]
            if samples_idx is not None:
                merged = torch.cat([sdf.reshape(-1, z_vals.shape[1] - samples.shape[1]), s_new.reshape(-1, samples.shape[1])], -1)
                sdf = torch.gather(merged, 1, samples_idx).reshape(-1, 1)
            else:
                sdf = s_new
            d = sdf.reshape(z_vals.shape)
            dists = z_vals[:, 1:] - z_vals[:, :-1]
            d_star = d_star_bound(z_vals, d)
            # line search on beta (ray_sampler.py:118-132)
            err = error_bound(beta0, d, dists, d_star)
            ok = err <= sc.eps
            beta = beta * ~ok + beta0 * ok
            b_min, b_max = beta0.unsqueeze(0).repeat(B), beta
            for _ in range(sc.beta_iters):
                b_mid = (b_min + b_max) / 2.0
                err = error_bound(b_mid.unsqueeze(-1), d, dists, d_star)
                ok = err <= sc.eps
                b_max = b_max * ~ok + b_mid * ok
                b_min = b_min * ok + b_mid * ~ok
            beta = b_max
            dens = laplace_density(d, beta.unsqueeze(-1))
            dists_e = torch.cat([dists, torch.full([B, 1], 1e10, dtype=dtype, device=dev)], -1)
            fe = dists_e * dens
            sfe = torch.cat([torch.zeros(B, 1, dtype=dtype, device=dev), fe[:, :-1]], dim=-1)
            alpha = 1 - torch.exp(-fe)
            trans = torch.exp(-torch.cumsum(sfe, dim=-1))
            weights = alpha * trans
            total_iters += 1
            if force_iters is None:
                not_converge = bool(beta.max() > beta0)            # ray_sampler.py:151 (batch-global)
            else:
                not_converge = total_iters < force_iters
            more = not_converge and total_iters < sc.max_total_iters
            if trace is not None:
                trace.betas.append(beta.clone()); trace.z_rows.append(z_vals.clone()); trace.sdf_rows.append(d.clone())
            if more:
                N = sc.N_samples_eval
                eps_sec = torch.exp(-d_star / beta.unsqueeze(-1)) * (dists_e[:, :-1] ** 2.0) / (4 * beta.unsqueeze(-1) ** 2)
                eint = torch.cumsum(eps_sec, dim=-1)
                bo = (torch.clamp(torch.exp(eint), max=1.0e6) - 1.0) * trans[:, :-1]
                pdf = bo + sc.add_tiny
            else:
                N = sc.N_samples
                pdf = weights[..., :-1] + 1e-5
            pdf = pdf / torch.sum(pdf, -1, keepdim=True)
            cdf = torch.cumsum(pdf, -1)
            cdf = torch.cat([torch.zeros_like(cdf[..., :1]), cdf], -1)
            if more or not training:
                u = torch.linspace(0.0, 1.0, steps=N, dtype=dtype, device=dev).unsqueeze(0).repeat(B, 1)
            else:
                u = draws.cdf_u
            samples = inverse_cdf(z_vals, cdf, u)
            if trace is not None:
                trace.new_samples.append(samples.clone())
            if more:
                z_vals, samples_idx = torch.sort(torch.cat([z_vals, samples], -1), -1)
        if trace is not None:
            trace.iters = total_iters
        near = torch.full((B, 1), sc.near, dtype=dtype, device=dev)
        far = torch.full((B, 1), 2.0 * cfg.scene_bounding_sphere, dtype=dtype, device=dev)
        if sc.N_samples_extra > 0:
            if training:
                idx = draws.extra_idx.long()
                if idx.dim() == 2:      # one randperm row per possible loop length (what a caller that cannot know the iteration count
                    idx = idx[total_iters - 1]      # in advance draws): row it-1 indexes the N_eval*it depths of a loop that ran `it` times
            else:
                idx = torch.linspace(0, z_vals.shape[1] - 1, sc.N_samples_extra).long().to(dev)   # ray_sampler.py:225
            extra = torch.cat([near, far, z_vals[:, idx]], -1)
        else:
            extra = torch.cat([near, far], -1)
        z_out, _ = torch.sort(torch.cat([samples, extra], -1), -1)
        if training and draws.eik_idx is not None:
            z_eik = torch.gather(z_out, 1, draws.eik_idx.long().unsqueeze(-1))
        else:
            z_eik = z_out[:, :1].clone()  # eval: the reference draws one too (ray_sampler.py:233) but never uses it
    return z_out, z_eik


# --------------------------------------------------------------------------------------
# V/C  volume rendering + composite -- model/network/__init__.py:223-240, 118-125, 204-219
# --------------------------------------------------------------------------------------
def volume_weights(z_vals: Tensor, z_max: Tensor, sdf: Tensor, beta) -> Tuple[Tensor, Tensor]:
    dens = laplace_density(sdf, beta).reshape(-1, z_vals.shape[1])
    dists = torch.cat([z_vals[:, 1:] - z_vals[:, :-1], z_max.unsqueeze(-1) - z_vals[:, -1:]], -1)
    fe = dists * dens
    sfe = torch.cat([torch.zeros_like(fe[:, :1]), fe], dim=-1)
    alpha = 1 - torch.exp(-fe)
    trans = torch.exp(-torch.cumsum(sfe, dim=-1))
    return alpha * trans[:, :-1], trans[:, -1]


# --------------------------------------------------------------------------------------
# F  full forward -- model/network/__init__.py:80-221 (bg disabled)
# --------------------------------------------------------------------------------------
def network_forward(sd, cfg: NetCfg, inputs: Dict[str, Tensor], training: bool, draws: Optional[Draws] = None,
                    predict_only: bool = False, force_iters: Optional[int] = None, z_override=None,
                    trace: Optional[SamplerTrace] = None) -> Dict[str, Tensor]:
    """I2SDFNetwork.forward.  `z_override=(z_vals(B,n+1), z_eik(B,1))` bypasses the sampler (dense-N runs/tests).
    The output dict additionally carries `_z_vals` / `_sdf` / `_grad` (prefixed with `_`, not part of the
    reference's dict) so tests can compare intermediates."""
    cam, dirs, dnorm = prepare_rays(inputs["uv"], inputs["pose"], inputs["intrinsics"])
    draws = draws or Draws()
    if z_override is None:
        z_all, z_eik = sample_z_vals(sd, cfg, dirs, cam, training, draws, force_iters, trace)
    else:
        z_all, z_eik = z_override
    z_max, z_vals = z_all[:, -1], z_all[:, :-1]
    n = z_vals.shape[1]
    pts = (cam.unsqueeze(1) + z_vals.unsqueeze(2) * dirs.unsqueeze(1)).reshape(-1, 3)
    dirs_flat = dirs.unsqueeze(1).repeat(1, n, 1).reshape(-1, 3)
    returns_grad = cfg.use_normal or (not training)                   # network/__init__.py:109
    if returns_grad:
        sdf, feat, grads = sdf_outputs(sd, cfg.sdf, pts, create_graph=training)
    else:
        o = sdf_forward(sd, cfg.sdf, pts)
        sdf, feat, grads = o[:, :1], o[:, 1:], None
    rgb = rgb_forward(sd, cfg.rgb, dirs_flat, feat).reshape(-1, n, 3)
    w, _bg_t = volume_weights(z_vals, z_max, sdf, get_beta(sd, cfg))
    out = {
        "rgb_values": torch.sum(w.unsqueeze(-1) * rgb, 1),
        "depth_values": torch.sum(w * z_vals, 1) / torch.clamp(dnorm, min=1e-6),
        "weight_sum": torch.sum(w, -1, keepdim=True),
    }
    if cfg.light is not None:
        lm = light_forward(sd, cfg.light, feat).reshape(-1, n, 1)
        out["light_mask"] = torch.sum(w.unsqueeze(-1).detach() * lm, 1)
    out["_z_vals"], out["_sdf"], out["_grad"], out["_weights"] = z_all, sdf, grads, w
    if predict_only:
        return out
    if training:
        R = cfg.scene_bounding_sphere
        near_pts = (cam.unsqueeze(1) + z_eik.unsqueeze(2) * dirs.unsqueeze(1)).reshape(-1, 3)
        eik = torch.cat([draws.eik_pts, near_pts, near_pts + draws.nbr_off], 0)     # network/__init__.py:178-187
        g = sdf_gradient(sd, cfg.sdf, eik, create_graph=True)
        nb = near_pts.shape[0]
        out["grad_theta"] = g[: 2 * nb]
        nrm = torch.nn.functional.normalize(g[nb:], dim=1, eps=1e-6)
        out["diff_norm"] = torch.norm(nrm[:nb] - nrm[nb:], dim=1)
        if "pointcloud" in inputs:
            # network/__init__.py:196-201: one extra (random) camera location is appended and dropped again,
            # so it never influences the returned rows.
            out["surface_sdf"] = sdf_forward(sd, cfg.sdf, inputs["pointcloud"])[:, :1]
        if cfg.use_normal:
            nm = torch.nn.functional.normalize(grads, dim=-1).reshape(-1, n, 3)
            out["normal_values"] = torch.nn.functional.normalize(torch.sum(w.unsqueeze(-1).detach() * nm, 1), dim=-1)
    else:
        nm = torch.nn.functional.normalize(grads.detach(), dim=-1).reshape(-1, n, 3)
        out["normal_map"] = torch.nn.functional.normalize(torch.sum(w.unsqueeze(-1) * nm, 1), dim=-1)
    return out


# --------------------------------------------------------------------------------------
# N1  loss -- model/network/__init__.py:289-406
# --------------------------------------------------------------------------------------
@dataclass
class LossCfg:
    eikonal_weight: float = 0.1
    smooth_weight: float = 0.0
    mask_weight: float = 0.0
    depth_weight: float = 0.1
    normal_weight: float = 0.05
    angular_weight: float = 0.05       # default applies: the shipped configs omit it (network/__init__.py:290)
    bubble_weight: float = 0.0
    smooth_iter: Optional[int] = None
    light_mask_weight: float = 0.0


def i2sdf_loss(out: Dict[str, Tensor], gt: Dict[str, Tensor], lc: LossCfg, step: int = 0) -> Dict[str, Tensor]:
    zero = torch.zeros((), dtype=out["rgb_values"].dtype)
    rgb_loss = torch.nn.functional.l1_loss(out["rgb_values"], gt["rgb"].reshape(-1, 3))
    eik = ((out["grad_theta"].norm(2, dim=1) - 1) ** 2).mean() if "grad_theta" in out else zero
    smooth_on = lc.smooth_iter is None or step > lc.smooth_iter
    smooth = out["diff_norm"].mean() if (smooth_on and lc.smooth_weight > 0 and "diff_norm" in out) else zero
    if "mask" in gt and lc.mask_weight > 0:
        mask = torch.nn.functional.binary_cross_entropy(out["weight_sum"].clip(1e-3, 1 - 1e-3), gt["mask"])
    else:
        mask = zero
    if "depth" in gt and lc.depth_weight > 0:
        dm = gt["depth_mask"].flatten()
        depth = torch.nn.functional.mse_loss(out["depth_values"][dm], gt["depth"].flatten()[dm])
    else:
        depth = zero

    def normal_l1():
        nm = gt["normal_mask"].flatten()
        return torch.abs(1 - torch.sum(out["normal_values"][nm] * gt["normal"].reshape(-1, 3)[nm], dim=-1)).mean()

    normal = normal_l1() if ("normal" in gt and lc.normal_weight > 0) else zero
    angular = normal_l1() if ("normal" in gt and lc.angular_weight > 0) else zero   # quirk: L1 form again (:368-369)
    bubble = out["surface_sdf"].abs().mean() if ("surface_sdf" in out and lc.bubble_weight > 0) else zero
    if "light_mask" in out and lc.light_mask_weight > 0:
        lmask = torch.nn.functional.binary_cross_entropy(out["light_mask"].reshape(-1, 1).clip(1e-3, 1 - 1e-3),
                                                         gt["light_mask"].reshape(-1, 1))
    else:
        lmask = zero
    loss = (rgb_loss + lc.eikonal_weight * eik + lc.smooth_weight * smooth + lc.mask_weight * mask + lc.depth_weight * depth
            + lc.normal_weight * normal + lc.angular_weight * angular + lc.bubble_weight * bubble + lc.light_mask_weight * lmask)
    return {"loss": loss, "rgb_loss": rgb_loss, "eikonal_loss": eik, "smooth_loss": smooth, "mask_loss": mask,
            "depth_loss": depth, "normal_loss": normal, "angular_loss": angular, "bubble_loss": bubble,
            "light_mask_loss": lmask}


def get_psnr(a: Tensor, b: Tensor) -> Tensor:
    """utils/rend_util.py:13-22."""
    return -10.0 * torch.log(torch.mean((a - b) ** 2)) / math.log(10)


def training_step_grads(sd, cfg: NetCfg, inputs, gt, lc: LossCfg, draws: Draws, step: int = 0,
                        force_iters: Optional[int] = None, z_override=None):
    """forward + loss + backward through torch.autograd (what Lightning's loss.backward() does).
    Returns (outputs, loss dict, {param name: grad})."""
    params = {k: v.detach().clone().requires_grad_(True) for k, v in sd.items()}
    out = network_forward(params, cfg, inputs, True, draws, force_iters=force_iters, z_override=z_override)
    losses = i2sdf_loss(out, gt, lc, step)
    names = list(params.keys())
    grads = torch.autograd.grad(losses["loss"], [params[k] for k in names], allow_unused=True)
    return out, losses, {k: (g if g is not None else torch.zeros_like(params[k])) for k, g in zip(names, grads)}


# --------------------------------------------------------------------------------------
# marching-cubes grids and the bubble PDF (callers of the SDF forward / of predict_only rendering)
# --------------------------------------------------------------------------------------
def get_grid_uniform(resolution: int, grid_boundary=(-2.0, 2.0)):
    """utils/plots.py:440-451.  Returns {grid_points (n,3) float32 in np.meshgrid(x,y,z) ravel order, xyz float64 axes, ...}."""
    import numpy as np
    x = np.linspace(grid_boundary[0], grid_boundary[1], resolution)
    y = x
    z = x
    xx, yy, zz = np.meshgrid(x, y, z)
    grid_points = torch.tensor(np.vstack([xx.ravel(), yy.ravel(), zz.ravel()]).T, dtype=torch.float)
    return {"grid_points": grid_points, "shortest_axis_length": 2.0, "xyz": [x, y, z], "shortest_axis_index": 0}


def get_grid(points: Tensor, resolution: int, input_min=None, input_max=None, eps: float = 0.1):
    """utils/plots.py:453-489: `resolution` points along the shortest axis of the bounding box (+-eps), the same step along the others."""
    import numpy as np
    if input_min is None or input_max is None:
        input_min = torch.min(points, dim=0)[0].squeeze().numpy()
        input_max = torch.max(points, dim=0)[0].squeeze().numpy()
    bounding_box = input_max - input_min
    shortest_axis = int(np.argmin(bounding_box))
    if shortest_axis == 0:
        x = np.linspace(input_min[0] - eps, input_max[0] + eps, resolution)
        length = np.max(x) - np.min(x)
        y = np.arange(input_min[1] - eps, input_max[1] + length / (x.shape[0] - 1) + eps, length / (x.shape[0] - 1))
        z = np.arange(input_min[2] - eps, input_max[2] + length / (x.shape[0] - 1) + eps, length / (x.shape[0] - 1))
    elif shortest_axis == 1:
        y = np.linspace(input_min[1] - eps, input_max[1] + eps, resolution)
        length = np.max(y) - np.min(y)
        x = np.arange(input_min[0] - eps, input_max[0] + length / (y.shape[0] - 1) + eps, length / (y.shape[0] - 1))
        z = np.arange(input_min[2] - eps, input_max[2] + length / (y.shape[0] - 1) + eps, length / (y.shape[0] - 1))
    else:
        z = np.linspace(input_min[2] - eps, input_max[2] + eps, resolution)
        length = np.max(z) - np.min(z)
        x = np.arange(input_min[0] - eps, input_max[0] + length / (z.shape[0] - 1) + eps, length / (z.shape[0] - 1))
        y = np.arange(input_min[1] - eps, input_max[1] + length / (z.shape[0] - 1) + eps, length / (z.shape[0] - 1))
    xx, yy, zz = np.meshgrid(x, y, z)
    grid_points = torch.tensor(np.vstack([xx.ravel(), yy.ravel(), zz.ravel()]).T, dtype=torch.float)
    return {"grid_points": grid_points, "shortest_axis_length": length, "xyz": [x, y, z], "shortest_axis_index": shortest_axis}


def align_grid_points(grid_points: Tensor, vecs: Tensor, s_mean: Tensor) -> Tensor:
    """model/eval/recon.py:82-85: every grid point p -> vecs^T p + s_mean (the PCA frame of the coarse mesh back to the world)."""
    return torch.bmm(vecs.unsqueeze(0).repeat(grid_points.shape[0], 1, 1).transpose(1, 2), grid_points.unsqueeze(-1)).squeeze(-1) + s_mean


def grid_volume(z_flat, xyz):
    """model/eval/recon.py:53-54,94: the flat SDF vector as the (nx,ny,nz) volume marching cubes is run on."""
    return z_flat.reshape(xyz[1].shape[0], xyz[0].shape[0], xyz[2].shape[0]).transpose(1, 0, 2) if not torch.is_tensor(z_flat) \
        else z_flat.reshape(xyz[1].shape[0], xyz[0].shape[0], xyz[2].shape[0]).permute(1, 0, 2)


def pdf_error(criterion: str, model_outputs: Dict[str, Tensor], ground_truth: Dict[str, Tensor]) -> Tensor:
    """model/trainer/recon.py:195-199 / :248-252."""
    if criterion == "RGB":
        return (model_outputs["rgb_values"].detach().clamp(0, 1) - ground_truth["rgb"].clamp(0, 1)).abs().mean(dim=-1)
    return (model_outputs["depth_values"].detach() - ground_truth["depth"]).abs()


def update_pdf(pdf: Tensor, value: Tensor, idx: Tensor, pointlinks: Tensor, pdf_max: Optional[float], pdf_prune: float) -> None:
    """model/trainer/recon.py:142-152 (in place on `pdf`)."""
    value = value.clone()
    if pdf_max is not None:
        value = value.clamp(max=pdf_max)
    value[value < pdf_prune] = 0
    link = pointlinks[idx]
    mask = link != -1
    pdf[link[mask]] = value[mask]


# --------------------------------------------------------------------------------------
# Analytic restatement of what autograd does -- mirrors the HIP kernels
# --------------------------------------------------------------------------------------
def _sp_prime(a: Tensor) -> Tensor:
    """softplus100'(a) as torch's backward defines it: sigmoid(100 a), exactly 1 where 100 a > 20."""
    z = torch.exp(a * SOFTPLUS_BETA)
    return torch.where(a * SOFTPLUS_BETA > SOFTPLUS_THRESHOLD, torch.ones_like(a), z / (z + 1.0))


def _sp_second(a: Tensor) -> Tensor:
    """softplus100''(a): 100 sigma (1 - sigma), exactly 0 in the threshold branch."""
    s = _sp_prime(a)
    return torch.where(a * SOFTPLUS_BETA > SOFTPLUS_THRESHOLD, torch.zeros_like(a), SOFTPLUS_BETA * s * (1.0 - s))


def pe_jacobian_apply(x: Tensor, n_freqs: int, pbar: Tensor) -> Tensor:
    """(d PE / d x)^T pbar: (M, 3+6L) -> (M,3)."""
    out = pbar[:, :3].clone()
    for k in range(n_freqs):
        f = float(2 ** k)
        s, c = pbar[:, 3 + 6 * k: 6 + 6 * k], pbar[:, 6 + 6 * k: 9 + 6 * k]
        out = out + f * (torch.cos(f * x) * s - torch.sin(f * x) * c)
    return out


def pe_jacobian_forward(x: Tensor, n_freqs: int, nbar: Tensor) -> Tensor:
    """(d PE / d x) nbar: (M,3) -> (M, 3+6L)."""
    parts = [nbar]
    for k in range(n_freqs):
        f = float(2 ** k)
        parts.append(f * torch.cos(f * x) * nbar)
        parts.append(-f * torch.sin(f * x) * nbar)
    return torch.cat(parts, dim=-1)


def sdf_analytic_forward(sd, cfg: SdfCfg, x: Tensor, prefix: str = "implicit_network"):
    """A.1 + A.2: returns dict with a_l (pre-activations), u_l (layer inputs), abar_l (d sdf / d a_l), sdf, feat, n."""
    W = [effective_weight(sd, f"{prefix}.lin{l}") for l in range(cfg.n_lin)]
    b = [sd[f"{prefix}.lin{l}.bias"] for l in range(cfg.n_lin)]
    L, rs2 = cfg.n_lin, 1.0 / math.sqrt(2)
    p = positional_encode(x, cfg.multires)
    h, a, u = p, [], []
    for l in range(L):
        ul = torch.cat([h, p], 1) * rs2 if l in cfg.skip_in else h
        al = ul @ W[l].t() + b[l]
        u.append(ul); a.append(al)
        h = softplus100(al) if l < L - 1 else al
    abar: List[Optional[Tensor]] = [None] * L
    e0 = torch.zeros_like(a[-1]); e0[:, 0] = 1.0
    abar[L - 1] = e0
    pbar = torch.zeros_like(p)
    for l in range(L - 1, -1, -1):
        ubar = abar[l] @ W[l]
        if l in cfg.skip_in:
            hw = ubar.shape[1] - p.shape[1]
            hbar = ubar[:, :hw] * rs2
            pbar = pbar + ubar[:, hw:] * rs2
        else:
            hbar = ubar
        if l > 0:
            abar[l - 1] = hbar * _sp_prime(a[l - 1])
        else:
            pbar = pbar + hbar
    n = pe_jacobian_apply(x, cfg.multires, pbar)
    return {"W": W, "a": a, "u": u, "abar": abar, "p": p, "sdf": a[-1][:, :1], "feat": a[-1][:, 1:], "n": n}


def sdf_analytic_backward(sd, cfg: SdfCfg, x: Tensor, fw, sbar: Tensor, fbar: Tensor, nbar: Tensor,
                          prefix: str = "implicit_network", g2_in_sweep2: bool = False):
    """A.3: parameter gradients (dW effective, db) for upstream (sbar (M,1), fbar (M,F), nbar (M,3)),
    then weight-norm backward -> {name: grad}.  Three sweeps, exactly what the HIP backward does.
    g2_in_sweep2: the form the bf16x3 sweeps use since round 5 (csrc/x3.h: X3Sweep2Src) -- sweep 1 keeps only G(hbar_{l+1}) = G(abar_l) sigma_l,
    sweep 2 recovers G(abar_l) as its quotient by sigma_l (0 where sigma_l = 0) and forms G2(a_l) = G(abar_l) abar_l 100 (1 - sigma_l)."""
    L, rs2 = cfg.n_lin, 1.0 / math.sqrt(2)
    W, a, u, abar = fw["W"], fw["a"], fw["u"], fw["abar"]
    dW = [torch.zeros_like(w) for w in W]
    db = [torch.zeros(w.shape[0], dtype=w.dtype) for w in W]
    # sweep 1: adjoint of the n-chain, bottom-up
    Gp = pe_jacobian_forward(x, cfg.multires, nbar)
    Gh = Gp
    G2: List[Optional[Tensor]] = [None] * L
    Ghbar: List[Optional[Tensor]] = [None] * (L + 1)          # G(hbar_l), what sweep 1 stores (gus[l])
    for l in range(L):
        Gu = torch.cat([Gh, Gp], 1) * rs2 if l in cfg.skip_in else Gh
        dW[l] += abar[l].t() @ Gu
        Gabar = Gu @ W[l].t()
        if l < L - 1:
            sig = _sp_prime(a[l])
            # hbar_{l+1} = abar_l / sigma_l ; use the recurrence value directly to avoid 0/0
            hbar_next = _hbar_from_chain(fw, cfg, l + 1)
            G2[l] = Gabar * hbar_next * _sp_second(a[l])
            Gh = Gabar * sig
            Ghbar[l + 1] = Gh
    # sweep 2: ordinary backward, top-down
    Ga = torch.cat([sbar, fbar], 1)
    for l in range(L - 1, -1, -1):
        if l < L - 1:
            if g2_in_sweep2:
                sig = _sp_prime(a[l])
                Gabar = torch.where(sig > 0, Ghbar[l + 1] / torch.where(sig > 0, sig, torch.ones_like(sig)), torch.zeros_like(sig))
                Ga = Ga + Gabar * abar[l] * (SOFTPLUS_BETA * (1.0 - sig))
            else:
                Ga = Ga + G2[l]
        db[l] += Ga.sum(0)
        dW[l] += Ga.t() @ u[l]
        Gu = Ga @ W[l]
        if l in cfg.skip_in:
            hw = Gu.shape[1] - fw["p"].shape[1]
            Ghl = Gu[:, :hw] * rs2
        else:
            Ghl = Gu
        if l > 0:
            Ga = Ghl * _sp_prime(a[l - 1])
    grads = {}
    for l in range(L):
        v, g = sd[f"{prefix}.lin{l}.weight_v"], sd[f"{prefix}.lin{l}.weight_g"]
        dg, dv = weight_norm_backward(v, g, dW[l])
        grads[f"{prefix}.lin{l}.weight_g"], grads[f"{prefix}.lin{l}.weight_v"] = dg, dv
        grads[f"{prefix}.lin{l}.bias"] = db[l]
    return grads, dW, db


def _hbar_from_chain(fw, cfg: SdfCfg, l: int) -> Tensor:
    """hbar_l = d sdf / d h_l (the value the n-chain carried into layer l's input), recomputed from abar_l."""
    rs2 = 1.0 / math.sqrt(2)
    ubar = fw["abar"][l] @ fw["W"][l]
    if l in cfg.skip_in:
        hw = ubar.shape[1] - fw["p"].shape[1]
        return ubar[:, :hw] * rs2
    return ubar


def weight_norm_backward(v: Tensor, g: Tensor, dW: Tensor) -> Tuple[Tensor, Tensor]:
    """A.3 step 3: dg_i = sum_j dW_ij v_ij/||v_i|| ; dv_ij = (g_i/||v_i||)(dW_ij - v_ij sum_k dW_ik v_ik/||v_i||^2)."""
    nrm = v.norm(dim=1, keepdim=True)
    dot = (dW * v).sum(1, keepdim=True)
    dg = dot / nrm
    dv = (g / nrm) * (dW - v * dot / (nrm * nrm))
    return dg, dv


def composite_forward(z_all: Tensor, sdf: Tensor, rgb: Tensor, normals: Optional[Tensor], dnorm: Tensor, beta):
    """A.5 forward on rows: z_all (B,n+1), sdf (B,n), rgb (B,n,3), normals (B,n,3) raw gradients."""
    z, zmax = z_all[:, :-1], z_all[:, -1]
    w, _ = volume_weights(z, zmax, sdf.reshape(-1, 1), beta)
    out = {"w": w, "rgb": (w.unsqueeze(-1) * rgb).sum(1), "depth": (w * z).sum(1) / dnorm.clamp(min=1e-6),
           "wsum": w.sum(-1, keepdim=True)}
    if normals is not None:
        nh = torch.nn.functional.normalize(normals, dim=-1)
        out["nsum"] = (w.unsqueeze(-1) * nh).sum(1)
        out["normal"] = torch.nn.functional.normalize(out["nsum"], dim=-1)
    return out


def composite_backward(z_all, sdf, rgb, dnorm, beta, g_rgb, g_depth, g_wsum):
    """A.5 backward: returns (sdf_bar (B,n), rgb_bar (B,n,3), beta_bar scalar) for upstream grads of
    rgb (B,3), depth (B,), weight_sum (B,1).  (The normal / light composites use w.detach().)"""
    z, zmax = z_all[:, :-1], z_all[:, -1]
    B, n = z.shape
    dens = laplace_density(sdf, beta)
    delta = torch.cat([z[:, 1:] - z[:, :-1], (zmax - z[:, -1]).unsqueeze(-1)], -1)
    E = delta * dens
    T = torch.exp(-torch.cumsum(torch.cat([torch.zeros(B, 1, dtype=z.dtype, device=z.device), E[:, :-1]], -1), -1))
    w = (1 - torch.exp(-E)) * T
    wbar = (g_rgb.unsqueeze(1) * rgb).sum(-1) + g_depth.unsqueeze(-1) * z / dnorm.clamp(min=1e-6).unsqueeze(-1) + g_wsum
    rgb_bar = w.unsqueeze(-1) * g_rgb.unsqueeze(1)
    ww = wbar * w
    suffix = torch.flip(torch.cumsum(torch.flip(ww, [-1]), -1), [-1]) - ww       # sum_{i>k}
    Ebar = wbar * (T - w) - suffix
    sig_bar = delta * Ebar
    ex = torch.exp(-sdf.abs() / beta)
    sdf_bar = sig_bar * (-ex / (2 * beta * beta))
    beta_bar = (sig_bar * (-dens / beta + sdf * ex / (2 * beta ** 3))).sum()
    return sdf_bar, rgb_bar, beta_bar
