"""Configuration of the render core: reads the same yaml keys the reference's I2SDFNetwork ctor reads
(model/network/__init__.py:20-47) and derives the layer shapes (model/network/mlp.py:31-53,176-198)."""
from __future__ import annotations

from dataclasses import dataclass, field
from typing import List, Optional, Tuple


def _get(node, key, default=None):
    if node is None:
        return default
    if isinstance(node, dict):
        return node.get(key, default)
    return getattr(node, key, default)


def _has(node, key):
    if isinstance(node, dict):
        return key in node
    return hasattr(node, key)


@dataclass
class MlpShape:
    """One weight-normed nn.Linear stack."""
    name: str                       # state_dict prefix: implicit_network / rendering_network / light_network
    dims: List[Tuple[int, int]]     # (out, in) per layer
    hidden: int
    d_in: int
    multires: int
    skip_layer: int = -1

    @property
    def n_lin(self):
        return len(self.dims)

    @property
    def pe_dim(self):
        return self.d_in + 2 * self.d_in * self.multires if self.multires > 0 else self.d_in


@dataclass
class SamplerConfig:
    """model/network/ray_sampler.py:47-61"""
    near: float = 0.0
    N_samples: int = 64
    N_samples_eval: int = 128
    N_samples_extra: int = 32
    eps: float = 0.1
    beta_iters: int = 10
    max_total_iters: int = 5
    add_tiny: float = 0.0


@dataclass
class NetConfig:
    feature_size: int
    sdf: MlpShape
    rgb: MlpShape
    light: Optional[MlpShape]
    sampler: SamplerConfig
    scene_bounding_sphere: float = 1.0
    beta_init: float = 0.1
    beta_min: float = 1e-4
    sdf_bias: float = 1.0
    use_normal: bool = False
    detach_light_feature: bool = True
    rgb_mode: str = "nerf"
    # arithmetic of the MFMA kernels that have a bf16x3 twin (csrc/x3.h): fp32 operands split into three bf16 terms, six
    # partial products accumulated in fp32 -- fp32-level accuracy (same parity bar) on the 16x faster bf16 matrix pipe.
    # `bf16x3: false` in the model conf selects the plain fp32-MFMA kernels everywhere.
    bf16x3: bool = True
    # The weight-gradient GEMMs (256x256 blocks) with TWO bf16 terms per operand and three products instead of three and six
    # (include/i2sdf.h: I2SDF_OPT_WGRAD_BF16X2).  Weight gradients are terminal sums over ~1e5 points -- their rounding errors average out
    # and propagate nowhere; measured 3e-6 of the fp64 oracle on every parameter gradient (bar 1e-4), above the reference's own
    # float32_matmul_precision('medium') (main_recon.py:61).  Forward and backward ACTIVATIONS always keep the fp32-equivalent bf16x3
    # form.  `wgrad_bf16x2: false` in the model conf (or I2SDF_WGRAD_BF16X2=0) selects the fp32-equivalent form here too.
    wgrad_bf16x2: bool = True
    # The sampler's sdf-only passes with TWO bf16 terms per operand (include/i2sdf.h: I2SDF_OPT_SAMPLER_BF16X2).  These passes run under
    # no_grad and only CHOOSE the depths (ray_sampler.py:83-95); every returned value is computed at the chosen depths by the fp32-equivalent
    # kernels.  Round 6 ran the WHOLE GPU suite with the option on before making it the default (profiles/r6_sampler_x2.txt: 198 passed --
    # G7's iteration counts exact, the depth bars of tests/test_gpu_sampler.py, G8/G9/G14/G15 end-to-end at 1e-4): sampler entry point
    # 1.09 -> 0.67 ms, step 5.42 -> 4.97 ms, a 640x480 eval render 1.18 -> 0.86 s.  `sampler_bf16x2: false` (or I2SDF_SAMPLER_BF16X2=0)
    # selects the three-plane form here too.
    sampler_bf16x2: bool = True
    # abars, G(hbar), G(a) of the SDF net -- the saved tensors whose consumers keep 16 significant bits of them anyway (the two-plane
    # weight-gradient GEMMs) or use them in the second-order injection only -- stored with 16 significant bits in 3 bytes per value
    # (include/i2sdf.h: I2SDF_OPT_SAVES24; csrc/x3.h P24).  Effective only together with `wgrad_bf16x2` and the point ranges; with
    # `wgrad_bf16x2: false` (the fp32-equivalent mode) these tensors keep fp32 storage.  `saves24: false` (or I2SDF_SAVES24=0) keeps fp32 storage here too.
    saves24: bool = True

    @staticmethod
    def from_conf(conf) -> "NetConfig":
        fvs = int(_get(conf, "feature_vector_size"))
        inet, rnet = _get(conf, "implicit_network"), _get(conf, "rendering_network")
        if _has(conf, "bg_network"):
            raise NotImplementedError("bg_network (inverse-sphere background) is outside the accelerated hot path")
        # --- ImplicitNetwork (mlp.py:31-53)
        d_in, d_out = int(_get(inet, "d_in")), int(_get(inet, "d_out"))
        if _get(inet, "embed_type", None) not in ("positional",):
            raise NotImplementedError("only embed_type='positional' is accelerated (the shipped configs)")
        if not _get(inet, "weight_norm", True) or not _get(rnet, "weight_norm", True):
            raise NotImplementedError("weight_norm=False is not supported")
        multires = int(_get(inet, "multires"))
        hid = list(_get(inet, "dims"))
        skip_in = tuple(_get(inet, "skip_in", ()))
        if len(skip_in) > 1:
            raise NotImplementedError("at most one skip connection")
        pe = d_in + 2 * d_in * multires
        full = [pe] + hid + [d_out + fvs]
        dims = []
        for l in range(len(full) - 1):
            out = full[l + 1] - full[0] if (l + 1) in skip_in else full[l + 1]
            dims.append((out, full[l]))
        if len(set(hid)) != 1:
            raise NotImplementedError("hidden widths must be uniform")
        sdf = MlpShape("implicit_network", dims, hid[0], d_in, multires, skip_in[0] if skip_in else -1)
        # --- RenderingNetwork (mlp.py:176-198)
        mode = _get(rnet, "mode")
        if mode != "nerf":
            raise NotImplementedError("rendering_network.mode must be 'nerf' (the shipped configs)")
        mr_v = int(_get(rnet, "multires", 0)) if _get(rnet, "embed_type", None) else 0
        rhid = list(_get(rnet, "dims"))
        r_in = int(_get(rnet, "d_in")) + fvs + (6 * mr_v if mr_v else 0)
        rfull = [r_in] + rhid + [int(_get(rnet, "d_out"))]
        rgb = MlpShape("rendering_network", [(rfull[l + 1], rfull[l]) for l in range(len(rfull) - 1)], rhid[0], 3, mr_v)
        light = None
        if _has(conf, "light_network"):
            lhid = list(_get(_get(conf, "light_network"), "dims"))
            lfull = [fvs] + lhid + [1]
            light = MlpShape("light_network", [(lfull[l + 1], lfull[l]) for l in range(len(lfull) - 1)], lhid[0], fvs, 0)
        rs = _get(conf, "ray_sampler")
        sam = SamplerConfig(near=float(_get(rs, "near")), N_samples=int(_get(rs, "N_samples")),
                            N_samples_eval=int(_get(rs, "N_samples_eval")), N_samples_extra=int(_get(rs, "N_samples_extra")),
                            eps=float(_get(rs, "eps")), beta_iters=int(_get(rs, "beta_iters")),
                            max_total_iters=int(_get(rs, "max_total_iters")), add_tiny=float(_get(rs, "add_tiny", 0.0)))
        # --- what the kernels are instantiated for (csrc): anything else is refused HERE, with the reason, not at the first launch
        if not ((hid[0] == 256 and fvs == 256 and rhid[0] == 256) or (hid[0] == 64 and fvs == 64 and rhid[0] == 64)):
            raise NotImplementedError(
                f"hidden width {hid[0]} / feature size {fvs} / radiance width {rhid[0]}: the MLP kernels are instantiated for 256/256/256 "
                "(synthetic.yml, synthetic_light_mask.yml: bf16x3 and fp32 MFMA) and 64/64/64 (plumbing size, fp32 MFMA) only -- one wave holds "
                "a whole layer's accumulators in registers (2 x width/32 x 16 of 512), which 512-wide layers do not fit; add an "
                "instantiation in csrc/mlp_*.hip and csrc/plan.cpp for another width")
        if len(set(rhid)) != 1:
            raise NotImplementedError("radiance hidden widths must be uniform")
        if multires != 6 or (mr_v not in (0, 4)):
            raise NotImplementedError(f"positional encodings are instantiated for multires 6 (SDF net) and 4 (view directions); got {multires} / {mr_v}")
        if sam.N_samples_eval > 128 or sam.N_samples > 128 or sam.N_samples_eval * sam.max_total_iters > 640 or sam.max_total_iters > 12 \
                or sam.N_samples + sam.N_samples_extra + 2 > 256:
            raise NotImplementedError(
                f"ray_sampler N_samples_eval={sam.N_samples_eval}, N_samples={sam.N_samples}, N_samples_extra={sam.N_samples_extra}, "
                f"max_total_iters={sam.max_total_iters}: the sampler kernels hold a ray's row in the registers of one wave, compiled for at most "
                "128 new samples per iteration, rows of N_samples_eval * max_total_iters <= 640 depths and N_samples + N_samples_extra + 2 <= 256 "
                "output depths (csrc/sampler.hip: NNEW, NMAX; the shipped configs use 128 / 64 / 32 / 5)")
        dens = _get(conf, "density")
        return NetConfig(feature_size=fvs, sdf=sdf, rgb=rgb, light=light, sampler=sam,
                         scene_bounding_sphere=float(_get(conf, "scene_bounding_sphere", 1.0)),
                         beta_init=float(_get(_get(dens, "params_init"), "beta")), beta_min=float(_get(dens, "beta_min", 1e-4)),
                         sdf_bias=float(_get(inet, "bias", 1.0)), use_normal=bool(_get(conf, "use_normal", False)),
                         detach_light_feature=bool(_get(conf, "detach_light_feature", True)),
                         bf16x3=bool(_get(conf, "bf16x3", True)), wgrad_bf16x2=bool(_get(conf, "wgrad_bf16x2", True)),
                         sampler_bf16x2=bool(_get(conf, "sampler_bf16x2", True)), saves24=bool(_get(conf, "saves24", True)))


def synthetic_conf(light: bool = False) -> dict:
    """The `model:` node of config/synthetic.yml (synthetic_light_mask.yml when light=True) as a plain dict."""
    n_sdf, n_rgb, skip = (6, 3, 3) if light else (8, 4, 4)
    conf = {
        "feature_vector_size": 256, "scene_bounding_sphere": 3.0,
        "implicit_network": {"d_in": 3, "d_out": 1, "dims": [256] * n_sdf, "geometric_init": True, "bias": 0.6, "skip_in": [skip],
                             "weight_norm": True, "embed_type": "positional", "multires": 6},
        "rendering_network": {"mode": "nerf", "d_in": 3, "d_out": 3, "dims": [256] * n_rgb, "weight_norm": True,
                              "embed_type": "positional", "multires": 4},
        "density": {"params_init": {"beta": 0.1}, "beta_min": 0.0001},
        "ray_sampler": {"near": 0.0, "N_samples": 64, "N_samples_eval": 128, "N_samples_extra": 32, "eps": 0.1, "beta_iters": 10,
                        "max_total_iters": 5, "N_samples_inverse_sphere": 32, "add_tiny": 1.0e-6},
    }
    if light:
        conf["light_network"] = {"dims": [128], "weight_norm": True}
    return conf


def plumbing_conf(skip: bool = False, light: bool = False) -> dict:
    """BASELINE.json configs[0]: 2-layer x 64 SDF MLP, 16 samples/ray."""
    conf = synthetic_conf(light)
    conf["feature_vector_size"] = 64
    conf["implicit_network"]["dims"] = [64, 64, 64] if skip else [64, 64]
    conf["implicit_network"]["skip_in"] = [2] if skip else []
    conf["rendering_network"]["dims"] = [64, 64]
    conf["ray_sampler"].update({"N_samples": 16, "N_samples_eval": 32, "N_samples_extra": 8})
    if light:
        conf["light_network"] = {"dims": [32], "weight_norm": True}
    return conf
