"""Drop-in replacement of the reference's `model.I2SDFNetwork` (model/network/__init__.py:19-286) for the shipped
configurations: same constructor argument (the yaml `model:` node), same call form `net(input, predict_only=False)`,
same output dict, same sub-module surface (`implicit_network(x)`, `density.beta`, `get_param_groups`,
`rendering_network.mode`) and the same `state_dict` keys (`implicit_network.lin3.weight_v`, ...), so the reference's
Lightning trainer and its checkpoints work unchanged.  The arithmetic runs in the HIP library (i2sdf_amd/csrc) through
the C ABI of include/i2sdf.h; there is no CPU / eager-torch fallback for it.
"""
from __future__ import annotations

import os

from typing import Dict, Optional

import torch
import torch.nn as nn
import torch.nn.functional as F

from .config import MlpShape, NetConfig
from .engine import RenderEngine
from .params import ParamLayout


class _WNLinear(nn.Module):
    """Parameter container with the names torch.nn.utils.weight_norm(nn.Linear) produces (mlp.py:53,71-74)."""

    def __init__(self, out_dim: int, in_dim: int):
        super().__init__()
        self.bias = nn.Parameter(torch.zeros(out_dim))
        self.weight_g = nn.Parameter(torch.ones(out_dim, 1))
        self.weight_v = nn.Parameter(torch.zeros(out_dim, in_dim))


class _Mlp(nn.Module):
    def __init__(self, shape: MlpShape):
        super().__init__()
        self.shape = shape
        self.num_layers = shape.n_lin + 1
        for l, (out, inn) in enumerate(shape.dims):
            setattr(self, f"lin{l}", _WNLinear(out, inn))

    def get_param_groups(self, lr):
        return [{"params": self.parameters(), "lr": lr}]


class ImplicitNetwork(_Mlp):
    """model/network/mlp.py:10-151.  Calls are forwarded to the owning I2SDFNetwork's engine."""

    def __init__(self, shape: MlpShape, owner: "I2SDFNetwork"):
        super().__init__(shape)
        object.__setattr__(self, "_owner", owner)
        self.skip_in = (shape.skip_layer,) if shape.skip_layer >= 0 else ()

    def forward(self, x):
        """(M,3) -> (M, 1 + feature): [sdf | feature], no autograd graph (marching cubes / grid queries)."""
        eng = self._owner._engine_for(x.device)
        sdf, feat = eng.sdf_forward(x, want_features=True)
        return torch.cat([sdf, feat], dim=1)

    def get_sdf_vals(self, x):
        return self._owner._engine_for(x.device).sdf_forward(x)

    def gradient(self, x):
        """d sdf / d x without a graph.  (With a graph it is part of I2SDFNetwork.forward.)"""
        eng = self._owner._engine_for(x.device)
        return eng.sdf_forward_grad(points=x, save=False, want_feat=False)["grad"]

    def get_outputs(self, x, returns_grad=True):
        eng = self._owner._engine_for(x.device)
        o = eng.sdf_forward_grad(points=x, want_grad=returns_grad, save=False)
        return o["sdf"], o["feat"][: x.shape[0]], o["grad"]


class RenderingNetwork(_Mlp):
    def __init__(self, shape: MlpShape, mode: str):
        super().__init__(shape)
        self.mode = mode


class LaplaceDensity(nn.Module):
    """model/network/density.py:16-30 (parameter container + the closed form for callers that use it directly)."""

    def __init__(self, beta_init: float, beta_min: float):
        super().__init__()
        self.beta = nn.Parameter(torch.tensor(float(beta_init)))
        self.beta_min = torch.tensor(float(beta_min))

    def get_beta(self):
        return self.beta.abs() + self.beta_min.to(self.beta.device)

    def forward(self, sdf, beta=None):
        if beta is None:
            beta = self.get_beta()
        return (1 / beta) * (0.5 + 0.5 * sdf.sign() * torch.expm1(-sdf.abs() / beta))


class _RenderFn(torch.autograd.Function):
    """The whole differentiable core as one autograd node: HIP forward, HIP backward."""

    @staticmethod
    def forward(ctx, net, st, *params):
        eng: RenderEngine = st["eng"]
        flat = net._flat
        B, n = st["z_all"].shape[0], st["z_all"].shape[1] - 1
        M_main = B * n
        n_extra = 0 if st["extra_pts"] is None else st["extra_pts"].shape[0]
        # one chain: with point ranges on (engine.parts) every range runs SDF forward -> d sdf/dx -> radiance net on its own stream
        with eng.chain(M_main + n_extra):
            fw = eng.sdf_forward_grad(points=st["extra_pts"], rays=(st["cam"], st["dirs"], st["z_all"], n), want_grad=True, save=True)
            rgb, rs, pev = eng.rgb_forward(st["dirs"], n, fw["feat"], M_main, save=True)
        lm = hl = None
        if net.use_light:
            lm, hl = eng.light_forward(fw["feat"], M_main, save=True)
        beta_param = flat[eng.layout.offset("density.beta"):]
        n_eik0 = st["n_eik"]
        comp = eng.composite_forward(beta_param, st["z_all"], fw["sdf"], rgb, fw["grad"], lm, st["dnorm"], want_normal=st["want_normal"],
                                     eik_grad=fw["grad"][M_main:M_main + n_eik0] if n_eik0 == 3 * B else None)
        # (the eikonal / smoothness outputs came out of the same launch: _EikonalOutputsFn.forward returns them instead of launching)
        st["eik_out"] = (comp.pop("grad_theta"), comp.pop("diff_norm")) if "grad_theta" in comp else None
        ctx.net, ctx.st, ctx.fw, ctx.rgb, ctx.rs, ctx.pev, ctx.lm, ctx.hl, ctx.comp = net, st, fw, rgb, rs, pev, lm, hl, comp
        ctx.M_main = M_main
        # for I2SDFLoss's fused path (loss.py: _FusedRenderLossFn -> i2sdf_render_loss_backward): the compositing inputs of this render.
        # Cleared in backward together with ctx; no entry refers to an output tensor (the handle is an attribute of one: no cycle)
        st["fused"] = {"fw": fw, "rgb_pts": rgb, "lm_pts": lm, "nsum": comp["nsum"], "beta_param": beta_param, "n": n, "M_main": M_main,
                       "use_light": net.use_light, "beta_min": eng.cfg.beta_min}
        # outputs the loss does not use (weight_sum without a mask term, ...) arrive in backward as None instead of as zero tensors that
        # autograd would fill with one launch each; the kernels take NULL for them (include/i2sdf.h)
        ctx.set_materialize_grads(False)
        outs = [comp["rgb"], comp["depth"], comp["wsum"]]
        outs.append(comp["normal"] if st["want_normal"] else torch.zeros(0, device=rgb.device))
        outs.append(comp["lmask"] if net.use_light else torch.zeros(0, device=rgb.device))
        n_eik = st["n_eik"]
        outs.append(fw["grad"][M_main:M_main + n_eik] if n_eik else torch.zeros(0, 3, device=rgb.device))
        n_pc = st["n_pc"]
        outs.append(fw["sdf"][M_main + n_eik:M_main + n_eik + n_pc] if n_pc else torch.zeros(0, 1, device=rgb.device))
        return tuple(outs)

    @staticmethod
    def backward(ctx, g_rgb, g_depth, g_wsum, g_normal, g_lmask, g_eik, g_surf):
        net, st, fw, comp, M_main = ctx.net, ctx.st, ctx.fw, ctx.comp, ctx.M_main
        eng: RenderEngine = st["eng"]
        flat = net._flat
        dev = flat.device
        M_sdf = fw["M"]
        off_beta = eng.layout.offset("density.beta")
        want_normal = st["want_normal"]
        n_eik, n_pc = st["n_eik"], st["n_pc"]
        # No whole-buffer fills: i2sdf_weight_grads writes every entry of the flat gradient except density.beta (accumulated by the
        # compositing backward), and the compositing backward writes the rays' rows [0, M_main) of sbar / nbar; only the rows of the
        # extra points are set here (eikonal points carry d loss / d grad, the bubble point cloud d loss / d sdf).
        # One launch (i2sdf_backward_seeds) instead of three fills and two copies.
        gflat = torch.empty_like(flat)
        from . import lib as L_
        pre = st.get("pre")
        same = lambda a, b: a is not None and b is not None and a.data_ptr() == b.data_ptr() and a.numel() == b.numel()
        if pre is not None and pre["g"] is not None:
            tk = pre["tok"]
            fast = (same(g_rgb, tk["rgb"]) and same(g_depth, tk["depth"]) and (g_wsum is None or same(g_wsum, tk["wsum"]))
                    and ((not st["want_normal"]) or tk["normal"] is None or same(g_normal, tk["normal"]))
                    and ((not net.use_light) or same(g_lmask, tk["lmask"]))
                    and ((not n_eik) or same(g_eik, pre["tok_eik"]))
                    and ((not n_pc) or tk["surface"] is None or same(g_surf, tk["surface"])))
            if fast:
                # I2SDFLoss already ran the loss AND the backward down to the per-sample gradients, for an upstream gradient of 1
                # (i2sdf_render_loss_backward); what autograd delivered are the loss's own untouched seeds: scale by its gradient, done
                return _RenderFn._backward_from_seeds(ctx, pre, gflat)
            # the loss's seeds were mixed with other gradients of the same outputs (another loss term on them): they were handed to
            # autograd UNSCALED -- correct what arrived to  g * seed + the rest,  then take the general path
            gs_ = pre["g"].reshape(()).to(torch.float32) - 1.0
            fix = lambda gx, t: gx if (gx is None or t is None) else gx + gs_ * t.reshape(gx.shape)
            g_rgb, g_depth, g_wsum = fix(g_rgb, tk["rgb"]), fix(g_depth, tk["depth"]), fix(g_wsum, tk["wsum"])
            g_normal, g_lmask, g_surf = fix(g_normal, tk["normal"]), fix(g_lmask, tk["lmask"]), fix(g_surf, tk["surface"])
            if n_eik and g_eik is not None and not same(g_eik, pre["tok_eik"]):
                pass      # (_EikonalOutputsFn.backward applied the same correction to grad_theta / diff_norm before its own backward)
            elif n_eik and same(g_eik, pre["tok_eik"]):
                g_eik = pre["eik_true"]()          # the placeholder reached us: recompute the extra points' gradient the general way
        sbar = torch.empty(M_sdf, device=dev)
        nbar = torch.empty(M_sdf, 3, device=dev)
        if g_rgb is None:                      # a loss without a colour term: the compositing backward wants the pointer
            g_rgb = torch.zeros_like(comp["rgb"])
        if net.use_light and g_lmask is None:
            g_lmask = torch.zeros_like(comp["lmask"])
        ge = g_eik.contiguous() if (n_eik and g_eik is not None) else None
        gs = g_surf.reshape(-1).contiguous() if (n_pc and g_surf is not None) else None
        with torch.cuda.device(dev):
            L_.check(L_.load().i2sdf_backward_seeds(L_.ptr(gflat[off_beta:]), flat.numel() - off_beta, L_.ptr(sbar), L_.ptr(nbar), M_main, M_sdf,
                                                   L_.ptr(ge) if ge is not None else None, n_eik, L_.ptr(gs) if gs is not None else None, n_pc,
                                                   0 if want_normal else 1, L_.stream_ptr()), "i2sdf_backward_seeds")
        cb = eng.composite_backward(flat[off_beta:], st["z_all"], fw["sdf"], ctx.rgb, fw["grad"], st["dnorm"], comp["nsum"],
                                    g_rgb, g_depth, None if g_wsum is None else g_wsum.reshape(-1), g_normal if want_normal else None,
                                    g_lmask.reshape(-1) if net.use_light else None, beta_grad_accum=gflat[off_beta:],
                                    sdf_bar_out=sbar, grad_bar_out=nbar if want_normal else None)
        light = None
        if net.use_light:                      # (before the chain: the head's kernels run on the caller's stream only)
            gal0, gal_last = eng.light_backward(ctx.lm, cb["lmask_bar"], ctx.hl, M_main)
            light = {"hl": ctx.hl, "gal0": gal0, "gal_last": gal_last}
        # one chain: radiance backward -> SDF sweeps -> weight-gradient GEMMs per point range (i2sdf_weight_grads joins the ranges)
        with eng.chain(M_sdf):
            gar, ga_last, fbar = eng.rgb_backward(ctx.rgb, cb["rgb_bar"], ctx.rs, M_main)
            bw = eng.sdf_backward(fw, sbar=sbar, fbar=fbar, m_fbar=M_main, nbar=nbar)
            eng.weight_grads(flat, gflat, fw, bw, M_main=M_main, fbar=fbar, rgb_fw={"pev": ctx.pev, "rs": ctx.rs},
                             rgb_bw={"gar": gar, "ga_last": ga_last}, light=light)
        return _RenderFn._finish(ctx, gflat)

    @staticmethod
    def _finish(ctx, gflat):
        net, st = ctx.net, ctx.st
        eng = st["eng"]
        if net.grad_sync is not None:
            net.grad_sync(gflat)
        grads = []
        for name, off, shape in eng.layout.entries:
            cnt = 1
            for s in shape:
                cnt *= s
            grads.append(gflat[off:off + cnt].view(shape))
        st.pop("fused", None)
        st.pop("pre", None)
        ctx.net = ctx.st = ctx.fw = ctx.comp = None
        return (None, None) + tuple(grads)

    @staticmethod
    def _backward_from_seeds(ctx, pre, gflat):
        """backward when I2SDFLoss's fused path has already produced the per-sample gradients (for an upstream gradient of 1): one launch
        scales them by the gradient autograd delivered, then radiance backward -> SDF sweeps -> weight gradients as in the general path"""
        from . import lib as L_
        net, st, fw, M_main = ctx.net, ctx.st, ctx.fw, ctx.M_main
        eng: RenderEngine = st["eng"]
        flat = net._flat
        off_beta = eng.layout.offset("density.beta")
        sbar, nbar, rgb_bar, lmask_bar = pre["sbar"], pre["nbar"], pre["rgb_bar"], pre["lmask_bar"]
        g = pre["g"].reshape(1).to(torch.float32).contiguous()
        with torch.cuda.device(flat.device):
            L_.check(L_.load().i2sdf_scale_seeds(L_.ptr(g), L_.ptr(sbar), sbar.numel(), L_.ptr(nbar), nbar.numel(), L_.ptr(rgb_bar), rgb_bar.numel(),
                                                 L_.ptr(lmask_bar), 0 if lmask_bar is None else lmask_bar.numel(), L_.ptr(pre["beta_g"]),
                                                 L_.ptr(gflat[off_beta:]), L_.stream_ptr()), "i2sdf_scale_seeds")
        light = None
        if net.use_light:
            gal0, gal_last = eng.light_backward(ctx.lm, lmask_bar, ctx.hl, M_main)
            light = {"hl": ctx.hl, "gal0": gal0, "gal_last": gal_last}
        with eng.chain(fw["M"]):
            gar, ga_last, fbar = eng.rgb_backward(ctx.rgb, rgb_bar, ctx.rs, M_main)
            bw = eng.sdf_backward(fw, sbar=sbar, fbar=fbar, m_fbar=M_main, nbar=nbar)
            eng.weight_grads(flat, gflat, fw, bw, M_main=M_main, fbar=fbar, rgb_fw={"pev": ctx.pev, "rs": ctx.rs},
                             rgb_bw={"gar": gar, "ga_last": ga_last}, light=light)
        return _RenderFn._finish(ctx, gflat)


class _EikonalOutputsFn(torch.autograd.Function):
    """grad_theta / diff_norm of the extra points (model/network/__init__.py:188-193): one HIP launch forward, one backward
    (i2sdf_eikonal_outputs_*), instead of ~8 + ~25 element-wise torch kernels on (2B,3) tensors."""

    @staticmethod
    def forward(ctx, g_all, B, st=None):
        from . import lib as L_
        g_all = g_all.contiguous()
        ctx.st = st
        pre_out = st.pop("eik_out", None) if st is not None else None
        if pre_out is not None:
            theta, diff = pre_out            # computed by the compositing launch of _RenderFn.forward (i2sdf_composite_forward_eik)
        else:
            theta = torch.empty(2 * B, 3, device=g_all.device)
            diff = torch.empty(B, device=g_all.device)
            with torch.cuda.device(g_all.device):
                L_.check(L_.load().i2sdf_eikonal_outputs_forward(L_.ptr(g_all), B, L_.ptr(theta), L_.ptr(diff), L_.stream_ptr()),
                         "i2sdf_eikonal_outputs_forward")
        ctx.save_for_backward(g_all)
        ctx.B = B
        ctx.set_materialize_grads(False)
        return theta, diff

    @staticmethod
    def backward(ctx, g_theta, g_diff):
        from . import lib as L_
        (g_all,) = ctx.saved_tensors
        if g_theta is None and g_diff is None:
            return None, None, None
        pre = ctx.st.get("pre") if ctx.st is not None else None
        ctx.st = None
        if pre is not None and pre["g"] is not None:
            tk = pre["tok"]
            same = lambda a, b: a is not None and b is not None and a.data_ptr() == b.data_ptr() and a.numel() == b.numel()
            if same(g_theta, tk["grad_theta"]) and (g_diff is None or same(g_diff, tk["diff_norm"])):
                # the fused loss's own seeds, untouched: the extra points' gradient rows were already written by i2sdf_render_loss_backward;
                # hand the placeholder on, so that _RenderFn.backward can tell that nobody else contributed.  `eik_true` computes the real
                # thing if the placeholder should meet other gradients further down after all.
                g_th, g_df, ga = g_theta, g_diff, g_all
                pre["eik_true"] = lambda: _EikonalOutputsFn._bwd(ga, g_th * pre["g"].reshape(()), None if g_df is None else g_df * pre["g"].reshape(()), ctx.B)
                return pre["tok_eik"], None, None
            gs_ = pre["g"].reshape(()).to(torch.float32) - 1.0          # unscaled seeds mixed with other gradients: g * seed + the rest
            if g_theta is not None and tk["grad_theta"] is not None:
                g_theta = g_theta + gs_ * tk["grad_theta"]
            if g_diff is not None and tk["diff_norm"] is not None:
                g_diff = g_diff + gs_ * tk["diff_norm"]
        return _EikonalOutputsFn._bwd(g_all, g_theta, g_diff, ctx.B), None, None

    @staticmethod
    def _bwd(g_all, g_theta, g_diff, B):
        from . import lib as L_
        c = lambda t: None if t is None else t.contiguous()
        g_theta, g_diff = c(g_theta), c(g_diff)
        out = torch.empty_like(g_all)
        with torch.cuda.device(g_all.device):
            L_.check(L_.load().i2sdf_eikonal_outputs_backward(L_.ptr(g_all), L_.ptr(g_theta), L_.ptr(g_diff), B, L_.ptr(out),
                                                              L_.stream_ptr()), "i2sdf_eikonal_outputs_backward")
        return out


class I2SDFNetwork(nn.Module):
    def __init__(self, conf):
        super().__init__()
        cfg = conf if isinstance(conf, NetConfig) else NetConfig.from_conf(conf)
        self.cfg = cfg
        self.feature_vector_size = cfg.feature_size
        self.scene_bounding_sphere = cfg.scene_bounding_sphere
        self.implicit_network = ImplicitNetwork(cfg.sdf, self)
        self.rendering_network = RenderingNetwork(cfg.rgb, cfg.rgb_mode)
        self.use_light = cfg.light is not None
        if self.use_light:
            self.light_network = _Mlp(cfg.light)
        self.density = LaplaceDensity(cfg.beta_init, cfg.beta_min)
        self.use_bg = False
        self.use_normal = cfg.use_normal
        self.detach_light_feature = cfg.detach_light_feature
        if not self.detach_light_feature:
            raise NotImplementedError("detach_light_feature=False is not supported by the fused light head")
        self.layout = ParamLayout(cfg)
        self.force_iters = 0            # >0: fixed sampler iteration count (benchmarks); 0: the reference's data-dependent loop
        # all random draws of a training forward in one library launch (False / I2SDF_FUSED_DRAWS=0: separate torch ops)
        self.fused_draws = os.environ.get("I2SDF_FUSED_DRAWS", "1") != "0"
        self.grad_sync = None           # callable(flat_grad) for data-parallel training (i2sdf_amd.dist)
        self.dp_state = None            # i2sdf_amd.dist.DataParallelState once attach_data_parallel() was called
        self.last_sampler_iters = None  # device int32 tensor of the last forward
        self.last_extra_idx = None
        object.__setattr__(self, "_engines", {})
        object.__setattr__(self, "_flat", None)
        object.__setattr__(self, "_packed", set())
        self._init_parameters()

    # ------------------------------------------------------------------------------------------
    def _param_list(self):
        sd = dict(self.named_parameters())
        return [sd[name] for name, _, _ in self.layout.entries]

    def _init_parameters(self, generator=None):
        flat = self.layout.init_flat(generator)
        with torch.no_grad():
            for (name, off, shape), p in zip(self.layout.entries, self._param_list()):
                p.copy_(flat[off:off + p.numel()].view(shape))

    def get_param_groups(self, lr):
        return [{"params": self.parameters(), "lr": lr}]

    def _ensure_flat(self):
        """Keep every parameter a view into one flat fp32 buffer (the layout the C ABI and the gradient all-reduce use)."""
        params = self._param_list()
        dev = params[0].device
        flat = self._flat
        ok = flat is not None and flat.device == dev
        if ok:
            base = flat.data_ptr()
            for (name, off, shape), p in zip(self.layout.entries, params):
                if p.data_ptr() != base + 4 * off or p.dtype != torch.float32:
                    ok = False
                    break
        if not ok:
            # Lightning's validate/test loops run under torch.inference_mode(): a buffer created there would be an inference
            # tensor (no version counter, not usable by autograd afterwards).  The flat buffer must be an ordinary tensor.
            with torch.inference_mode(False), torch.no_grad():
                flat = torch.empty(self.layout.n_params, dtype=torch.float32, device=dev)
                for (name, off, shape), p in zip(self.layout.entries, params):
                    flat[off:off + p.numel()].copy_(p.detach().reshape(-1).to(torch.float32))
                    p.data = flat[off:off + p.numel()].view(shape)
            object.__setattr__(self, "_flat", flat)
            self._packed.clear()
        return self._flat

    def _engine_for(self, device, fresh: bool = True) -> RenderEngine:
        device = torch.device(device)
        if device.type != "cuda":
            raise RuntimeError("i2sdf_amd.I2SDFNetwork runs on MI355X only (tensors must be on a cuda/HIP device); "
                               "there is no CPU fallback -- use oracle/ for CPU reference numbers")
        flat = self._ensure_flat()
        if flat.device != device:
            raise RuntimeError(f"module parameters are on {flat.device}, input on {device}")
        key = str(device)
        eng = self._engines.get(key)
        if eng is None:
            with torch.inference_mode(False), torch.cuda.device(device):
                eng = RenderEngine(self.cfg, device)
            self._engines[key] = eng
        # The parameters are views of `flat` made through `.data`, so they keep their OWN version counters, and fused /
        # foreach optimizers or raw-pointer writers need not bump any of them: no version key can tell whether the packed
        # weight streams are stale.  Packing costs ~17 us, so every public entry point (fresh=True) simply repacks; only
        # the internal second look-up of the same call (render() after forward()) reuses the streams.
        if fresh or key not in self._packed:
            with torch.cuda.device(device):
                eng.pack(flat)
            self._packed.add(key)
        return eng

    # ------------------------------------------------------------------------------------------
    def forward(self, input: Dict[str, torch.Tensor], predict_only: bool = False, draws: Optional[dict] = None):
        uv = input["uv"]
        eng = self._engine_for(uv.device)
        flat = self._flat
        sc = self.cfg.sampler
        rays = input.get("rays")
        if rays is not None:                # batch from i2sdf_amd.batcher.RayBatcher: rays were made together with the ground truth
            cam, dirs, dnorm = rays["cam_loc"], rays["dirs"], rays["dnorm"]
        else:
            with torch.cuda.device(uv.device), torch.no_grad():
                cam, dirs, dnorm = eng.ray_setup(uv, input["pose"], input["intrinsics"])
        N = cam.shape[0]
        dev = cam.device
        training = self.training
        draws = draws or {}
        user_extra = "extra_idx" in draws
        with torch.cuda.device(dev), torch.no_grad():
            if training:
                if not draws and self.fused_draws:
                    # production path: one launch for all draws of this forward, keyed by a 62-bit seed from torch's CPU generator
                    # (so torch.manual_seed / pl.seed_everything still decide the run); explicit `draws` (tests) bypass it
                    seed = int(torch.randint(0, 2 ** 62, (1,), dtype=torch.int64).item())
                    draws = eng.training_draws(N, seed, dev, self.scene_bounding_sphere, want_eik=not predict_only)
                    draws = {k: v for k, v in draws.items() if v is not None}
                strat_u = draws.get("strat_u")
                if strat_u is None:
                    strat_u = torch.rand(N, sc.N_samples_eval, device=dev)
                cdf_u = draws.get("cdf_u")
                if cdf_u is None:
                    cdf_u = torch.rand(N, sc.N_samples, device=dev)
                extra = draws.get("extra_idx")
                if extra is None and sc.N_samples_extra > 0:
                    # randperm(n)[:k] for every possible row length n = N_eval*(it+1) at once: the k smallest of n i.i.d.
                    # uniform keys are a uniformly random k-subset in random order (2 kernels instead of 5 sorts)
                    keys = torch.rand(sc.max_total_iters, sc.N_samples_eval * sc.max_total_iters, device=dev)
                    keys = keys + self._extra_mask(dev)
                    extra = keys.topk(sc.N_samples_extra, dim=1, largest=False).indices
                elif extra is not None and extra.dim() == 1:
                    extra = extra.repeat(sc.max_total_iters, 1)
                dp = self.dp_state
                if dp is not None and dp.equivalent:
                    # 1-GPU-equivalent data parallelism: the reference draws ONE randperm for all rays of a batch (ray_sampler.py:223),
                    # so every rank must use rank 0's columns; and the convergence test is the OR over all ranks (device-side hook)
                    if not user_extra and extra is not None:
                        extra = extra.contiguous()
                        if dp.comm is not None:
                            dp.comm.broadcast(extra, 0)
                        else:
                            import torch.distributed as tdist
                            tdist.broadcast(extra, src=tdist.get_global_rank(dp.group, 0) if dp.group is not None else 0, group=dp.group)
                    if getattr(eng, "_exchange", None) is None:
                        from . import lib as L_
                        eng.set_exchange(dp.xchg.exchange(), L_.DP_GLOBAL_SAMPLER)
                eik_idx = draws.get("eik_idx")
                if eik_idx is None:
                    eik_idx = torch.randint(eng.n_z, (N,), device=dev)
                self.last_extra_idx = extra          # (max_total_iters, N_extra) columns this forward used (shared by all rays / ranks)
                z_all, z_eik, iters = eng.sample_rays(flat, cam, dirs, training=True, strat_u=strat_u, cdf_u=cdf_u, extra_idx=extra,
                                                      eik_idx=eik_idx, force_iters=self.force_iters)
            else:
                z_all, z_eik, iters = eng.sample_rays(flat, cam, dirs, training=False, force_iters=self.force_iters)
        self.last_sampler_iters = iters
        return self.render(input, cam, dirs, dnorm, z_all, z_eik, predict_only, draws, _fresh=False)

    def _extra_mask(self, dev):
        m = getattr(self, "_extra_mask_t", None)
        if m is None or m.device != dev:
            sc = self.cfg.sampler
            col = torch.arange(sc.N_samples_eval * sc.max_total_iters, device=dev).unsqueeze(0)
            lim = (torch.arange(sc.max_total_iters, device=dev).unsqueeze(1) + 1) * sc.N_samples_eval
            m = (col >= lim).to(torch.float32) * 2.0          # keys of columns beyond the row length are pushed past 1
            object.__setattr__(self, "_extra_mask_t", m)
        return m

    def render(self, input, cam, dirs, dnorm, z_all, z_eik, predict_only=False, draws=None, _fresh=True):
        """Everything after the sampler (model/network/__init__.py:99-221) for given depths z_all (N, n+1)."""
        eng = self._engine_for(cam.device, fresh=_fresh)
        flat = self._flat
        dev = cam.device
        N, n = z_all.shape[0], z_all.shape[1] - 1
        training = self.training
        draws = draws or {}
        need_graph = torch.is_grad_enabled() and any(p.requires_grad for p in self.parameters())
        with_eik = training and not predict_only
        with torch.cuda.device(dev):
            if not need_graph:
                return self._render_nograd(eng, flat, input, cam, dirs, dnorm, z_all, z_eik, predict_only, draws, with_eik)
            extra_pts, n_eik, n_pc = None, 0, 0
            if with_eik:
                extra_pts, n_eik, n_pc = self._extra_points(input, cam, dirs, z_eik, draws)
            want_normal = training and self.use_normal and not predict_only
            st = {"eng": eng, "cam": cam, "dirs": dirs, "dnorm": dnorm, "z_all": z_all.contiguous(), "extra_pts": extra_pts, "n_eik": n_eik,
                  "n_pc": n_pc, "want_normal": want_normal or (not training)}
            rgb, depth, wsum, normal, lmask, g_eik, surf = _RenderFn.apply(self, st, *self._param_list())
            out = {"rgb_values": rgb, "depth_values": depth, "weight_sum": wsum}
            if self.use_light:
                out["light_mask"] = lmask
            if predict_only:
                return out
            if training:
                self._eikonal_outputs(out, g_eik, surf, N, n_pc, st)
                if self.use_normal:
                    out["normal_values"] = normal
                # I2SDFLoss looks for this handle: with it, loss + render backward run as one fused library call (loss.py).  The pointers
                # identify the outputs (a clone / detach / slice of one is a different tensor: the loss then takes its general path)
                st["out_ptrs"] = {k: v.data_ptr() for k, v in out.items()}
                rgb._i2sdf_render = st
            else:
                out["normal_map"] = normal.detach()
            return out

    # ------------------------------------------------------------------------------------------
    # "next" rows N3 / N4 of SURVEY.md section 8(f): the callers' chunk loops, kept on the device
    @torch.no_grad()
    def render_image(self, input: Dict[str, torch.Tensor], split_n_pixels: int = 12000, rank: int = 0, world_size: int = 1,
                     return_depths: bool = False) -> Dict[str, torch.Tensor]:
        """Full-image inference: utils.split_input -> self(chunk) -> utils.merge_output (utils/__init__.py:35-84,
        model/eval/recon.py:161-182) as ONE library call (i2sdf_render_image): every chunk of `split_n_pixels` rays is rendered
        exactly as the reference renders it (eval mode, the sampler's convergence test is per chunk) and written straight into
        the (P, C) outputs in pixel order.  input['uv'] is (1, P, 2) -- one view.

        world_size > 1: this rank renders only ITS chunks (whole chunks, contiguous: the chunk composition -- and with it every
        pixel -- is the same as on one GPU) and returns its rows; i2sdf_amd.dist.render_image gathers the image."""
        uv = input["uv"]
        assert uv.dim() == 3 and uv.shape[0] == 1, "eval layout: uv is (1, P, 2), one view"
        P = uv.shape[1]
        eng = self._engine_for(uv.device)
        n_chunks = (P + split_n_pixels - 1) // split_n_pixels
        per = (n_chunks + world_size - 1) // world_size
        lo = min(rank * per, n_chunks) * split_n_pixels
        hi = min(min((rank + 1) * per, n_chunks) * split_n_pixels, P)
        with torch.cuda.device(uv.device):
            o = eng.render_image(self._flat, uv[0, lo:hi], input["pose"][0], input["intrinsics"][0], split_n_pixels, want_z=return_depths)
        self.last_sampler_iters = o["iters"]                  # one count per chunk
        out = {"rgb_values": o["rgb"], "depth_values": o["depth"], "weight_sum": o["wsum"]}
        if self.use_light:
            out["light_mask"] = o["lmask"]
        out["normal_map"] = o["normal"]
        if return_depths:
            out["z_vals"] = o["z"]
        return out

    @torch.no_grad()
    def sdf_grid(self, points: torch.Tensor, chunk: int = 1 << 20) -> torch.Tensor:
        """SDF values of an arbitrary point set in chunks (marching-cubes grids: model/eval/recon.py:46-51,89-90,
        utils/plots.py:440-489).  Returns (M,)."""
        out = torch.empty(points.shape[0], dtype=torch.float32, device=points.device)
        eng = self._engine_for(points.device)
        for lo in range(0, points.shape[0], chunk):
            out[lo:lo + chunk] = eng.sdf_forward(points[lo:lo + chunk])[:, 0]
        return out

    @torch.no_grad()
    def sdf_volume(self, axes, rot=None, trans=None, order: str = "volume", chunk: int = 1 << 21, rank: int = 0, world_size: int = 1,
                   device=None) -> torch.Tensor:
        """SDF values on a grid given by its axis vectors (i2sdf_amd.grid.uniform_axes / aligned_axes, or any (x, y, z)) in ONE
        library call (i2sdf_sdf_grid): the points are generated on the device chunk by chunk, so neither the (n,3) point tensor
        (1.6 GB at 512^3) nor the 32-worker GridDataset loader of model/eval/recon.py:96-103 exists.
          order="volume"  : returns (nx, ny, nz) -- exactly the array the reference hands to measure.marching_cubes
                            (z.reshape(ny,nx,nz).transpose([1,0,2]), model/eval/recon.py:53-54,94)
          order="meshgrid": returns the flat (ny*nx*nz,) vector in np.meshgrid(x,y,z).ravel() order (the reference's `z`)
          rot / trans     : evaluated point = rot @ p + trans (the aligned grid passes rot = vecs.T, trans = s_mean, :82-85)
          world_size > 1  : this rank evaluates its contiguous slab of the flat output and returns it flat (concatenate in rank order)."""
        x, y, z = (axes.x, axes.y, axes.z) if hasattr(axes, "shortest_axis_index") else axes
        dev = torch.device(device) if device is not None else self.density.beta.device
        up = lambda a: torch.as_tensor(a).to(dev, torch.float32)
        x, y, z = up(x), up(y), up(z)
        eng = self._engine_for(dev)
        total = x.numel() * y.numel() * z.numel()
        per = (total + world_size - 1) // world_size
        lo, hi = min(rank * per, total), min((rank + 1) * per, total)
        from . import lib as L_
        o = {"volume": L_.GRID_ORDER_VOLUME, "meshgrid": L_.GRID_ORDER_MESHGRID}[order]
        with torch.cuda.device(dev):
            out = eng.sdf_grid(x, y, z, rot, trans, o, lo, hi - lo, chunk)
        if world_size == 1 and order == "volume":
            return out.view(x.numel(), y.numel(), z.numel())
        return out

    # ------------------------------------------------------------------------------------------
    def _extra_points(self, input, cam, dirs, z_eik, draws):
        """Eikonal / neighbour / bubble points (model/network/__init__.py:175-201)."""
        N, dev, R = cam.shape[0], cam.device, self.scene_bounding_sphere
        with torch.no_grad():
            eik = draws.get("eik_pts")
            if eik is None:
                eik = torch.empty(N, 3, device=dev).uniform_(-R, R)
            off = draws.get("nbr_off")
            if off is None:
                off = torch.empty(N, 3, device=dev).uniform_(-0.005, 0.005)
            pc = input["pointcloud"].to(torch.float32) if "pointcloud" in input else None
            n_pc = 0 if pc is None else pc.shape[0]
            # [uniform | cam + z_eik * dirs | that + offset] by one launch (i2sdf_extra_points) instead of a multiply, two adds and a cat
            pts = torch.empty(3 * N + n_pc, 3, device=dev)
            from . import lib as L_
            c = lambda t: t.to(torch.float32).contiguous()
            # raw pointers go to the library from here: a tensor of the wrong size (e.g. draws made for another N through the public
            # render()) would be an out-of-bounds device read, where the torch expression this replaced raised
            for name, t_, numel in (("cam_loc", cam, 3 * N), ("ray dirs", dirs, 3 * N), ("z_samples_eik", z_eik, N), ("draws['eik_pts']", eik, 3 * N),
                                    ("draws['nbr_off']", off, 3 * N)):
                if t_.numel() != numel or t_.device != dev:
                    raise ValueError(f"{name}: {tuple(t_.shape)} on {t_.device}, expected {numel} elements on {dev} (N = {N} rays)")
            with torch.cuda.device(dev):
                L_.check(L_.load().i2sdf_extra_points(L_.ptr(c(cam)), L_.ptr(c(dirs)), L_.ptr(c(z_eik)), L_.ptr(c(eik)), L_.ptr(c(off)), N,
                                                     L_.ptr(pts), L_.stream_ptr()), "i2sdf_extra_points")
            if n_pc:
                pts[3 * N:] = pc
            return pts, 3 * N, n_pc

    @staticmethod
    def _eikonal_outputs(out, g_all, surf, N, n_pc, st=None):
        out["grad_theta"], out["diff_norm"] = _EikonalOutputsFn.apply(g_all, N, st)
        if n_pc:
            out["surface_sdf"] = surf

    def _render_nograd(self, eng, flat, input, cam, dirs, dnorm, z_all, z_eik, predict_only, draws, with_eik):
        with torch.no_grad():
            N, n = z_all.shape[0], z_all.shape[1] - 1
            training = self.training
            M_main = N * n
            extra_pts, n_eik, n_pc = (None, 0, 0)
            if with_eik:
                extra_pts, n_eik, n_pc = self._extra_points(input, cam, dirs, z_eik, draws)
            returns_grad = self.use_normal or (not training) or with_eik
            fw = eng.sdf_forward_grad(points=extra_pts, rays=(cam, dirs, z_all.contiguous(), n), want_grad=returns_grad, save=False)
            rgb, _, _ = eng.rgb_forward(dirs, n, fw["feat"], M_main, save=False)
            lm = None
            if self.use_light:
                lm, _ = eng.light_forward(fw["feat"], M_main, save=False)
            want_normal = (not predict_only) and ((not training) or self.use_normal)
            comp = eng.composite_forward(flat[eng.layout.offset("density.beta"):], z_all.contiguous(), fw["sdf"], rgb, fw["grad"], lm, dnorm,
                                         want_normal=want_normal, save=False)
            out = {"rgb_values": comp["rgb"], "depth_values": comp["depth"], "weight_sum": comp["wsum"]}
            if self.use_light:
                out["light_mask"] = comp["lmask"]
            if predict_only:
                return out
            if training:
                self._eikonal_outputs(out, fw["grad"][M_main:M_main + n_eik], fw["sdf"][M_main + n_eik:], N, n_pc)
                if self.use_normal:
                    out["normal_values"] = comp["normal"]
            else:
                out["normal_map"] = comp["normal"]
            return out
