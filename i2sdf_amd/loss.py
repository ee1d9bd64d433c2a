"""I2SDFLoss with the reference's semantics (model/network/__init__.py:289-406), including its quirks:
`angular_loss` is the same L1 form as `normal_loss` (:368-369) and `angular_weight` defaults to 0.05 although the
shipped configs omit it (:290).  Value and gradients come from ONE fused HIP entry point (i2sdf_loss_forward_backward,
SURVEY row N1); the masked means are sums over selected rows / number of selected rows, i.e. what the reference's boolean
indexing (`depth[depth_mask]`, :320-329) computes, without its device->host synchronisation."""
from __future__ import annotations

import torch
import torch.nn as nn


class _FusedLossFn(torch.autograd.Function):
    """value + gradient of the whole loss in two HIP launches (include/i2sdf.h: i2sdf_loss_forward_backward).  Returns the total as a
    0-dim tensor of its own (the only differentiable output) and the vector of the ten reported values."""

    @staticmethod
    def forward(ctx, cfg, n_pc, gt, scratch, rgb, depth, wsum, normal, grad_theta, diff_norm, surface, lmask):
        from . import lib as L
        lib = L.load()
        dev = rgb.device
        B = rgb.shape[0]
        c = lambda t: None if t is None else t.detach().contiguous()
        f32 = lambda t: None if t is None else t.detach().to(torch.float32).contiguous()
        ins = [c(rgb), c(depth), c(wsum), c(normal), c(grad_theta), c(diff_norm), c(surface), c(lmask)]
        gts = [f32(gt.get("rgb")).reshape(-1, 3), f32(gt.get("depth")), c(gt.get("depth_mask")), f32(gt.get("normal")), c(gt.get("normal_mask")),
               f32(gt.get("mask")), f32(gt.get("light_mask"))]
        if gts[1] is None:
            gts[2] = None
        if gts[3] is None:
            gts[4] = None
        for m in (2, 4):
            if gts[m] is not None:
                assert gts[m].dtype == torch.bool or gts[m].dtype == torch.uint8
        grads = [torch.empty_like(t) if t is not None else None for t in ins]
        losses = torch.empty(10, dtype=torch.float32, device=dev)
        total = torch.empty((), dtype=torch.float32, device=dev)
        with torch.cuda.device(dev):
            L.check(lib.i2sdf_loss_forward_backward(cfg, B, n_pc, *[L.ptr(t) for t in ins], L.ptr(gts[0]), L.ptr(gts[1]), L.ptr(gts[2]),
                                                    L.ptr(gts[3]), L.ptr(gts[4]), L.ptr(gts[5]), L.ptr(gts[6]), L.ptr(scratch), L.ptr(losses),
                                                    L.ptr(total), *[L.ptr(g) for g in grads], L.stream_ptr()), "i2sdf_loss_forward_backward")
        if diff_norm is not None and not (cfg._obj.smooth_on and cfg._obj.smooth_w > 0):
            grads[5] = None      # smoothness term inactive (:347-349): its gradient is identically zero -- let autograd skip that branch
        ctx.grads = grads
        ctx.mark_non_differentiable(losses)
        ctx.set_materialize_grads(False)      # no zero tensor (one fill launch) for the gradient slot of the non-differentiable `losses`
        return total, losses

    @staticmethod
    def backward(ctx, g, _g_items):
        if g is None:
            return (None,) * (4 + len(ctx.grads))
        live = [gr for gr in ctx.grads if gr is not None]
        scaled = iter(torch._foreach_mul(live, g))          # one launch for all of them
        out = [None, None, None, None] + [(next(scaled) if gr is not None else None) for gr in ctx.grads]
        ctx.grads = None
        return tuple(out)


class _FusedRenderLossFn(torch.autograd.Function):
    """I2SDFLoss on the outputs of a training render of i2sdf_amd.I2SDFNetwork, fused with the render's backward down to the per-sample
    gradients (include/i2sdf.h: i2sdf_render_loss_backward): value, the ten reported terms and -- for an upstream gradient of 1 -- sdf_bar /
    rgb_bar / grad_bar / lmask_bar / d loss / d beta in TWO launches, where the separate path runs the loss (2 launches), autograd's
    scaling (1), the eikonal outputs' backward, the seeds, the compositing backward and the beta reduction (4 + 1).  The per-output
    gradients are handed to autograd UNSCALED as recognisable placeholders: _RenderFn.backward checks that they arrive untouched (nobody
    else differentiated the same outputs), multiplies the prepared per-sample gradients by the upstream gradient in one launch and
    continues with the radiance backward; if they were mixed with other gradients it corrects them and takes the general path."""

    @staticmethod
    def forward(ctx, cfg, n_pc, gt, h, rgb, depth, wsum, normal, grad_theta, diff_norm, surface, lmask):
        from . import lib as L
        lib = L.load()
        fz = h["fused"]
        fw, eng = fz["fw"], h["eng"]
        dev = rgb.device
        B, n, M_main, M_sdf = rgb.shape[0], fz["n"], fz["M_main"], fz["fw"]["M"]
        c = lambda t: None if t is None else t.detach().contiguous()
        f32 = lambda t: None if t is None else t.detach().to(torch.float32).contiguous()
        ins = [c(rgb), c(depth), c(wsum), c(normal), c(grad_theta), c(diff_norm), c(surface), c(lmask)]
        gts = [f32(gt.get("rgb")).reshape(-1, 3), f32(gt.get("depth")), c(gt.get("depth_mask")), f32(gt.get("normal")), c(gt.get("normal_mask")),
               f32(gt.get("mask")), f32(gt.get("light_mask"))]
        if gts[1] is None:
            gts[2] = None
        if gts[3] is None:
            gts[4] = None
        for m in (2, 4):
            if gts[m] is not None:
                assert gts[m].dtype == torch.bool or gts[m].dtype == torch.uint8
        toks = [torch.empty_like(t) if t is not None else None for t in ins]
        losses = torch.empty(10, dtype=torch.float32, device=dev)
        total = torch.empty((), dtype=torch.float32, device=dev)
        e = lambda *s_: torch.empty(*s_, dtype=torch.float32, device=dev)
        n_eik = 3 * B if grad_theta is not None else 0
        want_normal = bool(h["want_normal"]) and normal is not None
        pre = {"sbar": e(M_sdf), "nbar": e(M_sdf, 3), "rgb_bar": e(M_main, 3), "lmask_bar": e(M_main) if fz["use_light"] else None, "beta_g": e(1),
               "tok_eik": e(n_eik, 3) if n_eik else None, "g": None, "eik_true": None,
               "tok": dict(zip(("rgb", "depth", "wsum", "normal", "grad_theta", "diff_norm", "surface", "lmask"), toks))}
        scratch = e(int(lib.i2sdf_render_loss_scratch_floats(B)))
        z = h["z_all"]
        with torch.cuda.device(dev):
            L.check(lib.i2sdf_render_loss_backward(cfg, B, n, n_pc, M_main, M_sdf, n_eik,
                                                   L.ptr(fz["beta_param"]), float(fz["beta_min"]), L.ptr(z), z.shape[1], L.ptr(fw["sdf"]), L.ptr(fz["rgb_pts"]),
                                                   L.ptr(fw["grad"]), L.ptr(h["dnorm"]), L.ptr(fz["nsum"]) if want_normal else None,
                                                   *[L.ptr(t) for t in ins], *[L.ptr(t) for t in gts], L.ptr(scratch), L.ptr(losses), L.ptr(total),
                                                   *[L.ptr(t) for t in toks],
                                                   L.ptr(pre["sbar"]), L.ptr(pre["rgb_bar"]), L.ptr(pre["nbar"]), 1 if want_normal else 0, L.ptr(pre["lmask_bar"]),
                                                   L.ptr(pre["beta_g"]), L.stream_ptr()), "i2sdf_render_loss_backward")
        if diff_norm is not None and not (cfg._obj.smooth_on and cfg._obj.smooth_w > 0):
            toks[5] = None      # smoothness term inactive: no gradient for diff_norm (as in _FusedLossFn)
            pre["tok"]["diff_norm"] = None
        h["pre"] = pre
        ctx.pre, ctx.toks, ctx.h = pre, toks, h
        ctx.mark_non_differentiable(losses)
        ctx.set_materialize_grads(False)
        return total, losses

    @staticmethod
    def backward(ctx, g, _g_items):
        if g is None:
            return (None,) * (4 + len(ctx.toks))
        pre, toks, h = ctx.pre, ctx.toks, ctx.h
        ctx.pre = ctx.toks = ctx.h = None
        if h.get("pre") is not pre:
            # a later loss call on the same outputs replaced this call's prepared gradients (or the render's backward already ran): the
            # placeholders mean nothing to _RenderFn.backward any more -- hand autograd the real thing, seeds times the upstream gradient
            live = [t for t in toks if t is not None]
            scaled = iter(torch._foreach_mul(live, g))
            return (None, None, None, None) + tuple((next(scaled) if t is not None else None) for t in toks)
        pre["g"] = g
        return (None, None, None, None) + tuple(toks)


class I2SDFLoss(nn.Module):
    def __init__(self, eikonal_weight=0.1, smooth_weight=0.0, mask_weight=0.0, depth_weight=0.1, normal_weight=0.05, angular_weight=0.05,
                 bubble_weight=0.0, min_bubble_iter=0, max_bubble_iter=None, smooth_iter=None, light_mask_weight=0.0,
                 eikonal_weight_bubble=0.0):
        super().__init__()
        self.eikonal_weight, self.smooth_weight, self.mask_weight = eikonal_weight, smooth_weight, mask_weight
        self.depth_weight, self.normal_weight, self.angular_weight = depth_weight, normal_weight, angular_weight
        self.bubble_weight, self.min_bubble_iter, self.max_bubble_iter = bubble_weight, min_bubble_iter, max_bubble_iter
        self.smooth_iter = smooth_iter
        if self.bubble_weight > 0 and self.max_bubble_iter is not None and self.smooth_iter < self.max_bubble_iter:
            self.smooth_iter = self.max_bubble_iter
        self.light_mask_weight = light_mask_weight
        self._scratch = {}         # per (device, stream): the reduction workspace (stateless, include/i2sdf.h: i2sdf_loss_forward_backward)
        self.exchange = None       # i2sdf_amd.dist.attach_loss: lib.Exchange hook -> global denominators (1-GPU-equivalent data parallelism)
        self._dp_state = None      # the attached module's DataParallelState (its `enabled` flag: no_sync())

    def _forward_fused(self, out, gt, current_step):
        from . import lib as L
        smooth_on = self.smooth_iter is None or current_step > self.smooth_iter
        cfg = L.LossCfg(eikonal_w=self.eikonal_weight, smooth_w=self.smooth_weight, mask_w=self.mask_weight, depth_w=self.depth_weight,
                        normal_w=self.normal_weight, angular_w=self.angular_weight, bubble_w=self.bubble_weight,
                        light_w=self.light_mask_weight, smooth_on=1 if smooth_on else 0)
        import ctypes as C_
        # global denominators only for a training-mode loss of an attached, syncing module: a validation loss on some ranks, or a
        # micro-batch under no_sync(), must not enter a collective the other ranks do not issue (i2sdf_amd.dist.attach_loss)
        st = getattr(self, "_dp_state", None)
        if self.exchange is not None and self.training and (st is None or st.enabled):
            cfg.exchange = C_.pointer(self.exchange)
        elif self.exchange is not None and self.training and not getattr(self, "_warned_no_sync", False):
            import warnings
            warnings.warn("i2sdf_amd.I2SDFLoss: `equivalent` data parallelism under no_sync(): this micro-batch uses RANK-LOCAL loss denominators "
                          "(no exchange while the gradient all-reduce is suspended), so the accumulated step is not the 1-GPU step on the "
                          "concatenated batch; accumulate with plain (non-equivalent) data parallelism, or sync every micro-batch")
            self._warned_no_sync = True
        surf = out.get("surface_sdf")
        gtc = dict(gt)
        if not ("depth" in gt and self.depth_weight > 0):
            gtc.pop("depth", None)
        if not ("normal" in gt and (self.normal_weight > 0 or self.angular_weight > 0) and "normal_values" in out):
            gtc.pop("normal", None)
        if not ("mask" in gt and self.mask_weight > 0):
            gtc.pop("mask", None)
        if not ("light_mask" in out and self.light_mask_weight > 0 and "light_mask" in gt):
            gtc.pop("light_mask", None)
        import ctypes as C
        dev = out["rgb_values"].device
        surf_flat = None if surf is None else surf.reshape(-1)
        args = (out["rgb_values"], out["depth_values"], out["weight_sum"].reshape(-1), out.get("normal_values"), out.get("grad_theta"), out.get("diff_norm"),
                surf_flat, out["light_mask"].reshape(-1) if "light_mask" in out else None)
        names = ["loss", "rgb_loss", "eikonal_loss", "smooth_loss", "mask_loss", "depth_loss", "normal_loss", "angular_loss", "bubble_loss",
                 "light_mask_loss"]
        h = self._render_handle(out, cfg)
        if h is not None:
            total, vec = _FusedRenderLossFn.apply(C.byref(cfg), 0 if surf is None else surf.shape[0], gtc, h, *args)
            res = {n: vec[i] for i, n in enumerate(names)}
            res["loss"] = total
            return res
        # calls in flight on different streams must not share the workspace.  Keyed on the Stream OBJECT (a raw handle can be reused by a
        # later stream -- ADVICE r5) and bounded: a caller that makes a new stream per step does not grow the table
        key = (dev, torch.cuda.current_stream(dev))
        scratch = self._scratch.get(key)
        if scratch is None:
            if len(self._scratch) >= 8:
                self._scratch.pop(next(iter(self._scratch)))
            scratch = self._scratch[key] = torch.empty(int(L.load().i2sdf_loss_scratch_floats()), dtype=torch.float32, device=dev)
        total, vec = _FusedLossFn.apply(C.byref(cfg), 0 if surf is None else surf.shape[0], gtc, scratch, *args)
        res = {n: vec[i] for i, n in enumerate(names)}
        res["loss"] = total
        return res

    @staticmethod
    def _render_handle(out, cfg):
        """The state of the training render these outputs came from, if they ARE its outputs (same tensors, graph attached, grad mode on)
        and nothing needs the separate path (data-parallel exchange of the loss denominators; I2SDF_FUSED_RENDER_LOSS=0 for A/B runs)."""
        import os
        rgb = out["rgb_values"]
        h = getattr(rgb, "_i2sdf_render", None)
        if h is None or "fused" not in h or cfg.exchange or not torch.is_grad_enabled() or not rgb.requires_grad:
            return None
        if os.environ.get("I2SDF_FUSED_RENDER_LOSS", "1") == "0":
            return None
        ptrs = h.get("out_ptrs", {})
        for k in ("rgb_values", "depth_values", "weight_sum", "normal_values", "grad_theta", "diff_norm", "surface_sdf", "light_mask"):
            if (k in out) != (k in ptrs) or (k in out and (out[k].data_ptr() != ptrs[k] or not out[k].is_contiguous())):
                return None
        if ("grad_theta" in out) != ("diff_norm" in out) or out["rgb_values"].dtype != torch.float32:
            return None
        return h

    def forward(self, out, gt, current_step):
        if not (out["rgb_values"].is_cuda and out["rgb_values"].dtype == torch.float32):
            raise RuntimeError("i2sdf_amd.I2SDFLoss runs on MI355X only (fp32 tensors on a cuda/HIP device); there is no "
                               "eager-torch or CPU fallback -- oracle/i2sdf_oracle.py:i2sdf_loss is the CPU restatement")
        return self._forward_fused(out, gt, current_step)
