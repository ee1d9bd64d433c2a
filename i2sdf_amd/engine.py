"""Thin, stateless-as-possible Python driver over the C ABI: owns the plan and the packed-weight buffer,
allocates outputs/workspaces with torch (device memory + streams are torch's job; the arithmetic is the library's)."""
from __future__ import annotations

import ctypes as C
from typing import Optional

import os

import torch

from . import lib as L
from .config import NetConfig
from .params import ParamLayout


class RenderEngine:
    def __init__(self, cfg: NetConfig, device="cuda"):
        self.cfg = cfg
        self.layout = ParamLayout(cfg)
        self.device = torch.device(device)
        self._lib = L.load()
        self._desc = self.layout.net_desc()
        plan = C.c_void_p()
        L.check(self._lib.i2sdf_plan_create(C.byref(self._desc), C.byref(plan)), "i2sdf_plan_create")
        self._plan = plan
        self.pack_floats = int(self._lib.i2sdf_plan_pack_floats(plan))
        self.wgrad_floats = int(self._lib.i2sdf_plan_wgrad_floats(plan))
        self.packed = torch.zeros(self.pack_floats, dtype=torch.float32, device=self.device)
        self._pack_stream, self._pack_event, self._pack_readers = None, None, {}
        self.F = cfg.feature_size
        # every option is SET here, on or off: a fresh plan of a 256-wide net has the bf16x3 twins on by default since round 6 (plan.cpp)
        wide = cfg.sdf.hidden == 256 and cfg.feature_size == 256
        self.set_sdf_forward_bf16x3(bool(cfg.bf16x3 and cfg.sdf.hidden % 64 == 0))
        self.set_wgrad_bf16x3(bool(cfg.bf16x3))
        self.set_train_forward_bf16x3(bool(cfg.bf16x3 and wide))
        self.set_sdf_backward_bf16x3(bool(cfg.bf16x3 and wide))
        self.set_rgb_bf16x3(bool(cfg.bf16x3 and cfg.rgb.hidden == 256 and cfg.feature_size == 256 and cfg.rgb.n_lin >= 3))
        self.wgrad_bf16x2 = False
        # on by default since round 4 (config.py: wgrad_bf16x2; the whole GPU suite runs in both modes, tests/conftest.py);
        # I2SDF_WGRAD_BF16X2=0 / 1 overrides the conf; bench.py reports the fp32-equivalent form as the sub-record `wgrad_bf16x3`
        x2 = os.environ.get("I2SDF_WGRAD_BF16X2", "")
        self.set_wgrad_bf16x2(bool(cfg.bf16x3 and ((x2 != "0") if x2 != "" else cfg.wgrad_bf16x2)))
        # the sampler's sdf-only passes with two split planes (include/i2sdf.h: I2SDF_OPT_SAMPLER_BF16X2); conf `sampler_bf16x2`,
        # I2SDF_SAMPLER_BF16X2=0 / 1 overrides it
        self.sampler_bf16x2 = False
        sx2 = os.environ.get("I2SDF_SAMPLER_BF16X2", "")
        if wide:
            self.set_sampler_bf16x2(bool(self.sdf_forward_bf16x3 and ((sx2 != "0") if sx2 != "" else getattr(cfg, "sampler_bf16x2", False))))
        self.set_blocked_saves(bool(cfg.bf16x3 and os.environ.get("I2SDF_BLOCKED_SAVES", "1") != "0"))      # I2SDF_BLOCKED_SAVES=0 for A/B runs
        # abars / gus / gas as packed 24-bit records (include/i2sdf.h: I2SDF_OPT_SAVES24; takes effect only with the two-plane weight gradients
        # and the point ranges: the fp32-equivalent mode keeps fp32 storage); conf `saves24`, I2SDF_SAVES24=0 / 1 overrides it
        self.saves24 = False
        s24 = os.environ.get("I2SDF_SAVES24", "")
        if wide:
            self.set_saves24(bool(cfg.bf16x3 and ((s24 != "0") if s24 != "" else getattr(cfg, "saves24", True))))
        self.set_tail_overlap(os.environ.get("I2SDF_TAIL_OVERLAP", "1") != "0")                            # I2SDF_TAIL_OVERLAP=0 for A/B runs
        # point ranges on their own streams instead of split-K tail workgroups (include/i2sdf.h: I2SDF_OPT_PARTS); I2SDF_PARTS=0 for A/B runs
        self.parts = 0
        n_parts = int(os.environ.get("I2SDF_PARTS", "2"))
        self._parts_wanted = n_parts if (cfg.bf16x3 and n_parts >= 2) else 0
        self._sync_parts()
        sc = cfg.sampler
        self._scfg = L.SamplerCfg(near=sc.near, eps=sc.eps, add_tiny=sc.add_tiny, N_samples=sc.N_samples, N_samples_eval=sc.N_samples_eval,
                                  N_samples_extra=sc.N_samples_extra, beta_iters=sc.beta_iters, max_total_iters=sc.max_total_iters)
        # deterministic tables the reference builds with torch.linspace on every call (ray_sampler.py:30,188,225)
        dev = self.device
        self.t_lin = torch.linspace(0.0, 1.0, steps=sc.N_samples_eval, device=dev)
        self.u_more = torch.linspace(0.0, 1.0, steps=sc.N_samples_eval, device=dev)
        self.u_final = torch.linspace(0.0, 1.0, steps=sc.N_samples, device=dev)
        if sc.N_samples_extra > 0:
            self.extra_tab = torch.stack([torch.linspace(0, sc.N_samples_eval * (it + 1) - 1, sc.N_samples_extra).long()
                                          for it in range(sc.max_total_iters)]).to(torch.int32).to(dev).contiguous()
        else:
            self.extra_tab = None
        self.n_z = sc.N_samples + sc.N_samples_extra + 2

    # -- live per-kernel timing (bench.py) -------------------------------------------------------------
    def start_timing(self):
        self._timing = []
        eng = self

        class _Timed:
            def __init__(self, fn, name):
                self.fn, self.name = fn, name

            def __call__(self, *a):
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
                rc = self.fn(*a)
                e1.record()
                eng._timing.append((self.name, e0, e1))
                return rc

        self._timed_lib = type("TimedLib", (), {})()
        for name in L.SIGNATURES:
            fn = getattr(self._lib, name)
            if name in ("i2sdf_sdf_forward", "i2sdf_sdf_forward_grad", "i2sdf_rgb_forward", "i2sdf_rgb_backward", "i2sdf_sdf_backward",
                        "i2sdf_weight_grads", "i2sdf_sample_rays", "i2sdf_composite_forward", "i2sdf_composite_backward", "i2sdf_pack_weights",
                        "i2sdf_ray_setup", "i2sdf_light_forward", "i2sdf_light_backward"):
                fn = _Timed(fn, name)
            setattr(self._timed_lib, name, fn)
        self._real_lib, self._lib = self._lib, self._timed_lib

    def stop_timing(self):
        """-> {entry point: (total ms, launches)}; call after a device synchronisation."""
        self._lib = self._real_lib
        out = {}
        for name, e0, e1 in self._timing:
            tot, cnt = out.get(name, (0.0, 0))
            out[name] = (tot + e0.elapsed_time(e1), cnt + 1)
        self._timing = []
        return out

    def __del__(self):
        try:
            if getattr(self, "_plan", None):
                self._lib.i2sdf_plan_destroy(self._plan)
                self._plan = None
        except Exception:
            pass

    # -- weights -------------------------------------------------------------------------------
    def pack(self, flat_params: torch.Tensor):
        """weight-norm reparametrisation + stream packing; call after every parameter update."""
        assert flat_params.is_cuda and flat_params.dtype == torch.float32 and flat_params.numel() == self.layout.n_params
        # On a side stream: the two pack launches (35 us) then run beside the ray set-up, the draws and the sampler's first kernel instead
        # of in front of them; the first entry point that reads the streams waits for the event (`_pk`).  The side stream first waits
        # for everything enqueued so far (the optimizer's update of the parameters, the previous step's readers of `packed`).
        if os.environ.get("I2SDF_PACK_ASYNC", "1") == "0":
            L.check(self._lib.i2sdf_pack_weights(self._plan, L.ptr(flat_params), L.ptr(self.packed), L.stream_ptr()), "i2sdf_pack_weights")
            return
        dev = flat_params.device
        side = self._pack_stream
        if side is None:
            side = self._pack_stream = torch.cuda.Stream(device=dev)
        cur = torch.cuda.current_stream(dev)
        side.wait_stream(cur)
        # streams that read `packed` since the previous pack (entry points issued on other streams than the one calling pack()): the
        # new pack must not overwrite the weight streams under them
        for st in getattr(self, "_pack_readers", {}):
            if st != cur:
                side.wait_stream(st)
        with torch.cuda.stream(side):
            L.check(self._lib.i2sdf_pack_weights(self._plan, L.ptr(flat_params), L.ptr(self.packed), L.stream_ptr()), "i2sdf_pack_weights")
            self._pack_event = side.record_event()
        # the side stream reads the flat parameter buffer: the caching allocator must not hand its memory out again (module.to(), a
        # re-flattened buffer) while the pack is still reading it
        flat_params.record_stream(side)
        self._pack_readers = {}

    def _pk(self):
        """pointer to the packed weight streams, after making the CURRENT stream wait for the last pack.  The event is kept until the next
        pack(): every stream that has not waited for it yet does so once (an eval render or an autograd backward on another stream than
        the first entry point's reads the same weight streams -- ADVICE r4)."""
        ev = self._pack_event
        if ev is not None:
            cur = torch.cuda.current_stream(self.packed.device)
            # keyed on the torch Stream OBJECT (hashable; the dict keeps it alive until the next pack): a raw handle can be reused by a
            # new stream after its owner was destroyed, and that stream would be mistaken for one that has already waited
            readers = self.__dict__.setdefault("_pack_readers", {})
            if cur not in readers:
                cur.wait_event(ev)
                readers[cur] = True
        return L.ptr(self.packed)

    def set_sdf_forward_bf16x3(self, on: bool):
        """Evaluate the sdf-only forward (sampler passes, grid queries) in bf16x3 split arithmetic (include/i2sdf.h,
        I2SDF_OPT_SDF_FWD_BF16X3): fp32-level accuracy on the bf16 matrix pipe."""
        L.check(self._lib.i2sdf_plan_set_option(self._plan, L.OPT_SDF_FWD_BF16X3, int(bool(on))), "i2sdf_plan_set_option")
        self.sdf_forward_bf16x3 = bool(on)

    def set_train_forward_bf16x3(self, on: bool):
        """SDF forward + d sdf/dx kernel (full workgroups) in bf16x3 split arithmetic (I2SDF_OPT_TRAIN_FWD_BF16X3)."""
        L.check(self._lib.i2sdf_plan_set_option(self._plan, L.OPT_TRAIN_FWD_BF16X3, int(bool(on))), "i2sdf_plan_set_option")
        self.train_forward_bf16x3 = bool(on)
        if hasattr(self, "_parts_wanted"):
            self._sync_parts()

    def set_sdf_backward_bf16x3(self, on: bool):
        """SDF backward sweeps (full workgroups) in bf16x3 split arithmetic (I2SDF_OPT_SDF_BWD_BF16X3)."""
        L.check(self._lib.i2sdf_plan_set_option(self._plan, L.OPT_SDF_BWD_BF16X3, int(bool(on))), "i2sdf_plan_set_option")
        self.sdf_backward_bf16x3 = bool(on)
        if hasattr(self, "_parts_wanted"):
            self._sync_parts()

    def set_rgb_bf16x3(self, on: bool):
        """Radiance forward / backward (full workgroups) in bf16x3 split arithmetic (I2SDF_OPT_RGB_BF16X3)."""
        L.check(self._lib.i2sdf_plan_set_option(self._plan, L.OPT_RGB_BF16X3, int(bool(on))), "i2sdf_plan_set_option")
        self.rgb_bf16x3 = bool(on)
        if hasattr(self, "_parts_wanted"):
            self._sync_parts()

    def set_wgrad_bf16x2(self, on: bool):
        """256x256 weight-gradient blocks with two split planes / three products (I2SDF_OPT_WGRAD_BF16X2): 16+ mantissa bits per
        operand -- above the reference's own `float32_matmul_precision('medium')`, below fp32; the gradients stay inside the 1e-4 bar."""
        L.check(self._lib.i2sdf_plan_set_option(self._plan, L.OPT_WGRAD_BF16X2, int(bool(on))), "i2sdf_plan_set_option")
        self.wgrad_bf16x2 = bool(on)

    def set_sampler_bf16x2(self, on: bool):
        """The sampler's sdf-only passes (i2sdf_sample_rays, i2sdf_render_image) with two split planes / three products
        (I2SDF_OPT_SAMPLER_BF16X2): they choose depths; every returned value still comes from the fp32-equivalent kernels."""
        rc = self._lib.i2sdf_plan_set_option(self._plan, L.OPT_SAMPLER_BF16X2, int(bool(on)))
        if on or rc == 0:                      # (turning it OFF on a library build that predates the option is not an error: A/B runs)
            L.check(rc, "i2sdf_plan_set_option")
        self.sampler_bf16x2 = bool(on) and rc == 0

    def set_blocked_saves(self, on: bool):
        """Saved 256-wide tensors of the bf16x3 full workgroups in the blocked layout (I2SDF_OPT_BLOCKED_SAVES, csrc/mlp_common.h).
        Change it only between training steps: forward, backward and weight gradients of one step must agree."""
        L.check(self._lib.i2sdf_plan_set_option(self._plan, L.OPT_BLOCKED_SAVES, int(bool(on))), "i2sdf_plan_set_option")
        self.blocked_saves = bool(on)

    def set_saves24(self, on: bool):
        """abars / gus / gas with 16 significant bits in 3 bytes per value (I2SDF_OPT_SAVES24, csrc/x3.h P24).  Change it only between
        training steps.  `saves24_points` tells whether it takes effect under the other options."""
        rc = self._lib.i2sdf_plan_set_option(self._plan, L.OPT_SAVES24, int(bool(on)))
        if on or rc == 0:                      # (turning it OFF on a library build that predates the option is not an error: A/B runs)
            L.check(rc, "i2sdf_plan_set_option")
        self.saves24 = bool(on) and rc == 0

    def blocked_points(self, which: int, M: int, Mp: int, has_feat: bool = True) -> int:
        return int(self._lib.i2sdf_blocked_points(self._plan, which, M, Mp, int(has_feat)))

    def saves24_points(self, M: int, Mp: int, has_feat: bool = True) -> int:
        """leading points of a batch whose abars / gus / gas are packed 24-bit records (Mp or 0)"""
        return int(self._lib.i2sdf_blocked_points(self._plan, 2, M, Mp, int(has_feat))) if self.saves24 else 0

    @staticmethod
    def saved_to_point_major(t: torch.Tensor, n_blocked: int, p24_layers=None) -> torch.Tensor:
        """A copy of a saved (layers, Mp, 256) tensor with ordinary rows (inspection / tests): the first n_blocked points are stored
        [Mp/32][16 k-chunks][32 points][16 floats] (include/i2sdf.h: i2sdf_blocked_points).  p24_layers: for each layer, whether its
        (all-blocked) points are packed 24-bit records (csrc/x3.h P24: per 32-point block 16 k-chunks of [32 points][2][8 upper halves]
        + [32 points][2][8 mid bytes]; value u of lane hi = column (u < 4 ? 4 hi + u : 8 + 4 hi + u - 4) of the k-chunk) -- decoded to fp32."""
        if n_blocked <= 0:
            return t.clone()
        Lr, Mp, H = t.shape
        assert H == 256 and n_blocked % 32 == 0
        out = t.clone()
        out[:, :n_blocked] = t[:, :n_blocked].reshape(Lr, n_blocked // 32, 16, 32, 16).permute(0, 1, 3, 2, 4).reshape(Lr, n_blocked, 256)
        if p24_layers is not None and any(p24_layers):
            assert n_blocked == Mp, "packed records: every point is blocked"
            nb = Mp // 32
            for l in range(Lr):
                if not p24_layers[l]:
                    continue
                raw = t[l].reshape(-1)[: nb * 6144].view(torch.int32).reshape(nb, 16, 384)
                hi16 = raw[:, :, :256].contiguous().view(torch.int16).reshape(nb, 16, 32, 2, 8).to(torch.int32) & 0xFFFF     # [blk][kc][p][hi][u]
                mid = raw[:, :, 256:].contiguous().view(torch.uint8).reshape(nb, 16, 32, 2, 8).to(torch.int32)
                val = ((hi16 << 16) | (mid << 8)).view(torch.float32)                                                # wraps to the sign bit as intended
                # (hi, u) -> column of the k-chunk: u < 4: 4 hi + u ; u >= 4: 8 + 4 hi + (u - 4)
                v = val.reshape(nb, 16, 32, 2, 2, 4).permute(0, 2, 1, 4, 3, 5)       # [blk][p][kc][half][hi][4]
                out[l] = v.reshape(Mp, 256)
        return out

    def saved_pm(self, name: str, t: torch.Tensor, M: int, has_feat: bool = True) -> torch.Tensor:
        """rows 0..M-1 of a saved tensor (`hs`, `abars`, `gus`, `gas`, `rs`, `gar`) in point-major fp32 form, whatever its storage"""
        Mp = t.shape[1]
        if name in ("rs", "gar"):
            return self.saved_to_point_major(t, self.blocked_points(1, M, Mp))[:, :M]
        nb = self.blocked_points(0, M, Mp, has_feat)
        p24 = None
        if name in ("abars", "gus", "gas") and self.saves24_points(M, Mp, has_feat) == Mp:
            p24 = [True] * t.shape[0]
            if name == "gus":
                p24[-1] = False          # G(hbar_{L-1}) stays fp32 (csrc/x3.h: X3Sweep2Src)
                p24[0] = False           # (slot 0 is unused)
        return self.saved_to_point_major(t, nb, p24)[:, :M]

    def set_tail_overlap(self, on: bool):
        """Split-K tail workgroups on the plan's side stream, concurrent with the full workgroups (I2SDF_OPT_TAIL_OVERLAP)."""
        L.check(self._lib.i2sdf_plan_set_option(self._plan, L.OPT_TAIL_OVERLAP, int(bool(on))), "i2sdf_plan_set_option")
        self.tail_overlap = bool(on)

    def _sync_parts(self):
        """Point ranges need every per-point kernel family on the ranged (bf16x3 full-workgroup) path: with the flags mixed an entry point
        of a chain would run whole-batch between two ranged ones (the library then joins and fences around it, at the price of the overlap)."""
        want = self._parts_wanted if (self.train_forward_bf16x3 and self.sdf_backward_bf16x3 and self.rgb_bf16x3) else 0
        if want != self.parts:
            n = int(want) if int(want) >= 2 else 0
            L.check(self._lib.i2sdf_plan_set_option(self._plan, L.OPT_PARTS, n), "i2sdf_plan_set_option")
            self.parts = n

    def set_parts(self, n: int):
        """Cut the per-point entry points into n point ranges, each on its own stream (I2SDF_OPT_PARTS; 0 / 1 = off).  Change it only
        between training steps (the layout of the saved tensors depends on it)."""
        self._parts_wanted = int(n) if int(n) >= 2 else 0
        self._sync_parts()

    def chain(self, M: int):
        """Context manager: the per-point entry points called inside leave their point ranges un-joined (i2sdf_chain_begin / _end);
        a no-op without I2SDF_OPT_PARTS.  M = the largest point batch of the chain."""
        eng = self
        if not getattr(self, "use_chain", True) or not self.parts:
            import contextlib
            return contextlib.nullcontext()

        class _Chain:
            def __enter__(self_):
                L.check(eng._lib.i2sdf_chain_begin(eng._plan, int(M), L.stream_ptr()), "i2sdf_chain_begin")
                return self_

            def __exit__(self_, *exc):
                L.check(eng._lib.i2sdf_chain_end(eng._plan, L.stream_ptr()), "i2sdf_chain_end")
                return False

            def fence(self_):
                L.check(eng._lib.i2sdf_chain_fence(eng._plan, L.stream_ptr()), "i2sdf_chain_fence")

        return _Chain()

    def set_exchange(self, ex, flags: int = 0):
        """Install (or with ex=None remove) the data-parallel exchange hook of the plan (include/i2sdf.h: i2sdf_plan_set_exchange);
        flags: lib.DP_GLOBAL_SAMPLER = the sampler's convergence test is the OR over all ranks' rays."""
        self._exchange = ex                       # keep the struct (and through it the callback) alive
        L.check(self._lib.i2sdf_plan_set_exchange(self._plan, C.byref(ex) if ex is not None else None, int(flags)), "i2sdf_plan_set_exchange")

    def set_wgrad_bf16x3(self, on: bool):
        """Full 256x256 weight-gradient blocks in bf16x3 split arithmetic (I2SDF_OPT_WGRAD_BF16X3)."""
        L.check(self._lib.i2sdf_plan_set_option(self._plan, L.OPT_WGRAD_BF16X3, int(bool(on))), "i2sdf_plan_set_option")
        self.wgrad_bf16x3 = bool(on)

    # -- SDF queries ---------------------------------------------------------------------------
    def sdf_forward(self, points: torch.Tensor, want_features: bool = False):
        """ImplicitNetwork.forward without grad: returns sdf (M,1) [, feature (M,F)]."""
        pts = points.detach().to(torch.float32).contiguous()
        M = pts.shape[0]
        sdf = torch.empty(M, 1, dtype=torch.float32, device=pts.device)
        feat = torch.empty(M, self.F, dtype=torch.float32, device=pts.device) if want_features else None
        L.check(self._lib.i2sdf_sdf_forward(self._plan, self._pk(), L.ptr(pts), M, L.ptr(sdf), L.ptr(feat), self.F,
                                            L.stream_ptr()), "i2sdf_sdf_forward")
        return (sdf, feat) if want_features else sdf

    # -- training-mode forward pieces (low level; the nn.Module composes them) ------------------
    @staticmethod
    def pad_rows(M: int) -> int:
        return (M + 127) // 128 * 128

    def _ws(self, *shape, device=None):
        """A per-point workspace of Mp rows (include/i2sdf.h: the kernels may write its padding rows M..Mp, never a row beyond Mp).
        One allocation site, so that tests/test_gpu_edge_cases.py can hand out buffers with a canary block behind the last row."""
        return torch.empty(*shape, dtype=torch.float32, device=device)

    def sdf_forward_grad(self, points=None, rays=None, want_grad=True, save=True, want_feat=True):
        """Point batch = [points generated from rays=(cam (B,3), dirs (B,3), z (B,n)) | explicit points (P,3)]; either may be None.
        Returns dict(sdf, feat, grad, hs, abars, pe, Mp, M, n_ray_pts)."""
        cfgs = self.cfg.sdf
        H, NL = cfgs.hidden, cfgs.n_lin
        pts = cam = dirs = z = None
        n_ray, ldz, npr = 0, 0, 1
        if rays is not None:
            cam, dirs, z = (t.detach().to(torch.float32).contiguous() for t in rays[:3])
            npr = rays[3] if len(rays) > 3 else z.shape[1]      # samples used per ray (z may carry z_max in its last column)
            n_ray, ldz = z.shape[0] * npr, z.shape[1]
            dev = z.device
        if points is not None:
            pts = points.detach().to(torch.float32).contiguous()
            dev = pts.device
        M = n_ray + (pts.shape[0] if pts is not None else 0)
        Mp = self.pad_rows(M)
        out = {"Mp": Mp, "M": M, "n_ray_pts": n_ray, "pts": pts, "rays": (cam, dirs, z, npr), "sdf": torch.empty(M, 1, dtype=torch.float32, device=dev),
               "blk": self.blocked_points(0, M, Mp, want_feat)}         # leading points of hs / abars in the blocked layout
        out["feat"] = self._ws(Mp, self.F, device=dev) if want_feat else None
        out["grad"] = torch.empty(M, 3, dtype=torch.float32, device=dev) if want_grad else None
        out["hs"] = self._ws(NL - 1, Mp, H, device=dev) if (save or want_grad) else None
        out["abars"] = self._ws(NL - 1, Mp, H, device=dev) if (save and want_grad) else None
        out["pe"] = self._ws(Mp, 40, device=dev) if save else None
        L.check(self._lib.i2sdf_sdf_forward_grad(self._plan, self._pk(), L.ptr(pts), L.ptr(cam), L.ptr(dirs), L.ptr(z),
                                                  ldz, npr, n_ray, M, Mp, L.ptr(out["sdf"]), L.ptr(out["feat"]), L.ptr(out["grad"]),
                                                  L.ptr(out["hs"]), L.ptr(out["abars"]), L.ptr(out["pe"]), L.stream_ptr()),
                 "i2sdf_sdf_forward_grad")
        return out

    def rgb_forward(self, dirs, n_per_ray, feat, M, save=True):
        Mp = feat.shape[0]
        rgb = torch.empty(M, 3, dtype=torch.float32, device=feat.device)
        Lr, Hr = self.cfg.rgb.n_lin, self.cfg.rgb.hidden
        rs = self._ws(Lr - 1, Mp, Hr, device=feat.device) if save else None
        pev = self._ws(Mp, 32, device=feat.device) if save else None
        L.check(self._lib.i2sdf_rgb_forward(self._plan, self._pk(), L.ptr(dirs.contiguous()), n_per_ray, L.ptr(feat), M, Mp,
                                             L.ptr(rgb), L.ptr(rs), L.ptr(pev), L.stream_ptr()), "i2sdf_rgb_forward")
        return rgb, rs, pev

    def rgb_backward(self, rgb, rgb_bar, rs, M):
        Mp, dev = rs.shape[1], rs.device
        Lr, Hr = self.cfg.rgb.n_lin, self.cfg.rgb.hidden
        gar = self._ws(Lr - 1, Mp, Hr, device=dev)
        ga_last = self._ws(Mp, 4, device=dev)
        fbar = self._ws(Mp, self.F, device=dev)
        L.check(self._lib.i2sdf_rgb_backward(self._plan, self._pk(), L.ptr(rgb), L.ptr(rgb_bar.contiguous()), L.ptr(rs), M, Mp,
                                              L.ptr(gar), L.ptr(ga_last), L.ptr(fbar), L.stream_ptr()), "i2sdf_rgb_backward")
        return gar, ga_last, fbar

    def sdf_backward(self, fw, sbar=None, fbar=None, m_fbar=0, nbar=None):
        """fw = dict from sdf_forward_grad(save=True).  Returns the weight-gradient operands."""
        cfgs = self.cfg.sdf
        H, NL = cfgs.hidden, cfgs.n_lin
        Mp, M, dev = fw["Mp"], fw["M"], fw["hs"].device
        cam, dirs, z, npr = fw["rays"]
        o = {"gus": self._ws(NL, Mp, H, device=dev), "gpbar": self._ws(Mp, 40, device=dev), "gas": self._ws(NL - 1, Mp, H, device=dev),
             "ga_last4": self._ws(Mp, 4, device=dev), "ones4": self._ws(Mp, 4, device=dev)}
        c = lambda t: None if t is None else t.contiguous()
        L.check(self._lib.i2sdf_sdf_backward(self._plan, self._pk(), L.ptr(fw["pts"]), L.ptr(cam), L.ptr(dirs), L.ptr(z),
                                              z.shape[1] if z is not None else 0, npr, fw["n_ray_pts"], M, Mp,
                                              L.ptr(fw["hs"]), L.ptr(fw["abars"]), L.ptr(c(sbar)), L.ptr(c(fbar)), m_fbar, L.ptr(c(nbar)),
                                              L.ptr(o["gus"]), L.ptr(o["gpbar"]), L.ptr(o["gas"]), L.ptr(o["ga_last4"]), L.ptr(o["ones4"]),
                                              L.stream_ptr()), "i2sdf_sdf_backward")
        return o

    def weight_grads(self, flat_params, grad_flat, fw, bw, M_main=0, fbar=None, rgb_fw=None, rgb_bw=None, light=None):
        """Runs the weight-gradient GEMMs + weight-norm backward; writes into grad_flat (layout of flat_params)."""
        tb = L.TrainBuffers()
        tb.M_sdf, tb.M_main, tb.Mp = fw["M"], M_main, fw["Mp"]
        P = lambda t: None if t is None else t.data_ptr()
        tb.pe, tb.hs, tb.abars = P(fw["pe"]), P(fw["hs"]), P(fw["abars"])
        tb.gus, tb.gpbar, tb.gas, tb.ga_last4, tb.ones4 = P(bw["gus"]), P(bw["gpbar"]), P(bw["gas"]), P(bw["ga_last4"]), P(bw["ones4"])
        tb.fbar = P(fbar)
        tb.feat = P(fw["feat"])
        if rgb_fw is not None:
            tb.pev, tb.rs = P(rgb_fw["pev"]), P(rgb_fw["rs"])
            tb.gar, tb.ga_last_rgb = P(rgb_bw["gar"]), P(rgb_bw["ga_last"])
        if light is not None:
            tb.hl, tb.gal0, tb.gal_last = P(light["hl"]), P(light["gal0"]), P(light["gal_last"])
        ch = int(self._lib.i2sdf_wgrad_chunk_points())
        n_chunks = (fw["M"] + ch - 1) // ch
        partials = torch.empty(n_chunks * self.wgrad_floats, dtype=torch.float32, device=flat_params.device)
        L.check(self._lib.i2sdf_weight_grads(self._plan, C.byref(tb), L.ptr(flat_params), L.ptr(partials), n_chunks, L.ptr(grad_flat),
                                              L.stream_ptr()), "i2sdf_weight_grads")
        return partials

    # -- per-ray kernels -------------------------------------------------------------------------
    def ray_setup(self, uv, pose, intrinsics):
        """pose: (batch,4,4) cam->world or (batch,7) [quaternion, translation] (utils/rend_util.py:93-101)."""
        uv, pose, intrinsics = (t.detach().to(torch.float32).contiguous() for t in (uv, pose, intrinsics))
        quat = pose.dim() == 2 and pose.shape[1] == 7
        if not quat and tuple(pose.shape[1:]) != (4, 4):
            raise ValueError(f"pose must be (batch,4,4) or (batch,7), got {tuple(pose.shape)}")
        batch, pixels = uv.shape[0], uv.shape[1]
        N = batch * pixels
        cam = torch.empty(N, 3, dtype=torch.float32, device=uv.device)
        dirs = torch.empty_like(cam)
        dnorm = torch.empty(N, dtype=torch.float32, device=uv.device)
        L.check(self._lib.i2sdf_ray_setup_ex(L.ptr(uv), L.ptr(pose), int(quat), L.ptr(intrinsics), batch, pixels, L.ptr(cam),
                                              L.ptr(dirs), L.ptr(dnorm), L.stream_ptr()), "i2sdf_ray_setup_ex")
        return cam, dirs, dnorm

    def sphere_intersections(self, cam_loc, dirs, radius):
        """utils/rend_util.py:211-227; raises where the reference prints 'BOUNDING SPHERE PROBLEM!' and exit()s."""
        cam_loc, dirs = cam_loc.detach().float().contiguous(), dirs.detach().float().contiguous()
        N = cam_loc.shape[0]
        out = torch.empty(N, 2, dtype=torch.float32, device=cam_loc.device)
        miss = torch.zeros(1, dtype=torch.int32, device=cam_loc.device)
        L.check(self._lib.i2sdf_sphere_intersections(L.ptr(cam_loc), L.ptr(dirs), N, float(radius), L.ptr(out), L.ptr(miss),
                                                      L.stream_ptr()), "i2sdf_sphere_intersections")
        if int(miss.item()) > 0:
            raise ValueError(f"BOUNDING SPHERE PROBLEM: {int(miss.item())} rays do not intersect the sphere of radius {radius}")
        return out

    def composite_forward(self, beta_param, z_all, sdf, rgb, grad, lmask, dnorm, want_normal, save=True, eik_grad=None):
        """eik_grad (3B,3): d sdf / d x of the extra points -> also o["grad_theta"] (2B,3), o["diff_norm"] (B) from the same launch
        (i2sdf_composite_forward_eik)"""
        B, n = z_all.shape[0], z_all.shape[1] - 1
        dev = z_all.device
        o = {"rgb": torch.empty(B, 3, device=dev), "depth": torch.empty(B, device=dev), "wsum": torch.empty(B, 1, device=dev)}
        o["normal"] = torch.empty(B, 3, device=dev) if want_normal else None
        o["lmask"] = torch.empty(B, 1, device=dev) if lmask is not None else None
        o["w"] = torch.empty(B, n, device=dev) if save else None
        o["nsum"] = torch.empty(B, 3, device=dev) if (save and want_normal) else None
        if eik_grad is not None and eik_grad.shape[0] == 3 * B and B > 0:
            o["grad_theta"], o["diff_norm"] = torch.empty(2 * B, 3, device=dev), torch.empty(B, device=dev)
            L.check(self._lib.i2sdf_composite_forward_eik(L.ptr(beta_param), self.cfg.beta_min, L.ptr(z_all), z_all.shape[1], L.ptr(sdf),
                                                           L.ptr(rgb), L.ptr(grad) if want_normal else None, L.ptr(lmask), L.ptr(dnorm), B, n,
                                                           L.ptr(o["rgb"]), L.ptr(o["depth"]), L.ptr(o["wsum"]), L.ptr(o["normal"]),
                                                           L.ptr(o["lmask"]), L.ptr(o["w"]), L.ptr(o["nsum"]), L.ptr(eik_grad.contiguous()),
                                                           L.ptr(o["grad_theta"]), L.ptr(o["diff_norm"]), L.stream_ptr()), "i2sdf_composite_forward_eik")
            return o
        L.check(self._lib.i2sdf_composite_forward(L.ptr(beta_param), self.cfg.beta_min, L.ptr(z_all), z_all.shape[1], L.ptr(sdf),
                                                   L.ptr(rgb), L.ptr(grad) if want_normal else None, L.ptr(lmask), L.ptr(dnorm), B, n,
                                                   L.ptr(o["rgb"]), L.ptr(o["depth"]), L.ptr(o["wsum"]), L.ptr(o["normal"]),
                                                   L.ptr(o["lmask"]), L.ptr(o["w"]), L.ptr(o["nsum"]), L.stream_ptr()),
                 "i2sdf_composite_forward")
        return o

    def composite_backward(self, beta_param, z_all, sdf, rgb, grad, dnorm, nsum, g_rgb, g_depth, g_wsum, g_normal, g_lmask,
                           beta_grad_accum=None, sdf_bar_out=None, grad_bar_out=None):
        B, n = z_all.shape[0], z_all.shape[1] - 1
        dev = z_all.device
        M = B * n
        o = {"sdf_bar": sdf_bar_out if sdf_bar_out is not None else torch.empty(M, device=dev), "rgb_bar": torch.empty(M, 3, device=dev)}
        # (a caller-provided grad_bar is written even without g_normal -- zeros for the rays' rows -- so that it never stays uninitialised)
        o["grad_bar"] = grad_bar_out if grad_bar_out is not None else (torch.empty(M, 3, device=dev) if g_normal is not None else None)
        o["lmask_bar"] = torch.empty(M, device=dev) if g_lmask is not None else None
        part = torch.empty(B, device=dev)
        c = lambda t: None if t is None else t.contiguous()
        L.check(self._lib.i2sdf_composite_backward(L.ptr(beta_param), self.cfg.beta_min, L.ptr(z_all), z_all.shape[1], L.ptr(sdf),
                                                    L.ptr(rgb), L.ptr(grad), L.ptr(dnorm), L.ptr(nsum), B, n, L.ptr(c(g_rgb)),
                                                    L.ptr(c(g_depth)), L.ptr(c(g_wsum)), L.ptr(c(g_normal)), L.ptr(c(g_lmask)),
                                                    L.ptr(o["sdf_bar"]), L.ptr(o["rgb_bar"]), L.ptr(o["grad_bar"]), L.ptr(o["lmask_bar"]),
                                                    L.ptr(part), L.ptr(beta_grad_accum), L.stream_ptr()), "i2sdf_composite_backward")
        o["beta_partial"] = part
        return o

    # -- sampler -----------------------------------------------------------------------------------
    def training_draws(self, B: int, seed: int, device, eik_radius: float = 0.0, want_eik: bool = True):
        """Every random draw of one training forward in ONE launch (include/i2sdf.h: i2sdf_training_draws): dict(strat_u (B,N_eval),
        cdf_u (B,N_samples), extra_idx (max_total_iters, N_extra) int32 | None, eik_idx (B) int32, eik_pts (B,3), nbr_off (B,3))."""
        sc = self.cfg.sampler
        e = lambda *s, dt=torch.float32: torch.empty(*s, dtype=dt, device=device)
        d = {"strat_u": e(B, sc.N_samples_eval), "cdf_u": e(B, sc.N_samples),
             "extra_idx": e(sc.max_total_iters, sc.N_samples_extra, dt=torch.int32) if sc.N_samples_extra > 0 else None,
             "eik_idx": e(B, dt=torch.int32), "eik_pts": e(B, 3) if want_eik else None, "nbr_off": e(B, 3) if want_eik else None}
        L.check(self._lib.i2sdf_training_draws(int(seed) & (2 ** 64 - 1), B, sc.N_samples_eval, sc.N_samples, sc.N_samples_extra, sc.max_total_iters,
                                               self.n_z, float(eik_radius), 0.005, L.ptr(d["strat_u"]), L.ptr(d["cdf_u"]), L.ptr(d["extra_idx"]),
                                               L.ptr(d["eik_idx"]), L.ptr(d["eik_pts"]), L.ptr(d["nbr_off"]), L.stream_ptr()), "i2sdf_training_draws")
        return d

    def sample_rays(self, flat_params, cam, dirs, training=False, strat_u=None, cdf_u=None, extra_idx=None, eik_idx=None, force_iters=0):
        """ErrorBoundSampler.get_z_vals on the device.  Returns z_all (B, N_samples+N_extra+2), z_eik (B,1), iters (device int32)."""
        B, dev = cam.shape[0], cam.device
        ws = torch.empty(int(self._lib.i2sdf_sampler_workspace_floats(B)), dtype=torch.float32, device=dev)
        z_out = torch.empty(B, self.n_z, dtype=torch.float32, device=dev)
        z_eik = torch.empty(B, 1, dtype=torch.float32, device=dev)
        iters = torch.empty(1, dtype=torch.int32, device=dev) if B > 0 else torch.zeros(1, dtype=torch.int32, device=dev)      # written by the final kernel
        i32 = lambda t: None if t is None else t.to(torch.int32).contiguous()
        if training:
            u_final, ldu = cdf_u.contiguous(), cdf_u.shape[1]
        else:
            u_final, ldu = self.u_final, 0
        ex, ek = i32(extra_idx), i32(eik_idx)
        su = None if strat_u is None else strat_u.contiguous()
        L.check(self._lib.i2sdf_sample_rays(self._plan, self._pk(), L.ptr(flat_params), C.byref(self._scfg), L.ptr(cam.contiguous()),
                                             L.ptr(dirs.contiguous()), B, 1 if training else 0, L.ptr(self.t_lin), L.ptr(self.u_more),
                                             L.ptr(u_final), ldu, L.ptr(self.extra_tab), L.ptr(su), L.ptr(ex), L.ptr(ek), force_iters,
                                             L.ptr(ws), L.ptr(z_out), self.n_z, L.ptr(z_eik), L.ptr(iters), L.stream_ptr()),
                 "i2sdf_sample_rays")
        self._last_sampler_ws = ws
        return z_out, z_eik, iters

    def error_bound(self, z, sdf, beta, d_star=None, want_d_star=False):
        """ErrorBoundSampler.get_error_bound (ray_sampler.py:243-251) on rows z, sdf (B,n); beta scalar tensor or (B,) / (B,1).
        d_star=None computes the Theorem-1 bound (:99-114).  Returns bound (B,) [, d_star (B,n-1)]."""
        z, sdf = z.detach().float().contiguous(), sdf.detach().float().reshape(z.shape).contiguous()
        B, n = z.shape
        beta = beta.detach().float().reshape(-1).contiguous().to(z.device)
        if beta.numel() not in (1, B):
            raise ValueError("beta must be a scalar or one value per ray")
        bound = torch.empty(B, dtype=torch.float32, device=z.device)
        ds_out = torch.empty(B, n - 1, dtype=torch.float32, device=z.device) if want_d_star else None
        ds_in = None if d_star is None else d_star.detach().float().contiguous()
        L.check(self._lib.i2sdf_error_bound(L.ptr(z), L.ptr(sdf), B, n, L.ptr(beta), 0 if beta.numel() == 1 else 1, L.ptr(ds_in),
                                            L.ptr(ds_out), L.ptr(bound), L.stream_ptr()), "i2sdf_error_bound")
        return (bound, ds_out) if want_d_star else bound

    # -- full-image inference (N3) ---------------------------------------------------------------
    def render_image(self, flat_params, uv, pose, intrinsics, chunk, want_normal=True, want_z=False):
        """All chunks of one view in ONE library call (include/i2sdf.h: i2sdf_render_image).  uv (P,2); pose (4,4)|(7); K (4,4).
        Returns dict(rgb (P,3), depth (P), wsum (P,1), normal (P,3)|None, lmask (P,1)|None, z (P,n_z)|None, iters (n_chunks) int32)."""
        uv = uv.detach().to(torch.float32).reshape(-1, 2).contiguous()
        pose = pose.detach().to(torch.float32).contiguous()
        intrinsics = intrinsics.detach().to(torch.float32).reshape(4, 4).contiguous()
        quat = pose.numel() == 7
        if not quat and pose.numel() != 16:
            raise ValueError(f"one view: pose must be (4,4) or (7,), got {tuple(pose.shape)}")
        P, dev = uv.shape[0], uv.device
        chunk = int(max(1, min(chunk, max(P, 1))))
        n_ws = int(self._lib.i2sdf_render_image_workspace_floats(self._plan, C.byref(self._scfg), chunk))
        ws = getattr(self, "_img_ws", None)
        if ws is None or ws.numel() < n_ws or ws.device != dev:
            ws = torch.empty(n_ws, dtype=torch.float32, device=dev)
            self._img_ws = ws                                  # kept: a video / test set renders many views at the same chunk size
        e = lambda *s: torch.empty(*s, dtype=torch.float32, device=dev)
        o = {"rgb": e(P, 3), "depth": e(P), "wsum": e(P, 1), "normal": e(P, 3) if want_normal else None,
             "lmask": e(P, 1) if self.cfg.light is not None else None, "z": e(P, self.n_z) if want_z else None,
             "iters": torch.zeros((P + chunk - 1) // chunk, dtype=torch.int32, device=dev)}
        L.check(self._lib.i2sdf_render_image(self._plan, self._pk(), L.ptr(flat_params), C.byref(self._scfg), L.ptr(uv), L.ptr(pose),
                                             int(quat), L.ptr(intrinsics), P, chunk, L.ptr(self.t_lin), L.ptr(self.u_more), L.ptr(self.u_final),
                                             L.ptr(self.extra_tab), L.ptr(ws), L.ptr(o["rgb"]), L.ptr(o["depth"]), L.ptr(o["wsum"]),
                                             L.ptr(o["normal"]), L.ptr(o["lmask"]), L.ptr(o["z"]), L.ptr(o["iters"]), L.stream_ptr()),
                "i2sdf_render_image")
        return o

    def sdf_grid(self, x, y, z, rot=None, trans=None, order: int = L.GRID_ORDER_VOLUME, first: int = 0, count: Optional[int] = None,
                 chunk: int = 1 << 21):
        """SDF values on the grid x (nx) x y (ny) x z (nz) without materialising the points (include/i2sdf.h: i2sdf_sdf_grid).
        x, y, z: device fp32 vectors; rot (3,3) / trans (3,): host values, evaluated point = rot @ p + trans.  Returns (count,)."""
        x, y, z = (t.detach().to(torch.float32).contiguous() for t in (x, y, z))
        dev = x.device
        nx, ny, nz = x.numel(), y.numel(), z.numel()
        total = nx * ny * nz
        count = total - first if count is None else count
        chunk = int(max(128, min(chunk, max(count, 128))))
        ws = getattr(self, "_grid_ws", None)
        n_ws = int(self._lib.i2sdf_sdf_grid_workspace_floats(chunk))
        if ws is None or ws.numel() < n_ws or ws.device != dev:
            ws = torch.empty(n_ws, dtype=torch.float32, device=dev)
            self._grid_ws = ws
        out = torch.empty(count, dtype=torch.float32, device=dev)
        host = lambda v, n: None if v is None else (C.c_float * n)(*[float(a) for a in torch.as_tensor(v, dtype=torch.float32).reshape(-1).tolist()])
        r, t = host(rot, 9), host(trans, 3)
        L.check(self._lib.i2sdf_sdf_grid(self._plan, self._pk(), L.ptr(x), L.ptr(y), L.ptr(z), nx, ny, nz, int(order),
                                         C.cast(r, C.c_void_p) if r is not None else None, C.cast(t, C.c_void_p) if t is not None else None,
                                         int(first), int(count), L.ptr(out), L.ptr(ws), chunk, L.stream_ptr()), "i2sdf_sdf_grid")
        return out

    # -- light-mask head ---------------------------------------------------------------------------
    def light_forward(self, feat, M, save=True):
        Mp, dev = feat.shape[0], feat.device
        HL = self.cfg.light.hidden
        lm = torch.empty(M, dtype=torch.float32, device=dev)
        hl = self._ws(Mp, HL, device=dev) if save else None
        L.check(self._lib.i2sdf_light_forward(self._plan, self._pk(), L.ptr(feat), M, Mp, L.ptr(lm), L.ptr(hl), L.stream_ptr()),
                 "i2sdf_light_forward")
        return lm, hl

    def light_backward(self, lm, lm_bar, hl, M):
        Mp, dev = hl.shape[0], hl.device
        gal0 = self._ws(Mp, hl.shape[1], device=dev)
        gal_last = self._ws(Mp, 4, device=dev)
        L.check(self._lib.i2sdf_light_backward(self._plan, self._pk(), L.ptr(lm), L.ptr(lm_bar.contiguous()), L.ptr(hl), M, Mp,
                                                L.ptr(gal0), L.ptr(gal_last), L.stream_ptr()), "i2sdf_light_backward")
        return gal0, gal_last
