"""Thin, stateless-as-possible Python driver over the C ABI: owns the plan and the packed-weight buffer,
allocates outputs/workspaces with torch (device memory + streams are torch's job; the arithmetic is the library's)."""
from __future__ import annotations

import ctypes as C
from typing import Optional

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
        self.F = cfg.feature_size

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
        L.check(self._lib.i2sdf_pack_weights(self._plan, L.ptr(flat_params), L.ptr(self.packed), L.stream_ptr()), "i2sdf_pack_weights")

    # -- SDF queries ---------------------------------------------------------------------------
    def sdf_forward(self, points: torch.Tensor, want_features: bool = False):
        """ImplicitNetwork.forward without grad: returns sdf (M,1) [, feature (M,F)]."""
        pts = points.detach().to(torch.float32).contiguous()
        M = pts.shape[0]
        sdf = torch.empty(M, 1, dtype=torch.float32, device=pts.device)
        feat = torch.empty(M, self.F, dtype=torch.float32, device=pts.device) if want_features else None
        L.check(self._lib.i2sdf_sdf_forward(self._plan, L.ptr(self.packed), L.ptr(pts), M, L.ptr(sdf), L.ptr(feat), self.F,
                                            L.stream_ptr()), "i2sdf_sdf_forward")
        return (sdf, feat) if want_features else sdf
