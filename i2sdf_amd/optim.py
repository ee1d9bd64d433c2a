"""FusedAdam: the reference's optimizer (`optim.Adam(model.get_param_groups(lr), eps=1e-15)`, model/trainer/recon.py:201-203)
as ONE HIP launch per parameter group over the flat parameter buffer of `I2SDFNetwork` (include/i2sdf.h: i2sdf_adam_step).

It is a `torch.optim.Optimizer`: same constructor form, `param_groups` / `state` / `state_dict()` layout as torch.optim.Adam
(`step`, `exp_avg`, `exp_avg_sq` per parameter), so LR schedulers (`ExponentialLR`, :204-206) and Lightning checkpoints work
unchanged.  The per-parameter state tensors are views into two flat moment buffers; parameters and their gradients are views into
the module's flat buffers (network.py), so a group whose tensors are contiguous in memory is updated by one launch -- otherwise
the same kernel runs once per tensor.  There is no eager-torch arithmetic here."""
from __future__ import annotations

import torch

from . import lib as L


class FusedAdam(torch.optim.Optimizer):
    def __init__(self, params, lr=1e-3, betas=(0.9, 0.999), eps=1e-8, weight_decay=0.0):
        if isinstance(params, torch.nn.Module):
            params = params.parameters()
        if lr < 0 or eps < 0 or not (0 <= betas[0] < 1 and 0 <= betas[1] < 1) or weight_decay < 0:
            raise ValueError("invalid Adam hyper-parameters")
        super().__init__(params, dict(lr=lr, betas=betas, eps=eps, weight_decay=weight_decay))
        self._flat_state = {}        # group index -> (exp_avg flat, exp_avg_sq flat, [numel])
        self.grad_scale = 1.0        # multiplies every gradient inside the kernel (1/world after a summed all-reduce)

    @staticmethod
    def _contiguous_run(tensors):
        """data_ptr of the first tensor if the tensors tile one contiguous fp32 range in order, else None."""
        base = tensors[0].data_ptr()
        off = 0
        for t in tensors:
            if t.dtype != torch.float32 or not t.is_contiguous() or t.data_ptr() != base + 4 * off:
                return None
            off += t.numel()
        return base

    def _moments(self, gi, params):
        """Flat moment buffers of group gi; per-parameter state entries are views into them (re-adopted after load_state_dict)."""
        n = sum(p.numel() for p in params)
        fs = self._flat_state.get(gi)
        if fs is None or fs[0].numel() != n or fs[0].device != params[0].device:
            with torch.inference_mode(False), torch.no_grad():
                fs = (torch.zeros(n, dtype=torch.float32, device=params[0].device), torch.zeros(n, dtype=torch.float32, device=params[0].device))
            self._flat_state[gi] = fs
        off = 0
        for p in params:
            k = p.numel()
            if p.grad is None and p not in self.state:      # torch.optim.Adam creates state lazily, for parameters that have a gradient
                off += k
                continue
            st = self.state[p]
            for name, flat in (("exp_avg", fs[0]), ("exp_avg_sq", fs[1])):
                view = flat[off:off + k].view(p.shape)
                cur = st.get(name)
                if cur is None:
                    st[name] = view
                elif cur.data_ptr() != view.data_ptr():          # loaded from a checkpoint (or buffers rebuilt): copy in, re-point
                    with torch.no_grad():
                        view.copy_(cur.to(view.device, torch.float32))
                    st[name] = view
            if "step" not in st:
                st["step"] = torch.zeros((), dtype=torch.float32)
            off += k
        return fs

    @torch.no_grad()
    def step(self, closure=None):
        loss = None
        if closure is not None:
            with torch.enable_grad():
                loss = closure()
        lib = L.load()
        for gi, group in enumerate(self.param_groups):
            if group.get("amsgrad") or group.get("maximize"):      # e.g. a torch.optim.Adam checkpoint's param_groups
                raise RuntimeError("FusedAdam implements plain Adam: amsgrad / maximize are not supported")
            params = [p for p in group["params"] if p.grad is not None]
            if not params:
                continue
            # (parameters without a gradient are skipped, exactly like torch.optim.Adam; they break the contiguous run below)
            for p in params:
                if not p.is_cuda or p.dtype != torch.float32:
                    raise RuntimeError("FusedAdam updates fp32 parameters on a cuda/HIP device only (no CPU fallback)")
                if p.grad.is_sparse:
                    raise RuntimeError("FusedAdam does not support sparse gradients")
            all_params = list(group["params"])
            m_flat, v_flat = self._moments(gi, all_params)
            beta1, beta2 = group["betas"]
            steps = []
            for p in params:
                st = self.state[p]
                st["step"] = st["step"] + 1
                steps.append(int(st["step"]))
            with torch.cuda.device(params[0].device):
                stream = L.stream_ptr()
                same_step = all(s == steps[0] for s in steps)
                pbase = self._contiguous_run(params) if (same_step and len(params) == len(all_params)) else None
                grads = [p.grad for p in params]
                gbase = self._contiguous_run(grads) if pbase is not None else None
                if pbase is not None and gbase is not None:
                    n = m_flat.numel()
                    L.check(lib.i2sdf_adam_step(pbase, gbase, L.ptr(m_flat), L.ptr(v_flat), n, group["lr"], beta1, beta2, group["eps"],
                                                group["weight_decay"], steps[0], self.grad_scale, stream), "i2sdf_adam_step")
                else:
                    for p, s in zip(params, steps):
                        st = self.state[p]
                        if not p.is_contiguous():
                            raise RuntimeError("FusedAdam needs contiguous parameters (the kernel updates them in place through a raw pointer)")
                        g = p.grad if p.grad.is_contiguous() else p.grad.contiguous()
                        L.check(lib.i2sdf_adam_step(L.ptr(p.data), L.ptr(g), L.ptr(st["exp_avg"]), L.ptr(st["exp_avg_sq"]), p.numel(), group["lr"],
                                                    beta1, beta2, group["eps"], group["weight_decay"], s, self.grad_scale, stream),
                                "i2sdf_adam_step")
        return loss
