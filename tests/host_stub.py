"""A CPU-constructible stand-in for the HIP render core UNDER the real module wiring -- test infrastructure, never a product path.

`HostStubNetwork` is `i2sdf_amd.I2SDFNetwork` (same constructor, same parameters as views of ONE flat fp32 buffer, same
`grad_sync` / `dp_state` hooks, same state_dict) whose forward/backward -- in the product one autograd node of HIP kernels
(`network._RenderFn`) -- is replaced by a tiny closed-form function of the flat parameter buffer evaluated with torch on the CPU.  Its
backward ends exactly like the product's: the gradient is written into one flat buffer, `net.grad_sync(gflat)` is called, and the
per-parameter gradients are returned as views of that buffer.  What it lets the CPU box test is everything AROUND the kernels on the
N > 1 path: `i2sdf_amd.dist.attach_data_parallel`, `no_sync`, `broadcast_parameters`, and bench.py's launch logic
(`bench.py --selftest-launch`)."""
import torch

from i2sdf_amd import I2SDFNetwork


class _StubRenderFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, net, uv, *params):
        flat = net._flat
        n = flat.numel()
        B = uv.shape[0] * uv.shape[1]
        x = uv.reshape(B, 2).to(torch.float32) * 1e-3
        # a smooth function of EVERY parameter (so every entry of the flat gradient is exercised) and of the ray
        phase = torch.linspace(0.0, 3.0, n, dtype=torch.float32)
        s = torch.sin(flat.detach() + phase)                                   # (n,)
        basis = torch.stack([s[0::3].sum(), s[1::3].sum(), s[2::3].sum()]) / n   # (3,)
        rgb = torch.sigmoid(x.sum(1, keepdim=True) + basis.unsqueeze(0))        # (B,3)
        ctx.net, ctx.phase, ctx.B = net, phase, B
        ctx.save_for_backward(rgb)
        return rgb

    @staticmethod
    def backward(ctx, g_rgb):
        net = ctx.net
        flat = net._flat
        n = flat.numel()
        (rgb,) = ctx.saved_tensors
        d_basis = (g_rgb * rgb * (1 - rgb)).sum(0)                               # (3,)
        gflat = torch.zeros_like(flat)
        c = torch.cos(flat.detach() + ctx.phase) / n
        for j in range(3):
            gflat[j::3] = c[j::3] * d_basis[j]
        if net.grad_sync is not None:                                          # the same hook call as network._RenderFn.backward
            net.grad_sync(gflat)
        grads = []
        for name, off, shape in net.layout.entries:
            cnt = 1
            for s_ in shape:
                cnt *= s_
            grads.append(gflat[off:off + cnt].view(shape))
        return (None, None) + tuple(grads)


class HostStubNetwork(I2SDFNetwork):
    def forward(self, input, predict_only=False, draws=None):
        self._ensure_flat()
        rgb = _StubRenderFn.apply(self, input["uv"], *self._param_list())
        return {"rgb_values": rgb}
