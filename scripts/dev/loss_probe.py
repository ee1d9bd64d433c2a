import sys, torch
sys.path.insert(0, "tests"); sys.path.insert(0, ".")
from oracle import i2sdf_oracle as orc
from i2sdf_amd import I2SDFLoss
for B in (50, 256, 257, 400, 1024):
    g = torch.Generator().manual_seed(B)
    out = {"rgb_values": torch.rand(B, 3, generator=g), "depth_values": torch.rand(B, generator=g) * 3, "weight_sum": torch.rand(B, 1, generator=g),
           "grad_theta": torch.randn(2 * B, 3, generator=g), "diff_norm": torch.rand(B, generator=g),
           "normal_values": torch.nn.functional.normalize(torch.randn(B, 3, generator=g), dim=1)}
    gt = {"rgb": torch.rand(B, 3, generator=g), "depth": torch.rand(B, generator=g) * 3, "depth_mask": torch.rand(B, generator=g) > 0.3,
          "normal": torch.nn.functional.normalize(torch.randn(B, 3, generator=g), dim=1), "normal_mask": torch.rand(B, generator=g) > 0.3}
    kw = dict(eikonal_weight=0.1, smooth_weight=0.01, smooth_iter=None, depth_weight=0.1, normal_weight=0.05)
    o64 = {k: v.double().requires_grad_(True) for k, v in out.items()}
    l64 = orc.i2sdf_loss(o64, {k: (v.double() if v.dtype.is_floating_point else v) for k, v in gt.items()}, orc.LossCfg(**kw), 10)
    g64 = torch.autograd.grad(l64["loss"], list(o64.values()), allow_unused=True)
    oc = {k: v.cuda().requires_grad_(True) for k, v in out.items()}
    lh = I2SDFLoss(**kw)(oc, {k: v.cuda() for k, v in gt.items()}, 10)
    gh = torch.autograd.grad(lh["loss"], list(oc.values()), allow_unused=True)
    errs = {}
    for k, a, b in zip(out, gh, g64):
        if b is None: continue
        a = a.cpu().double() if a is not None else torch.zeros_like(b)
        errs[k] = float((a - b).abs().max() / b.abs().max().clamp_min(1e-30))
    print(B, "loss", float(lh["loss"]), float(l64["loss"]), {k: f"{v:.1e}" for k, v in errs.items()})
