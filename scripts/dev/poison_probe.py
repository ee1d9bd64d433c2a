"""Every torch.empty float buffer of the engine is filled with NaN: a kernel that reads rows nobody wrote shows up as NaN gradients."""
import os, sys, torch
sys.path.insert(0, "tests"); sys.path.insert(0, ".")
_empty = torch.empty
def poisoned(*a, **k):
    t = _empty(*a, **k)
    if t.is_floating_point() and t.is_cuda: t.fill_(float("nan"))
    return t
torch.empty = poisoned
from oracle import i2sdf_oracle as orc
from helpers import camera_inputs, make_draws, make_gt
from test_gpu_network import build, cuda
from i2sdf_amd import synthetic_conf, I2SDFLoss
B = int(os.environ.get("PB", "320"))
for mode in (True, False):
    ocfg = orc.synthetic_cfg(False); ocfg.use_normal = True
    sd = orc.perturb_params(orc.init_params(ocfg, seed=41), 0.03, seed=42); sd["density.beta"] = torch.tensor(0.05)
    conf = dict(synthetic_conf(False)); conf["bf16x3"] = mode
    net = build(conf, sd, train=True)
    inp = camera_inputs(B, (0.0, 0.0, -2.0), seed=5); gt = make_gt(B)
    out = net(cuda(inp))
    losses = I2SDFLoss(eikonal_weight=0.1, smooth_weight=0.01, smooth_iter=None, depth_weight=0.1, normal_weight=0.05)(out, cuda(gt), 10)
    net.zero_grad(); losses["loss"].backward()
    bad = [n for n, p in net.named_parameters() if p.grad is not None and not torch.isfinite(p.grad).all()]
    print("bf16x3", mode, "B", B, "loss", float(losses["loss"]), "outputs finite", {k: bool(torch.isfinite(v).all()) for k, v in out.items()}, "non-finite grads:", bad)
