#!/usr/bin/env python3
"""per-tensor weight differences of tests/test_gpu_training_parity.py::test_training_curves_match_oracle (light=True) after the 24 steps"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch
import test_gpu_training_parity as T
from oracle import i2sdf_oracle as orc
from helpers import camera_inputs, make_gt, make_draws
from i2sdf_amd import plumbing_conf, I2SDFLoss
light = True
ocfg = orc.plumbing_cfg(skip=True, light=light); ocfg.use_normal = True
sd = orc.init_params(ocfg, seed=3); sd["density.beta"] = torch.tensor(0.05)
lkw = dict(eikonal_weight=0.1, smooth_weight=0.01, smooth_iter=8, depth_weight=0.1, normal_weight=0.05, light_mask_weight=0.5)
lc = orc.LossCfg(**lkw)
n_row = ocfg.sampler.N_samples_eval + ocfg.sampler.N_samples
leaves = {k: torch.nn.Parameter(v.clone()) for k, v in sd.items()}
opt_o = torch.optim.Adam(list(leaves.values()), lr=T.LR, eps=1e-15)
for step in range(T.STEPS):
    inp = camera_inputs(T.B, (0.0, 0.0, -2.0), W=32, H=32, f=30.0, seed=100 + step); gt = make_gt(T.B, seed=step, light=light)
    dr = make_draws(ocfg, T.B, n_row=n_row, seed=1000 + step)
    out, losses, grads = orc.training_step_grads({k: p.detach() for k, p in leaves.items()}, ocfg, inp, gt, lc, dr, step=step)
    opt_o.zero_grad()
    for k, p in leaves.items():
        p.grad = grads[k].reshape(p.shape).clone()
    opt_o.step()
for fused in ("1", "0"):
    os.environ["I2SDF_FUSED_RENDER_LOSS"] = fused
    net = T.build(plumbing_conf(skip=True, light=light), sd, train=True)
    loss_fn = I2SDFLoss(**lkw)
    opt_h = torch.optim.Adam(net.get_param_groups(T.LR), eps=1e-15)
    for step in range(T.STEPS):
        inp = camera_inputs(T.B, (0.0, 0.0, -2.0), W=32, H=32, f=30.0, seed=100 + step); gt = make_gt(T.B, seed=step, light=light)
        dr = make_draws(ocfg, T.B, n_row=n_row, seed=1000 + step)
        out = net(T.cuda(inp), draws=T._draws_dict(dr))
        l = loss_fn(out, T.cuda(gt), step)
        opt_h.zero_grad(); l["loss"].backward(); opt_h.step()
    got = net.state_dict()
    rows = []
    for k in leaves:
        d = (got[k].cpu().reshape(-1).double() - leaves[k].detach().reshape(-1).double()).abs()
        rows.append((float(d.mean()) / T.LR, float(d.max()) / T.LR, d.numel(), k))
    rows.sort(reverse=True)
    print(f"fused={fused}: worst per-tensor mean |diff| / LR:")
    for r in rows[:6]:
        print(f"   mean {r[0]:.3f} LR  max {r[1]:.2f} LR  numel {r[2]:6d}  {r[3]}")
