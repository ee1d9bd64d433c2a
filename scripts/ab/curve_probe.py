"""Where does the production training path leave the restatement's curve?  (tests/test_gpu_training_curve_full.py diagnostics)
usage: python scripts/ab/curve_probe.py STEPS parts fused_adam(0/1) [rays]"""
import sys, os
sys.path.insert(0, "tests"); sys.path.insert(0, ".")
import torch
from oracle import i2sdf_oracle as orc
import test_gpu_training_curve_full as T

STEPS, parts, fused = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3])
B = int(sys.argv[4]) if len(sys.argv) > 4 else T.B
from i2sdf_amd import I2SDFNetwork, I2SDFLoss, FusedAdam, synthetic_conf
dev = torch.device("cuda:0")
conf = dict(synthetic_conf(False)); conf["use_normal"] = True
ocfg = orc.synthetic_cfg(False); ocfg.use_normal = True
sd0 = orc.init_params(ocfg, seed=11); sd0["density.beta"] = torch.tensor(0.05)
lkw = dict(eikonal_weight=0.1, smooth_weight=0.01, smooth_iter=100, depth_weight=0.1, normal_weight=0.05)
lc = orc.LossCfg(**lkw)
net = I2SDFNetwork(conf); net.load_state_dict(sd0); net = net.to(dev).train()
loss_fn = I2SDFLoss(**lkw)
eng = net._engine_for(dev); eng.set_parts(parts)
opt_h = FusedAdam(net, lr=T.LR, eps=1e-15) if fused else torch.optim.Adam(net.get_param_groups(T.LR), eps=1e-15)
leaves = {k: torch.nn.Parameter(v.clone().to(dev)) for k, v in sd0.items()}
opt_o = torch.optim.Adam(list(leaves.values()), lr=T.LR, eps=1e-15)
# a second restatement run whose initial weights differ by fp32 rounding noise (relative 1e-7): how far do the restatement's OWN curves move?
gN = torch.Generator().manual_seed(99)
leaves2 = {k: torch.nn.Parameter((v * (1 + 1e-7 * torch.randn(v.shape, generator=gN))).to(dev)) for k, v in sd0.items()}
opt_o2 = torch.optim.Adam(list(leaves2.values()), lr=T.LR, eps=1e-15)
names = [n for n, _ in net.named_parameters()]
for step in range(STEPS):
    inp, gt = T._batch(step, dev, B=B)
    draws = eng.training_draws(B, 7_000_000 + step, dev, net.scene_bounding_sphere, want_eik=True)
    out = net(inp, draws=draws)
    losses = loss_fn(out, gt, step)
    opt_h.zero_grad(set_to_none=True)
    losses["loss"].backward()
    gh = {n: p.grad.detach().clone() for n, p in net.named_parameters()}
    dr = orc.Draws(strat_u=draws["strat_u"], cdf_u=draws["cdf_u"], extra_idx=draws["extra_idx"], eik_idx=draws["eik_idx"],
                   eik_pts=draws["eik_pts"], nbr_off=draws["nbr_off"])
    cur = {k: p.detach() for k, p in leaves.items()}
    tr_it = None
    o_out, o_losses, grads = orc.training_step_grads(cur, ocfg, inp, gt, lc, dr, step=step)
    # gradient agreement AT THE ORACLE'S WEIGHTS is only meaningful while the weights agree: report both
    wdiff = max(float((dict(net.named_parameters())[k].detach() - leaves[k].detach()).abs().max()) for k in names)
    gerr = max(float((gh[k] - grads[k].reshape(gh[k].shape)).abs().max() / grads[k].abs().max().clamp_min(1e-30)) for k in names)
    ph = float(orc.get_psnr(out["rgb_values"].detach(), gt["rgb"])); po = float(orc.get_psnr(o_out["rgb_values"].detach(), gt["rgb"]))
    cur2 = {k: p.detach() for k, p in leaves2.items()}
    o2_out, o2_losses, grads2 = orc.training_step_grads(cur2, ocfg, inp, gt, lc, dr, step=step)
    po2 = float(orc.get_psnr(o2_out["rgb_values"].detach(), gt["rgb"]))
    opt_o2.zero_grad(set_to_none=True)
    for k, p in leaves2.items():
        p.grad = grads2[k].reshape(p.shape).clone()
    opt_o2.step()
    if step < 4 or step % 10 == 0:
        print(f"      restatement with 1e-7 weight noise: psnr {po2:.4f}  d vs O {po2-po:+.4f}")
    if step < 4 or step % 10 == 0:
        print(f"step {step:3d} it {int(net.last_sampler_iters.item())} psnr H {ph:.4f} O {po:.4f} d {ph-po:+.4f}  loss H {float(losses['loss']):.5f} O {float(o_losses['loss']):.5f}"
              f"  |w_H-w_O|max {wdiff:.2e}  grad relerr {gerr:.2e}", flush=True)
    opt_h.step()
    opt_o.zero_grad(set_to_none=True)
    for k, p in leaves.items():
        p.grad = grads[k].reshape(p.shape).clone()
    opt_o.step()
