"""HIP training step vs fp64 oracle at several batch sizes (same construction as tests/test_gpu_network.py::test_train_step_given_depths_full_size)."""
import sys, torch
sys.path.insert(0, "tests"); sys.path.insert(0, ".")
from oracle import i2sdf_oracle as orc
from helpers import camera_inputs, make_draws, make_gt, rel_max
from test_gpu_network import build, cuda
from i2sdf_amd import synthetic_conf, I2SDFLoss
D = torch.float64
import os
for B, seeds in ((int(os.environ.get('PB', '320')), (41, 42, 5, 2)),):
    ocfg = orc.synthetic_cfg(False); ocfg.use_normal = True
    sd = orc.perturb_params(orc.init_params(ocfg, seed=seeds[0]), 0.03, seed=seeds[1]); sd["density.beta"] = torch.tensor(0.05)
    net = build(synthetic_conf(False), sd, train=True)
    inp = camera_inputs(B, (0.0, 0.0, -2.0), seed=seeds[2]); gt = make_gt(B)
    cam, dirs, dn = orc.prepare_rays(inp["uv"], inp["pose"], inp["intrinsics"])
    dr = make_draws(ocfg, B, n_row=128, seed=seeds[3])
    z_all, z_eik = orc.sample_z_vals(sd, ocfg, dirs, cam, training=True, draws=dr, force_iters=1)
    kw = dict(eikonal_weight=0.1, smooth_weight=0.01, smooth_iter=None, depth_weight=0.1, normal_weight=0.05)
    d64 = orc.Draws(eik_pts=dr.eik_pts.to(D), nbr_off=dr.nbr_off.to(D))
    gt64 = {k: (v.to(D) if v.dtype.is_floating_point else v) for k, v in gt.items()}
    ref_out, ref_loss, ref_g = orc.training_step_grads({k: v.to(D) for k, v in sd.items()}, ocfg, {k: v.to(D) for k, v in inp.items()}, gt64,
                                                       orc.LossCfg(**kw), d64, step=10, z_override=(z_all.to(D), z_eik.to(D)))
    eng = net._engine_for("cuda:0")
    c, d, n = eng.ray_setup(inp["uv"].cuda(), inp["pose"].cuda(), inp["intrinsics"].cuda())
    out = net.render(cuda(inp), c, d, n, z_all.cuda(), z_eik.cuda(), draws={"eik_pts": dr.eik_pts.cuda(), "nbr_off": dr.nbr_off.cuda()})
    losses = I2SDFLoss(**kw)(out, cuda(gt), 10)
    net.zero_grad(); losses["loss"].backward()
    eo = {k: rel_max(out[k].detach().cpu(), ref_out[k]) for k in ("rgb_values", "depth_values", "weight_sum", "grad_theta", "diff_norm")}
    eg = sorted(((rel_max((p.grad if p.grad is not None else torch.zeros_like(p)).cpu(), ref_g[n_]), n_) for n_, p in net.named_parameters()), reverse=True)
    print(B, seeds, "loss", float(losses["loss"]), float(ref_loss["loss"]), {k: f"{v:.1e}" for k, v in eo.items()}, "min diff_norm", float(ref_out["diff_norm"].min()),
          "worst grads", [(f"{e:.1e}", n_) for e, n_ in eg[:8]])
