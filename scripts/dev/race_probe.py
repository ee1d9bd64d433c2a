"""Run-to-run determinism of the parameter gradients (same net, same inputs, given depths)."""
import os, sys, torch
sys.path.insert(0, "tests"); sys.path.insert(0, ".")
from oracle import i2sdf_oracle as orc
from helpers import camera_inputs, make_draws, make_gt, rel_max
from test_gpu_network import build, cuda
from i2sdf_amd import synthetic_conf, I2SDFLoss
B = int(os.environ.get("PB", "320"))
ocfg = orc.synthetic_cfg(False); ocfg.use_normal = True
sd = orc.perturb_params(orc.init_params(ocfg, seed=41), 0.03, seed=42); sd["density.beta"] = torch.tensor(0.05)
inp = camera_inputs(B, (0.0, 0.0, -2.0), seed=5); gt = make_gt(B)
cam, dirs, dn = orc.prepare_rays(inp["uv"], inp["pose"], inp["intrinsics"])
dr = make_draws(ocfg, B, n_row=128, seed=2)
z_all, z_eik = orc.sample_z_vals(sd, ocfg, dirs, cam, training=True, draws=dr, force_iters=1)
ref = {}
for mode in (True, False, True, False):
    conf = dict(synthetic_conf(False)); conf["bf16x3"] = mode
    net = build(conf, sd, train=True)
    eng = net._engine_for("cuda:0")
    c, d, n = eng.ray_setup(inp["uv"].cuda(), inp["pose"].cuda(), inp["intrinsics"].cuda())
    gs = []
    for rep in range(3):
        out = net.render(cuda(inp), c, d, n, z_all.cuda(), z_eik.cuda(), draws={"eik_pts": dr.eik_pts.cuda(), "nbr_off": dr.nbr_off.cuda()})
        losses = I2SDFLoss(eikonal_weight=0.1, smooth_weight=0.01, smooth_iter=None, depth_weight=0.1, normal_weight=0.05)(out, cuda(gt), 10)
        net.zero_grad(); losses["loss"].backward()
        gs.append({n_: p.grad.detach().cpu().clone() for n_, p in net.named_parameters() if p.grad is not None})
    rr = max((rel_max(gs[i][k], gs[0][k]), k) for i in (1, 2) for k in gs[0])
    if mode not in ref: ref[mode] = gs[0]
    vs_first = max((rel_max(gs[0][k], ref[mode][k]), k) for k in gs[0])
    x = max((rel_max(gs[0][k], ref[True][k]), k) for k in gs[0])
    print("bf16x3", mode, "B", B, "run-to-run", f"{rr[0]:.1e}", rr[1], "| vs first net of this mode", f"{vs_first[0]:.1e}", vs_first[1], "| vs first bf16x3 net", f"{x[0]:.1e}", x[1])
