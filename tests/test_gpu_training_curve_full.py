"""GPU: PSNR / loss curves of the synthetic.yml networks over 200 training steps on the PRODUCTION path, beside the restatement.

The loop mirrored is model/trainer/recon.py:201-207,219-287 (Adam(lr 5e-4, eps 1e-15), one batch of rays per step, loss of
model/network/__init__.py:289-406, PSNR utils/rend_util.py:13-22).

  * HIP side  : `I2SDFNetwork` as a trainer runs it -- bf16x3 kernels, data-dependent sampler loop on the device, point ranges on
                their own streams, fused loss, `FusedAdam` (one launch).  The random draws of a step come from the library's own
                fused draws kernel (`i2sdf_training_draws`, Philox); they are taken out of the module call only so that the very
                same numbers can be handed to the other side.
  * oracle    : the fp32 restatement's torch ops (autograd double backward) as stock PyTorch-ROCm eager kernels on the same GPU,
                `torch.optim.Adam`, its own sampler decisions.

Both start from identical weights and see identical batches / draws.  The scene is learnable (a shaded sphere: colour a smooth
function of the pixel, analytic depth and normals), so the curves actually move.  Bars: PSNR of the rendered batch
within 0.1 dB at EVERY step, total loss within 1e-2 relative, and the held-out PSNR of the two trained weight sets within 0.1 dB."""
import math

import pytest
import torch

from oracle import i2sdf_oracle as orc

pytestmark = pytest.mark.gpu
STEPS, B, LR = 200, 512, 5.0e-4
W, H, F0 = 640, 480, 600.0


def _batch(step, dev, B=B):
    """Rays of camera (ii) through random pixels + the targets of a unit sphere at the origin seen from (0,0,-2)."""
    g = torch.Generator().manual_seed(5000 + step)
    px = torch.stack([torch.randint(0, W, (B,), generator=g), torch.randint(0, H, (B,), generator=g)], -1).float()
    K = torch.eye(4); K[0, 0] = K[1, 1] = F0; K[0, 2], K[1, 2] = W / 2, H / 2
    pose = torch.eye(4); pose[:3, 3] = torch.tensor([0.0, 0.0, -2.0])
    inp = {"uv": px.reshape(B, 1, 2), "intrinsics": K.repeat(B, 1, 1), "pose": pose.repeat(B, 1, 1)}
    d = torch.stack([(px[:, 0] - W / 2) / F0, (px[:, 1] - H / 2) / F0, torch.ones(B)], -1)
    dn = d.norm(dim=1, keepdim=True)
    d = d / dn
    o = torch.tensor([0.0, 0.0, -2.0])
    b = (d * o).sum(-1)
    disc = b * b - (o.dot(o) - 1.0)
    hit = disc > 0
    t = (-b - disc.clamp_min(0).sqrt())
    n = torch.nn.functional.normalize(o + t.unsqueeze(-1) * d, dim=1)
    u, v = px[:, 0] / W, px[:, 1] / H
    rgb = torch.stack([0.5 + 0.4 * torch.sin(6.0 * u + 1.0), 0.5 + 0.4 * torch.sin(5.0 * v + 2.0), 0.5 + 0.4 * torch.cos(4.0 * (u + v))], -1)
    rgb = torch.where(hit.unsqueeze(-1), rgb * (0.6 + 0.4 * (-n[:, 2:3]).clamp(0, 1)), torch.full_like(rgb, 0.1))
    gt = {"rgb": rgb, "depth": torch.where(hit, t * d[:, 2], torch.zeros_like(t)), "depth_mask": hit.clone(),
          "normal": torch.where(hit.unsqueeze(-1), n, torch.tensor([0.0, 0.0, -1.0]).expand(B, 3)), "normal_mask": hit.clone()}
    mv = lambda x: {k: t_.to(dev) for k, t_ in x.items()}
    return mv(inp), mv(gt)


def test_200_step_curves_production_path_vs_restatement():
    from i2sdf_amd import I2SDFNetwork, I2SDFLoss, FusedAdam, synthetic_conf
    dev = torch.device("cuda:0")
    conf = dict(synthetic_conf(False))
    conf["use_normal"] = True
    ocfg = orc.synthetic_cfg(False)
    ocfg.use_normal = True
    sd0 = orc.init_params(ocfg, seed=11)
    sd0["density.beta"] = torch.tensor(0.05)
    lkw = dict(eikonal_weight=0.1, smooth_weight=0.01, smooth_iter=100, depth_weight=0.1, normal_weight=0.05)   # smooth term on from step 101
    lc = orc.LossCfg(**lkw)

    # ---- production side
    net = I2SDFNetwork(conf)
    net.load_state_dict(sd0)
    net = net.to(dev).train()
    assert net.fused_draws and net.force_iters == 0
    loss_fn = I2SDFLoss(**lkw)
    opt_h = FusedAdam(net, lr=LR, eps=1e-15)
    eng = net._engine_for(dev)
    assert eng.train_forward_bf16x3 and eng.sdf_backward_bf16x3 and eng.wgrad_bf16x3 and eng.rgb_bf16x3 and eng.sdf_forward_bf16x3
    # ---- restatement side (eager ROCm ops)
    leaves = {k: torch.nn.Parameter(v.clone().to(dev)) for k, v in sd0.items()}
    opt_o = torch.optim.Adam(list(leaves.values()), lr=LR, eps=1e-15)

    psnr_h, psnr_o, loss_h, loss_o, it_h = [], [], [], [], []
    for step in range(STEPS):
        inp, gt = _batch(step, dev)
        draws = eng.training_draws(B, 7_000_000 + step, dev, net.scene_bounding_sphere, want_eik=True)        # the fused draws kernel
        out = net(inp, draws=draws)
        losses = loss_fn(out, gt, step)
        opt_h.zero_grad(set_to_none=True)
        losses["loss"].backward()
        opt_h.step()
        psnr_h.append(float(orc.get_psnr(out["rgb_values"].detach(), gt["rgb"])))
        loss_h.append(float(losses["loss"].detach()))
        it_h.append(int(net.last_sampler_iters.item()))

        dr = orc.Draws(strat_u=draws["strat_u"], cdf_u=draws["cdf_u"], extra_idx=draws["extra_idx"], eik_idx=draws["eik_idx"],
                       eik_pts=draws["eik_pts"], nbr_off=draws["nbr_off"])
        cur = {k: p.detach() for k, p in leaves.items()}
        o_out, o_losses, grads = orc.training_step_grads(cur, ocfg, inp, gt, lc, dr, step=step)
        opt_o.zero_grad(set_to_none=True)
        for k, p in leaves.items():
            p.grad = grads[k].reshape(p.shape).clone()
        opt_o.step()
        psnr_o.append(float(orc.get_psnr(o_out["rgb_values"].detach(), gt["rgb"])))
        loss_o.append(float(o_losses["loss"].detach()))

    dps = max(abs(a - b) for a, b in zip(psnr_o, psnr_h))
    dl = max(abs(a - b) / abs(a) for a, b in zip(loss_o, loss_h))
    print(f"PSNR restatement {psnr_o[0]:.3f} -> {psnr_o[-1]:.3f} dB, production {psnr_h[0]:.3f} -> {psnr_h[-1]:.3f} dB, max |dPSNR| over {STEPS} steps "
          f"{dps:.2e} dB; loss {loss_o[0]:.4f} -> {loss_o[-1]:.4f}, max rel diff {dl:.2e}; sampler iterations seen {sorted(set(it_h))}")
    assert psnr_o[-1] - psnr_o[0] > 1.0, "the run must actually train (PSNR should rise by more than 1 dB)"
    assert dps < 0.1, [(i, a, b) for i, (a, b) in enumerate(zip(psnr_o, psnr_h)) if abs(a - b) >= 0.1][:5]
    assert dl < 1e-2, [(i, a, b) for i, (a, b) in enumerate(zip(loss_o, loss_h)) if abs(a - b) / abs(a) >= 1e-2][:5]

    # ---- held-out PSNR of the two trained weight sets, both rendered by the library's eval path (same renderer, different weights)
    ref_net = I2SDFNetwork(conf)
    ref_net.load_state_dict({k: p.detach().cpu() for k, p in leaves.items()})
    ref_net = ref_net.to(dev).eval()
    net.eval()
    vin, vgt = _batch(10_000, dev, B=4096)
    with torch.no_grad():
        pa = float(orc.get_psnr(net(vin)["rgb_values"], vgt["rgb"]))
        pb = float(orc.get_psnr(ref_net(vin)["rgb_values"], vgt["rgb"]))
    print(f"held-out PSNR (4096 rays): production-trained {pa:.3f} dB, restatement-trained {pb:.3f} dB")
    assert math.isfinite(pa) and abs(pa - pb) < 0.1, (pa, pb)
