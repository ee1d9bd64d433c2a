"""GPU: the first steps of a training run on the PRODUCTION path beside the restatement, step by step (a smoke check since round 5).

The loop mirrored is model/trainer/recon.py:201-207,219-287 (Adam(lr 5e-4, eps 1e-15), one batch of rays per step, loss of
model/network/__init__.py:289-406, PSNR utils/rend_util.py:13-22).

  * HIP side  : `I2SDFNetwork` as a trainer runs it -- bf16x3 kernels, data-dependent sampler loop on the device, point ranges on
                their own streams, fused loss, `FusedAdam` (one launch).  The random draws of a step come from the library's own
                fused draws kernel (`i2sdf_training_draws`, Philox); they are taken out of the module call only so that the very
                same numbers can be handed to the other side.
  * oracle    : the fp32 restatement's torch ops (autograd double backward) as stock PyTorch-ROCm eager kernels on the same GPU,
                `torch.optim.Adam`, its own sampler decisions, from the same initial weights, batches and draws.

This training loop amplifies rounding noise -- Adam with eps = 1e-15 turns a gradient entry of noise magnitude into a full +-lr step, so
after ONE step two fp32 executions differ by 2 lr in some weights -- and after ~30 steps the per-batch PSNR of two executions of the
restatement ITSELF is 0.05 dB apart, after 50 steps +-0.5 ... 1.5 dB (rounds 3-4, scripts/ab/curve_probe.py).  So a single
trajectory can be held to "0.1 dB at every step" only while the trajectories are still comparable: steps 0..24 (measured 4e-4 dB at
step 20), loss within 1e-3 relative.  Rounds 3-4 additionally compared the rest of ONE 200-step trajectory with an envelope around the
restatement's own rounding-noise twins (bars 2.0 x the twins' RMS, 1.5 dB on the tail mean) -- a comparison without the power to resolve
0.1 dB.  Those bars are gone: the long-run claim is now tests/test_gpu_psnr_ensemble.py (16 members per arm, 300
steps, a learnable teacher-rendered target, |difference of mean tail PSNR| <= 0.1 dB + 2 SE with SE <= 0.1 dB)."""
import math

import pytest
import torch

from oracle import i2sdf_oracle as orc

pytestmark = pytest.mark.gpu
STEPS, B, LR = 60, 512, 5.0e-4
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


def test_first_steps_production_path_vs_restatement_step_by_step():
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
    # ---- restatement side (eager ROCm ops), from the same weights
    inits = [sd0]
    leaves = [{k: torch.nn.Parameter(v.clone().to(dev)) for k, v in sd.items()} for sd in inits]
    opts = [torch.optim.Adam(list(lv.values()), lr=LR, eps=1e-15) for lv in leaves]

    # the restatement's three runs do not depend on the weight-gradient mode of the library (tests/conftest.py runs this module once per
    # mode): the first parametrisation computes them, the second reuses curves and final weights
    import helpers
    cached = helpers._MEMO.get("first-steps curve of the restatement")
    psnr_h, loss_h, it_h = [], [], []
    psnr_o, loss_o = [[] for _ in leaves], [[] for _ in leaves]
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
        if cached is not None:
            continue
        dr = orc.Draws(strat_u=draws["strat_u"], cdf_u=draws["cdf_u"], extra_idx=draws["extra_idx"], eik_idx=draws["eik_idx"],
                       eik_pts=draws["eik_pts"], nbr_off=draws["nbr_off"])
        for r, (lv, opt) in enumerate(zip(leaves, opts)):
            cur = {k: p.detach() for k, p in lv.items()}
            o_out, o_losses, grads = orc.training_step_grads(cur, ocfg, inp, gt, lc, dr, step=step)
            opt.zero_grad(set_to_none=True)
            for k, p in lv.items():
                p.grad = grads[k].reshape(p.shape).clone()
            opt.step()
            psnr_o[r].append(float(orc.get_psnr(o_out["rgb_values"].detach(), gt["rgb"])))
            loss_o[r].append(float(o_losses["loss"].detach()))

    if cached is None:
        helpers._MEMO["first-steps curve of the restatement"] = (psnr_o, loss_o, [{k: p.detach().clone() for k, p in lv.items()} for lv in leaves])
    else:
        psnr_o, loss_o, leaves = cached
    A = psnr_o[0]
    last = lambda x: sum(x[-10:]) / 10.0
    early = max(abs(a - b) for a, b in zip(A[:25], psnr_h[:25]))
    early_loss = max(abs(a - b) / abs(a) for a, b in zip(loss_o[0][:25], loss_h[:25]))
    print(f"PSNR restatement {A[0]:.3f} -> mean of steps {STEPS - 10}..{STEPS - 1} {last(A):.3f} dB, production {psnr_h[0]:.3f} -> {last(psnr_h):.3f} dB; "
          f"steps 0..24: max |dPSNR| {early:.2e} dB, max rel loss diff {early_loss:.2e}; step 40: {abs(A[40] - psnr_h[40]):.3f} dB; "
          f"sampler iterations seen {sorted(set(it_h))}")
    assert last(A) - A[0] > 0.5 and last(psnr_h) - psnr_h[0] > 0.5, "the run must actually train"
    assert early < 0.1, [(i, a, b) for i, (a, b) in enumerate(zip(A[:25], psnr_h[:25])) if abs(a - b) >= 0.1][:5]
    assert early_loss < 1e-3, early_loss
    assert all(math.isfinite(x) for x in psnr_h + loss_h)
