"""GPU: PSNR / loss curves of the synthetic.yml networks over 200 training steps on the PRODUCTION path, beside the restatement.

The loop mirrored is model/trainer/recon.py:201-207,219-287 (Adam(lr 5e-4, eps 1e-15), one batch of rays per step, loss of
model/network/__init__.py:289-406, PSNR utils/rend_util.py:13-22).

  * HIP side  : `I2SDFNetwork` as a trainer runs it -- bf16x3 kernels, data-dependent sampler loop on the device, point ranges on
                their own streams, fused loss, `FusedAdam` (one launch).  The random draws of a step come from the library's own
                fused draws kernel (`i2sdf_training_draws`, Philox); they are taken out of the module call only so that the very
                same numbers can be handed to the other side.
  * oracle    : the fp32 restatement's torch ops (autograd double backward) as stock PyTorch-ROCm eager kernels on the same GPU,
                `torch.optim.Adam`, its own sampler decisions -- run THREE times: A from the same initial weights as the HIP side,
                A' and A'' from weights that differ from them by fp32 rounding noise (relative 1e-7).

All runs see identical batches / draws.  The scene is learnable (a shaded sphere: colour a smooth function of the pixel, analytic depth
and normals), so the curves actually move (8.6 -> ~24 dB).

What can be asked of such a comparison was measured first (scripts/ab/curve_probe.py): this training loop amplifies rounding
noise -- Adam with eps = 1e-15 turns a gradient entry of noise magnitude into a full +-lr step, so after ONE step two fp32 executions
differ by 2 lr in some weights -- and after ~30 steps the per-batch PSNR of the restatement's own twins A / A' is 0.05 dB apart, after 50
steps +-0.5 ... 1.5 dB (the curve itself fluctuates by +-1 dB from batch to batch).  "Within 0.1 dB at every one of 200 steps" is therefore
not a property any two fp32 runs of the reference have, the reference on two different GPUs included.  The bars:
  1. while the trajectories are still comparable (steps 0..24) the production PSNR is within 0.1 dB of A at every step (measured 4e-4 dB
     at step 20) and the loss within 1e-3 relative;
  2. over the whole run the production curve is no further from A than rounding-noise-level changes move such a run: RMS PSNR difference
     over steps 50..199 <= 2 x the twins' + 0.1 dB, and the mean PSNR of the last 50 steps within 0.1 dB + 2 x the twins' spread of that
     mean, at least 1.5 dB.  Two twins are a small sample of that distribution, and round 4 measured how small: seven trajectories from
     the same start -- the restatement and its two twins, and four builds of the library that differ only in the arithmetic of the
     weight-gradient GEMMs (three bf16 planes everywhere / two planes in the 256x256 blocks with the narrow blocks in three planes, in
     two planes, in fp32-input MFMA: gradient differences of 1e-6 relative) -- have last-50-step means of 25.13, 25.47, 25.36 | 25.58,
     25.41, 24.15, 24.57 dB (standard deviation 0.5 dB, range 1.4 dB) and RMS distances to A of 1.02 | 1.06, 0.82, 1.76, 1.51 dB, in no
     order of arithmetic accuracy (the fp32-input variant is the second furthest); each is bit-reproducible run to run.  The round-3
     bars (1.5 x, 0.5 dB) were inside that spread;
  3. the held-out PSNR (4096 fresh rays, the library's eval renderer) of the production-trained weights lies within the same envelope
     around the restatement-trained ones (at least 1 dB: one evaluation of one set of final weights)."""
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
    # ---- restatement side (eager ROCm ops): A from the same weights, two twins from weights with fp32 rounding noise
    gN = torch.Generator().manual_seed(99)
    inits = [sd0] + [{k: v * (1 + 1e-7 * torch.randn(v.shape, generator=gN)) for k, v in sd0.items()} for _ in range(2)]
    leaves = [{k: torch.nn.Parameter(v.clone().to(dev)) for k, v in sd.items()} for sd in inits]
    opts = [torch.optim.Adam(list(lv.values()), lr=LR, eps=1e-15) for lv in leaves]

    # the restatement's three runs do not depend on the weight-gradient mode of the library (tests/conftest.py runs this module once per
    # mode): the first parametrisation computes them, the second reuses curves and final weights
    import helpers
    cached = helpers._MEMO.get("200-step curves of the restatement")
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
        helpers._MEMO["200-step curves of the restatement"] = (psnr_o, loss_o, [{k: p.detach().clone() for k, p in lv.items()} for lv in leaves])
    else:
        psnr_o, loss_o, leaves = cached
    A = psnr_o[0]
    rms = lambda x, y, lo=50: math.sqrt(sum((a - b) ** 2 for a, b in zip(x[lo:], y[lo:])) / len(x[lo:]))
    tail = lambda x: sum(x[-50:]) / 50.0
    early = max(abs(a - b) for a, b in zip(A[:25], psnr_h[:25]))
    early_loss = max(abs(a - b) / abs(a) for a, b in zip(loss_o[0][:25], loss_h[:25]))
    rms_h, rms_tw = rms(psnr_h, A), max(rms(psnr_o[1], A), rms(psnr_o[2], A))
    spread_tail = max(abs(tail(psnr_o[1]) - tail(A)), abs(tail(psnr_o[2]) - tail(A)))
    print(f"PSNR restatement {A[0]:.3f} -> mean of last 50 steps {tail(A):.3f} dB, production {psnr_h[0]:.3f} -> {tail(psnr_h):.3f} dB "
          f"(twins {tail(psnr_o[1]):.3f}, {tail(psnr_o[2]):.3f}); steps 0..24: max |dPSNR| {early:.2e} dB, max rel loss diff {early_loss:.2e}; "
          f"steps 50..199: RMS dPSNR production {rms_h:.3f} dB, restatement twins {rms_tw:.3f} dB; sampler iterations seen {sorted(set(it_h))}")
    assert tail(A) - A[0] > 5.0, "the run must actually train"
    assert early < 0.1, [(i, a, b) for i, (a, b) in enumerate(zip(A[:25], psnr_h[:25])) if abs(a - b) >= 0.1][:5]
    assert early_loss < 1e-3, early_loss
    assert rms_h <= 2.0 * rms_tw + 0.1, (rms_h, rms_tw)
    # (two twins give a noisy estimate of the spread: seven trajectories of this loop measured in round 4 have last-50-step means with a
    # standard deviation of 0.5 dB, see the module docstring; 1.5 dB = 3 sigma)
    assert abs(tail(psnr_h) - tail(A)) <= max(0.1 + 2.0 * spread_tail, 1.5), (tail(psnr_h), tail(A), spread_tail)

    # ---- held-out PSNR of the trained weight sets, all rendered by the library's eval path (same renderer, different weights)
    net.eval()
    vin, vgt = _batch(10_000, dev, B=4096)
    held = []
    with torch.no_grad():
        pa = float(orc.get_psnr(net(vin)["rgb_values"], vgt["rgb"]))
        for lv in leaves:
            ref_net = I2SDFNetwork(conf)
            ref_net.load_state_dict({k: p.detach().cpu() for k, p in lv.items()})
            ref_net = ref_net.to(dev).eval()
            held.append(float(orc.get_psnr(ref_net(vin)["rgb_values"], vgt["rgb"])))
    spread_held = max(abs(held[1] - held[0]), abs(held[2] - held[0]))
    print(f"held-out PSNR (4096 rays): production-trained {pa:.3f} dB, restatement-trained {held[0]:.3f} dB (twins {held[1]:.3f}, {held[2]:.3f})")
    assert math.isfinite(pa) and abs(pa - held[0]) <= max(0.1 + 2.0 * spread_held, 1.0), (pa, held)
