"""GPU: long-run PSNR parity of the PRODUCTION path with the restatement as an ENSEMBLE comparison (VERDICT r4 task 3).

north_star asks for "PSNR within 0.1 dB" over a training run (`get_psnr`, utils/rend_util.py:13-22; the loop is
model/trainer/recon.py:201-207,219-287: Adam(lr 5e-4, eps 1e-15), one ray batch per step, the loss of model/network/__init__.py:289-406).
One trajectory cannot show that: this loop amplifies fp32 rounding noise (Adam with eps = 1e-15 turns a gradient entry of noise magnitude
into a full +-lr step), so two executions of the reference itself are 0.5-1.5 dB apart per batch after 50 steps (measured in rounds 3-4,
DESIGN.md).  Round 4 compared one production trajectory with a 3-sigma envelope around three restatement trajectories -- a test without
the power to resolve 0.1 dB.  This test compares MEANS over an ensemble instead:

  * a LEARNABLE target: every batch's colours, depths and normals are rendered from a fixed "teacher" set of weights of the same
    architecture (synthetic.yml networks, the library's eval renderer), so the PSNR converges instead of chasing noise;
  * N_RUNS = 16 members per arm.  Member s starts from the student's initial weights times (1 + 1e-6 N(0,1)) and uses its own seed for the
    random draws of every step (stratified jitter, inverse-CDF samples, extra columns, eikonal points); the ray batches and their targets
    are shared.  Arm P = the production path (bf16x3 kernels, device-side data-dependent sampler loop, point ranges, fused loss, FusedAdam);
    arm R = the fp32 restatement's torch ops (autograd double backward) as eager ROCm kernels + torch.optim.Adam.  Member s of both arms
    sees the same initial weights, batches and draws (a paired design; the trajectories decorrelate anyway);
  * the learning rate follows the reference's scheduler (model/trainer/recon.py:204-206: ExponentialLR; there a total decay of
    `sched_decay_rate` = 0.1 over some 10^5 steps), compressed to this run with a total decay of 0.01: with a CONSTANT rate the members of ONE
    arm end up 1-2 dB apart (Adam's +-lr steps on noise-level gradient entries never settle; scripts/ab/ensemble_probe.py measured tail
    means of 36.2 +- 1.9 dB and held-out values +- 2.8 dB over 8 members of the production path alone -- a standard error of 0.5-0.7 dB
    with 16 members: no power), with the decaying rate +- 0.11 / +- 0.16 dB (profiles/r5_psnr_ensemble.txt);
  * STEPS = 300, 256 rays per step (the eager arm is launch-bound: ~30 ms per step at any batch size, 145 s for its 4800 steps -- computed
    once for both weight-gradient modes; with 1024 rays it measured +0.025 dB, SE 0.023 dB, but 51 ms per eager step); statistic of a member = mean PSNR of its last 50 training batches (and, second, the PSNR of its final weights on 2048
    held-out rays, both arms rendered by the library's eval renderer);
  * bar:  |mean_s(P_s - R_s)| <= 0.1 dB + 2 SE,  SE = std_s(P_s - R_s) / sqrt(N_RUNS), printed -- and SE itself must be <= 0.1 dB, i.e. the
    test must be able to see what it claims.
Runs in both weight-gradient modes (tests/conftest.py); arm R does not depend on the mode and is computed once (helpers.memo).
The single-trajectory test (test_gpu_training_curve_full.py) stays as a smoke check of the first 25 steps, where two fp32 executions still
agree step by step."""
import math
import os

import pytest
import torch

from oracle import i2sdf_oracle as orc

pytestmark = pytest.mark.gpu
# I2SDF_ENS_FAST=1: 8 members per arm for local runs (half the 145 s of the eager arm; SE grows by sqrt(2) and must STILL meet its bar);
# the driver's run uses 16.  Whatever the overrides, the power check below is asserted, never skipped.
N_RUNS = int(os.environ.get("I2SDF_ENS_RUNS", "8" if os.environ.get("I2SDF_ENS_FAST", "0") == "1" else "16"))
STEPS = int(os.environ.get("I2SDF_ENS_STEPS", "300"))
B, LR, TAIL = 256, 5.0e-4, 50
LR_DECAY = 0.01           # total decay of the exponential schedule over the run
W, H, F0 = 640, 480, 600.0
LKW = dict(eikonal_weight=0.1, smooth_weight=0.01, smooth_iter=150000, depth_weight=0.1, normal_weight=0.05)      # config/synthetic.yml:15-23


def _rays(step, dev, n=B):
    g = torch.Generator().manual_seed(9000 + step)
    px = torch.stack([torch.randint(120, 520, (n,), generator=g), torch.randint(40, 440, (n,), generator=g)], -1).float()     # mostly on the object
    K = torch.eye(4); K[0, 0] = K[1, 1] = F0; K[0, 2], K[1, 2] = W / 2, H / 2
    pose = torch.eye(4); pose[:3, 3] = torch.tensor([0.0, 0.0, -2.0])
    return {"uv": px.reshape(n, 1, 2).to(dev), "intrinsics": K.repeat(n, 1, 1).to(dev), "pose": pose.repeat(n, 1, 1).to(dev)}


def _targets(teacher, inp):
    """the teacher's eval render of these rays as ground truth (rgb, depth, unit normals; masks = rays that hit something)"""
    with torch.no_grad():
        o = teacher(inp)
    hit = o["weight_sum"].reshape(-1) > 0.5
    nrm = torch.nn.functional.normalize(o["normal_map"], dim=1)
    fallback = torch.tensor([0.0, 0.0, -1.0], device=nrm.device).expand_as(nrm)
    return {"rgb": o["rgb_values"].clone(), "depth": o["depth_values"].reshape(-1).clone(), "depth_mask": hit,
            "normal": torch.where(hit.unsqueeze(-1), nrm, fallback).contiguous(), "normal_mask": hit.clone()}


def _teacher_weights(ocfg):
    """A scene worth learning, made of the same architecture: the geometric-init sphere with radius 0.75 instead of the student's 0.6 and
    every weight perturbed by 5 % (a bumpy sphere); a radiance net that reacts to the view direction (its PE columns x 6) with a wide
    output range (last layer x 25) and distinct channel means.  Probed with the CPU oracle: 85 % of these rays hit, colours of the hits
    (0.51, 0.19, 0.35) +- 0.06, the student's first render is 14.6 dB away; 300 steps take it to ~39 dB."""
    sd = orc.perturb_params(orc.init_params(ocfg, seed=101), scale=0.05, seed=3)
    sd["density.beta"] = torch.tensor(0.05)
    b = sd["implicit_network.lin8.bias"].clone()
    b[0] = -0.75
    sd["implicit_network.lin8.bias"] = b
    v = sd["rendering_network.lin0.weight_v"].clone()
    v[:, :27] *= 6.0
    sd["rendering_network.lin0.weight_v"] = v
    sd["rendering_network.lin0.weight_g"] = v.norm(dim=1, keepdim=True)
    sd["rendering_network.lin4.weight_g"] = sd["rendering_network.lin4.weight_g"] * 25.0
    sd["rendering_network.lin4.bias"] = torch.tensor([0.8, -0.4, 0.2])
    return sd


def _member_init(sd0, s):
    g = torch.Generator().manual_seed(31_000 + s)
    return {k: (v * (1 + 1e-6 * torch.randn(v.shape, generator=g))).to(torch.float32) for k, v in sd0.items()}


def _stats(d):
    n = len(d)
    m = sum(d) / n
    sd = math.sqrt(sum((x - m) ** 2 for x in d) / (n - 1)) if n > 1 else float("nan")
    return m, sd, sd / math.sqrt(n)


@pytest.mark.parametrize("saves", ["default", "fp32-saves"])
def test_ensemble_tail_psnr_production_vs_restatement(saves, wgrad_mode):
    """`saves`: storage of the saved SDF tensors abars / G(hbar) / G(a) in the production arm -- "default" = the module's (packed 24-bit records with the
    two-plane weight gradients, I2SDF_OPT_SAVES24; fp32 in the fp32-equivalent mode), "fp32-saves" = fp32 storage with the two-plane weight gradients: the
    THIRD parametrisation the round-5 review asked the ensemble bar to hold in before narrower storage may be the default."""
    if saves == "fp32-saves" and wgrad_mode != "wgrad-bf16x2":
        pytest.skip("fp32 storage is what the fp32-equivalent mode runs by default: covered by [default]")
    from i2sdf_amd import I2SDFNetwork, I2SDFLoss, FusedAdam, synthetic_conf
    import helpers
    dev = torch.device("cuda:0")
    conf = dict(synthetic_conf(False))
    conf["use_normal"] = True
    ocfg = orc.synthetic_cfg(False)
    ocfg.use_normal = True
    lc = orc.LossCfg(**LKW)

    # ---- the teacher (see _teacher_weights)
    sd_t = _teacher_weights(ocfg)
    teacher = I2SDFNetwork(conf)
    teacher.load_state_dict(sd_t)
    teacher = teacher.to(dev).eval()
    batches = []
    for step in range(STEPS):
        inp = _rays(step, dev)
        batches.append((inp, _targets(teacher, inp)))
    vin = _rays(1_000_000, dev, n=2048)
    vgt = _targets(teacher, vin)
    hit_frac = float(torch.stack([gt["depth_mask"].float().mean() for _, gt in batches]).mean())
    assert 0.3 < hit_frac < 1.0, hit_frac                      # the masked loss terms are exercised, and so is the background

    # ---- the student's initial weights: another seed (other radiance net, other random parts of the geometry net)
    sd0 = orc.init_params(ocfg, seed=11)
    sd0["density.beta"] = torch.tensor(0.05)

    def held_out(sd):
        n = I2SDFNetwork(conf)
        n.load_state_dict({k: v.detach().cpu() for k, v in sd.items()})
        n = n.to(dev).eval()
        with torch.no_grad():
            return float(orc.get_psnr(n(vin)["rgb_values"], vgt["rgb"]))

    def draws_for(eng, net, s, step):
        return eng.training_draws(B, 7_000_000 + 100_003 * s + step, dev, net.scene_bounding_sphere, want_eik=True)

    # ---- arm P: the production path.  One module, re-loaded per member (the engine, its streams and the optimizer state are rebuilt)
    tails_p, held_p, first_p, iters_seen = [], [], [], set()
    net = I2SDFNetwork(conf).to(dev).train()
    eng = net._engine_for(dev)
    assert eng.train_forward_bf16x3 and eng.sdf_backward_bf16x3 and eng.wgrad_bf16x3 and eng.rgb_bf16x3 and eng.sdf_forward_bf16x3
    assert net.fused_draws and net.force_iters == 0 and eng.parts >= 2
    if saves == "fp32-saves":
        eng.set_saves24(False)
    assert (eng.saves24_points(B * 100, (B * 100 + 127) // 128 * 128) > 0) == (saves == "default" and wgrad_mode == "wgrad-bf16x2"), "storage mode of this arm"
    loss_fn = I2SDFLoss(**LKW)
    for s in range(N_RUNS):
        net.load_state_dict(_member_init(sd0, s))
        net.train()
        opt = FusedAdam(net, lr=LR, eps=1e-15)
        sched = torch.optim.lr_scheduler.ExponentialLR(opt, LR_DECAY ** (1.0 / STEPS))
        ps = []
        for step in range(STEPS):
            inp, gt = batches[step]
            out = net(inp, draws=draws_for(eng, net, s, step))
            losses = loss_fn(out, gt, step)
            opt.zero_grad(set_to_none=True)
            losses["loss"].backward()
            opt.step()
            sched.step()
            ps.append(orc.get_psnr(out["rgb_values"].detach(), gt["rgb"]))
            if step % 50 == 0:
                iters_seen.add(int(net.last_sampler_iters.item()))
        ps = torch.stack(ps).tolist()
        first_p.append(ps[0])
        tails_p.append(sum(ps[-TAIL:]) / TAIL)
        held_p.append(held_out({k: v for k, v in net.state_dict().items()}))

    # ---- arm R: the restatement as eager ROCm ops (independent of the library's weight-gradient mode: computed once per session).
    # The draws are the library's fused draws kernel keyed by the same seeds -- inputs, not arithmetic under test.
    def arm_r():
        tails, held, first = [], [], []
        for s in range(N_RUNS):
            lv = {k: torch.nn.Parameter(v.clone().to(dev)) for k, v in _member_init(sd0, s).items()}
            opt = torch.optim.Adam(list(lv.values()), lr=LR, eps=1e-15)
            sched = torch.optim.lr_scheduler.ExponentialLR(opt, LR_DECAY ** (1.0 / STEPS))
            ps = []
            for step in range(STEPS):
                inp, gt = batches[step]
                d = draws_for(eng, net, s, step)
                dr = orc.Draws(strat_u=d["strat_u"], cdf_u=d["cdf_u"], extra_idx=d["extra_idx"], eik_idx=d["eik_idx"], eik_pts=d["eik_pts"],
                               nbr_off=d["nbr_off"])
                o_out, _, grads = orc.training_step_grads({k: p.detach() for k, p in lv.items()}, ocfg, inp, gt, lc, dr, step=step)
                opt.zero_grad(set_to_none=True)
                for k, p in lv.items():
                    p.grad = grads[k].reshape(p.shape).clone()
                opt.step()
                sched.step()
                ps.append(orc.get_psnr(o_out["rgb_values"].detach(), gt["rgb"]))
            ps = torch.stack(ps).tolist()
            first.append(ps[0])
            tails.append(sum(ps[-TAIL:]) / TAIL)
            held.append(held_out({k: p.detach() for k, p in lv.items()}))
        return tails, held, first
    tails_r, held_r, first_r = helpers.memo(f"psnr ensemble arm R {N_RUNS}x{STEPS}", arm_r)

    d_tail = [a - b for a, b in zip(tails_p, tails_r)]
    d_held = [a - b for a, b in zip(held_p, held_r)]
    m_t, sd_t_, se_t = _stats(d_tail)
    m_h, sd_h, se_h = _stats(d_held)
    mp, sdp, _ = _stats(tails_p)
    mr, sdr, _ = _stats(tails_r)
    print(f"ensemble of {N_RUNS} x {STEPS} steps, {B} rays/step, hit fraction {hit_frac:.2f}, sampler iterations seen {sorted(iters_seen)}: "
          f"PSNR at step 0 {sum(first_r) / N_RUNS:.2f} dB; tail ({TAIL} last batches) production {mp:.3f} +- {sdp:.3f} dB, restatement {mr:.3f} +- {sdr:.3f} dB; "
          f"paired difference {m_t:+.4f} dB, SE {se_t:.4f} dB (std {sd_t_:.3f}); held-out (2048 rays) production {sum(held_p) / N_RUNS:.3f}, "
          f"restatement {sum(held_r) / N_RUNS:.3f} dB, difference {m_h:+.4f} dB, SE {se_h:.4f} dB")
    assert max(abs(a - b) for a, b in zip(first_p, first_r)) < 1e-2, "step 0 renders the same weights with the same draws"
    assert mr - sum(first_r) / N_RUNS > 5.0, "the runs must actually train"
    # the power of the test: it must resolve what it claims -- with ANY member / step count the environment selected (a reduced run that
    # cannot resolve 0.1 dB fails here instead of passing on a looser effective bar)
    assert se_t <= 0.1 and se_h <= 0.1, (se_t, se_h, N_RUNS, STEPS)
    # effective tolerance: |difference| <= 0.1 dB + 2 SE <= 0.3 dB in the worst admissible case; measured SE 0.05 dB -> 0.2 dB (README)
    assert abs(m_t) <= 0.1 + 2.0 * se_t, (m_t, se_t)
    assert abs(m_h) <= 0.1 + 2.0 * se_h, (m_h, se_h)
