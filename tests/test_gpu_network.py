"""GPU: the drop-in I2SDFNetwork module end to end (ray set-up -> sampler -> MLPs -> composite, forward and backward)
vs the reference's golden outputs/gradients and vs the oracle at the full synthetic.yml shapes.

Tolerances.  With the depths given (z override) every output and gradient must match to 1e-4 max-norm relative
(north_star).  With the sampler in the loop individual depths are ill-conditioned (tests/test_gpu_sampler.py), which
propagates into depth/rgb at the ~1e-4..1e-3 level for the reference itself (its own fp32-vs-fp64 spread is measured
in the test and bounds the tolerance used)."""
import numpy as np
import pytest
import torch

from oracle import i2sdf_oracle as orc
from helpers import assert_close, rel_max, sd_from_npz, t, camera_inputs, make_draws, make_gt

pytestmark = pytest.mark.gpu


def build(conf, sd, use_normal=True, train=False):
    from i2sdf_amd import I2SDFNetwork
    conf = dict(conf)
    conf["use_normal"] = use_normal
    net = I2SDFNetwork(conf)
    net.load_state_dict(sd)
    net = net.cuda()
    net.train(train)
    return net


def cuda(d):
    return {k: (v.cuda() if torch.is_tensor(v) else v) for k, v in d.items()}


def _eval_rays(tvec, B=1024, W=32, H=32, f=30.0):
    K = torch.eye(4); K[0, 0], K[1, 1], K[0, 2], K[1, 2] = f, f, W / 2, H / 2
    pose = torch.eye(4); pose[:3, 3] = torch.as_tensor(tvec)
    ys, xs = torch.meshgrid(torch.arange(H), torch.arange(W), indexing="ij")
    uv = torch.stack([xs, ys], -1).float().reshape(1, -1, 2)[:, :B]
    return {"uv": uv, "intrinsics": K.unsqueeze(0), "pose": pose.unsqueeze(0)}


def test_state_dict_keys_and_order_match_reference(golden):
    from i2sdf_amd import I2SDFNetwork, plumbing_conf
    z = golden("g9_train_light")
    ref_keys = [k[3:] for k in z.files if k.startswith("sd.")]
    net = I2SDFNetwork(plumbing_conf(skip=True, light=True))
    assert list(net.state_dict().keys()) == ref_keys
    for k, v in net.state_dict().items():
        assert tuple(v.shape) == tuple(z["sd." + k].shape), k


@pytest.mark.parametrize("tag", ["in", "out"])
def test_eval_forward_vs_reference_golden(golden, tag):
    from i2sdf_amd import plumbing_conf
    z = golden("g7_g8_eval")
    sd = sd_from_npz(z, "sd.")
    sd["density.beta"] = torch.tensor(float(z[f"{tag}.beta_param"]))
    net = build(plumbing_conf(), sd)
    with torch.no_grad():
        out = net(cuda(_eval_rays(z[f"{tag}.t"])))
    assert int(net.last_sampler_iters.item()) == int(z[f"{tag}.iters"])
    for k, tol in (("rgb_values", 5e-4), ("depth_values", 1e-3), ("weight_sum", 5e-4)):
        assert out[k].shape == tuple(z[f"{tag}.out.{k}"].shape)
        assert_close(out[k].cpu(), z[f"{tag}.out.{k}"], tol, k)
    # the normal of a ray that hits nothing (weight_sum ~ 0) is the direction of a vanishing sum: compare where it is defined
    hit = t(z[f"{tag}.out.weight_sum"]).reshape(-1) > 1e-2
    assert out["normal_map"].shape == tuple(z[f"{tag}.out.normal_map"].shape)
    assert_close(out["normal_map"].cpu()[hit], t(z[f"{tag}.out.normal_map"])[hit], 5e-3, "normal_map (rays with weight_sum > 0.01)")


@pytest.mark.parametrize("light", [False, True])
def test_eval_render_given_depths_full_size(light):
    """z override: identical samples on both sides -> 1e-4 parity of every output (incl. light mask)."""
    from i2sdf_amd import synthetic_conf
    ocfg = orc.synthetic_cfg(light)
    sd = orc.perturb_params(orc.init_params(ocfg, seed=31), 0.03, seed=32)
    sd["density.beta"] = torch.tensor(0.05)
    net = build(synthetic_conf(light), sd)
    B = 96
    inp = camera_inputs(B, (0.0, 0.0, -2.0), train_layout=False, seed=4)
    cam, dirs, dn = orc.prepare_rays(inp["uv"], inp["pose"], inp["intrinsics"])
    z_all, z_eik = orc.sample_z_vals(sd, ocfg, dirs, cam, training=False)
    ref = orc.network_forward({k: v.double() for k, v in sd.items()}, ocfg, {k: v.double() for k, v in inp.items()}, training=False,
                              z_override=(z_all.double(), z_eik.double()))
    eng = net._engine_for("cuda:0")
    c, d, n = eng.ray_setup(inp["uv"].cuda(), inp["pose"].cuda(), inp["intrinsics"].cuda())
    with torch.no_grad():
        out = net.render(cuda(inp), c, d, n, z_all.cuda(), z_eik.cuda())
    keys = ["rgb_values", "depth_values", "weight_sum"] + (["light_mask"] if light else [])
    for k in keys:
        assert_close(out[k].cpu(), ref[k], 1e-4, k)
    hit = ref["weight_sum"].reshape(-1) > 1e-2
    assert_close(out["normal_map"].cpu()[hit], ref["normal_map"][hit], 1e-3, "normal_map (rays with weight_sum > 0.01)")


@pytest.mark.parametrize("name,light", [("g9_train", False), ("g9_train_light", True)])
def test_train_step_vs_reference_golden(golden, name, light):
    """Forward + I2SDFLoss + backward with the reference's own recorded random draws; compares every output, the loss
    and every parameter gradient with the reference's (fixture G9)."""
    from i2sdf_amd import plumbing_conf, I2SDFLoss
    z = golden(name)
    sd = sd_from_npz(z, "sd.")
    net = build(plumbing_conf(skip=True, light=light), sd, train=True)
    inp = {k[3:]: t(z[k]) for k in z.files if k.startswith("in.")}
    gt = {k[3:]: t(z[k]) for k in z.files if k.startswith("gt.")}
    draws = {k[5:]: t(z[k]).cuda() for k in z.files if k.startswith("draw.")}
    lk = {k: v for k, v in z["loss_kwargs"]}
    loss_fn = I2SDFLoss(**{k: (None if v == "None" else float(v)) for k, v in lk.items()})
    out = net(cuda(inp), draws=draws)
    losses = loss_fn(out, cuda(gt), 10)
    net.zero_grad()
    losses["loss"].backward()
    hit = t(z["out.weight_sum"]).reshape(-1) > 1e-2
    for k in z.files:
        if k.startswith("out."):
            # grad_theta / diff_norm are evaluated AT sampler-chosen near-surface points (z_eik): they inherit the
            # ill-conditioning of individual sample depths; the strict 1e-4 check is test_train_step_given_depths_full_size
            tol = 5e-3 if k.endswith(("normal_values", "diff_norm", "grad_theta")) else 1e-3
            assert out[k[4:]].shape == tuple(z[k].shape), k
            if k.endswith("normal_values"):
                assert_close(out[k[4:]].detach().cpu()[hit], t(z[k])[hit], tol, k + " (weight_sum > 0.01)")
            else:
                assert_close(out[k[4:]].detach().cpu(), z[k], tol, k)
    assert_close(losses["loss"].detach().cpu(), z["loss.loss"], 1e-3, "loss")
    for n_, p in net.named_parameters():
        g = p.grad if p.grad is not None else torch.zeros_like(p)
        assert_close(g.cpu(), z["grad." + n_], 2e-2, "grad " + n_)


@pytest.mark.parametrize("light", [False, True])
def test_train_step_given_depths_full_size(light):
    """Identical depths and draws on both sides, synthetic.yml shapes: outputs and all parameter gradients to 1e-4
    (fp64 oracle as arbiter)."""
    from i2sdf_amd import synthetic_conf, I2SDFLoss
    ocfg = orc.synthetic_cfg(light)
    ocfg.use_normal = True
    sd = orc.perturb_params(orc.init_params(ocfg, seed=41), 0.03, seed=42)
    sd["density.beta"] = torch.tensor(0.05)
    net = build(synthetic_conf(light), sd, train=True)
    B = 40
    inp = camera_inputs(B, (0.0, 0.0, -2.0), seed=5)
    gt = make_gt(B, light=light)
    cam, dirs, dn = orc.prepare_rays(inp["uv"], inp["pose"], inp["intrinsics"])
    dr = make_draws(ocfg, B, n_row=128, seed=2)
    z_all, z_eik = orc.sample_z_vals(sd, ocfg, dirs, cam, training=True, draws=dr, force_iters=1)
    lc = orc.LossCfg(eikonal_weight=0.1, smooth_weight=0.01, smooth_iter=None, depth_weight=0.1, normal_weight=0.05,
                     light_mask_weight=0.5 if light else 0.0)
    D = torch.float64
    d64 = orc.Draws(eik_pts=dr.eik_pts.to(D), nbr_off=dr.nbr_off.to(D))
    gt64 = {k: (v.to(D) if v.dtype.is_floating_point else v) for k, v in gt.items()}
    ref_out, ref_loss, ref_g = orc.training_step_grads({k: v.to(D) for k, v in sd.items()}, ocfg, {k: v.to(D) for k, v in inp.items()}, gt64,
                                                       lc, d64, step=10, z_override=(z_all.to(D), z_eik.to(D)))
    eng = net._engine_for("cuda:0")
    c, d, n = eng.ray_setup(inp["uv"].cuda(), inp["pose"].cuda(), inp["intrinsics"].cuda())
    out = net.render(cuda(inp), c, d, n, z_all.cuda(), z_eik.cuda(), draws={"eik_pts": dr.eik_pts.cuda(), "nbr_off": dr.nbr_off.cuda()})
    loss_fn = I2SDFLoss(eikonal_weight=0.1, smooth_weight=0.01, smooth_iter=None, depth_weight=0.1, normal_weight=0.05,
                        light_mask_weight=0.5 if light else 0.0)
    losses = loss_fn(out, cuda(gt), 10)
    net.zero_grad()
    losses["loss"].backward()
    for k in ("rgb_values", "depth_values", "weight_sum", "grad_theta") + (("light_mask",) if light else ()):
        assert_close(out[k].detach().cpu(), ref_out[k], 1e-4, k)
    hit = ref_out["weight_sum"].reshape(-1) > 1e-2
    assert_close(out["normal_values"].detach().cpu()[hit], ref_out["normal_values"][hit], 1e-3, "normal_values (weight_sum > 0.01)")
    assert_close(losses["loss"].detach().cpu(), ref_loss["loss"], 1e-5, "loss")
    worst = 0.0
    for n_, p in net.named_parameters():
        g = p.grad if p.grad is not None else torch.zeros_like(p)
        worst = max(worst, assert_close(g.cpu(), ref_g[n_], 1e-4, "grad " + n_))
    print("worst relative parameter-gradient error", worst)


def test_train_step_bf16x3_matches_fp32_kernels(B=400):
    """The same training step (identical depths and draws) with every bf16x3 kernel enabled (the default) and with the plain
    fp32-MFMA kernels (`bf16x3: false`): outputs agree to 1e-5, and outputs and all parameter gradients of BOTH are within the 1e-4
    parity bar of the fp64 oracle.  Seeds / batch as chosen here are well conditioned.  Two things make ANY fp32 evaluation (the
    reference's included, bf16x3 or not) differ from fp64 by 1e-3 on unlucky batches, both measured with scripts/dev/*_probe.py:
    a ray whose weighted normal sum nearly cancels, and radiance-net pre-activations within fp32 rounding of zero,
    where the ReLU mask of the backward flips (one point's contribution to a bias gradient appears or disappears).  The
    kernel-level tests in test_gpu_backward.py bound the arithmetic itself."""
    from i2sdf_amd import synthetic_conf, I2SDFLoss
    ocfg = orc.synthetic_cfg(False)
    ocfg.use_normal = True
    sd = orc.perturb_params(orc.init_params(ocfg, seed=41), 0.03, seed=42)
    sd["density.beta"] = torch.tensor(0.05)
    # 400 x 97 + 3 x 400 points = 313 workgroups: bulk (bf16x3 or fp32) and split-K tail kernels both run
    inp = camera_inputs(B, (0.0, 0.0, -2.0), seed=5)
    gt = make_gt(B)
    dr = make_draws(ocfg, B, n_row=128, seed=2)
    cam, dirs, dn = orc.prepare_rays(inp["uv"], inp["pose"], inp["intrinsics"])
    z_all, z_eik = orc.sample_z_vals(sd, ocfg, dirs, cam, training=True, draws=dr, force_iters=1)
    res = {}
    for mode in (True, False):
        conf = dict(synthetic_conf(False))
        conf["bf16x3"] = mode
        net = build(conf, sd, train=True)
        eng = net._engine_for("cuda:0")
        assert eng.sdf_forward_bf16x3 == mode and eng.wgrad_bf16x3 == mode and eng.train_forward_bf16x3 == mode
        assert eng.sdf_backward_bf16x3 == mode and eng.rgb_bf16x3 == mode
        c, d, n = eng.ray_setup(inp["uv"].cuda(), inp["pose"].cuda(), inp["intrinsics"].cuda())
        out = net.render(cuda(inp), c, d, n, z_all.cuda(), z_eik.cuda(), draws={"eik_pts": dr.eik_pts.cuda(), "nbr_off": dr.nbr_off.cuda()})
        loss_fn = I2SDFLoss(eikonal_weight=0.1, smooth_weight=0.01, smooth_iter=None, depth_weight=0.1, normal_weight=0.05)
        losses = loss_fn(out, cuda(gt), 10)
        net.zero_grad()
        losses["loss"].backward()
        res[mode] = ({k: v.detach().cpu() for k, v in out.items()}, float(losses["loss"].detach()),
                     {n_: (p.grad if p.grad is not None else torch.zeros_like(p)).cpu() for n_, p in net.named_parameters()})
    (o3, l3, g3), (o1, l1, g1) = res[True], res[False]
    # fp64 oracle as the arbiter for both
    D = torch.float64
    lc = orc.LossCfg(eikonal_weight=0.1, smooth_weight=0.01, smooth_iter=None, depth_weight=0.1, normal_weight=0.05)
    d64 = orc.Draws(eik_pts=dr.eik_pts.to(D), nbr_off=dr.nbr_off.to(D))
    gt64 = {k: (v.to(D) if v.dtype.is_floating_point else v) for k, v in gt.items()}
    ref_out, ref_loss, ref_g = orc.training_step_grads({k: v.to(D) for k, v in sd.items()}, ocfg, {k: v.to(D) for k, v in inp.items()}, gt64,
                                                       lc, d64, step=10, z_override=(z_all.to(D), z_eik.to(D)))
    for k in ("rgb_values", "depth_values", "weight_sum", "grad_theta"):
        assert_close(o3[k], o1[k], 1e-5, k + " (bf16x3 vs fp32 kernels)")
        assert_close(o3[k], ref_out[k], 1e-4, k + " (bf16x3 vs fp64)")
    assert abs(l3 - l1) <= 1e-6 * abs(l1)
    e3 = max(assert_close(g3[k], ref_g[k], 1e-4, "grad " + k + " (bf16x3 vs fp64)") for k in g1)
    e1 = max(assert_close(g1[k], ref_g[k], 1e-4, "grad " + k + " (fp32 kernels vs fp64)") for k in g1)
    print(f"worst relative parameter-gradient error vs fp64: bf16x3 kernels {e3:.2e}, fp32-MFMA kernels {e1:.2e}")
