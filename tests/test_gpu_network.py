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


@pytest.mark.wgrad_independent
def test_state_dict_keys_and_order_match_reference(golden):
    from i2sdf_amd import I2SDFNetwork, plumbing_conf
    z = golden("g9_train_light")
    ref_keys = [k[3:] for k in z.files if k.startswith("sd.")]
    net = I2SDFNetwork(plumbing_conf(skip=True, light=True))
    assert list(net.state_dict().keys()) == ref_keys
    for k, v in net.state_dict().items():
        assert tuple(v.shape) == tuple(z["sd." + k].shape), k


def _spread(a, b, keys, hit=None):
    """max-norm relative difference of two oracle runs (fp32 vs fp64): the conditioning of the quantity itself."""
    out = {}
    for k in keys:
        x, y = a[k].detach().double(), b[k].detach().double()
        if hit is not None and k.startswith("normal"):
            x, y = x[hit], y[hit]
        out[k] = rel_max(x, y)
    return out


@pytest.mark.wgrad_independent
@pytest.mark.parametrize("tag", ["in", "out"])
def test_eval_forward_vs_reference_golden(golden, tag):
    """Sampler in the loop.  Individual depths are ill-conditioned where the inverse CDF is flat, so the bound used per output
    is MEASURED here: the oracle's own fp32-vs-fp64 spread on these rays (x3 margin), never below the 1e-4 parity bar."""
    from i2sdf_amd import plumbing_conf
    z = golden("g7_g8_eval")
    sd = sd_from_npz(z, "sd.")
    sd["density.beta"] = torch.tensor(float(z[f"{tag}.beta_param"]))
    net = build(plumbing_conf(), sd)
    inp = _eval_rays(z[f"{tag}.t"])
    with torch.no_grad():
        out = net(cuda(inp))
    assert int(net.last_sampler_iters.item()) == int(z[f"{tag}.iters"])
    ocfg = orc.plumbing_cfg()
    o32 = orc.network_forward(sd, ocfg, inp, training=False)
    o64 = orc.network_forward({k: v.double() for k, v in sd.items()}, ocfg, {k: v.double() for k, v in inp.items()}, training=False)
    hit = t(z[f"{tag}.out.weight_sum"]).reshape(-1) > 1e-2
    keys = ("rgb_values", "depth_values", "weight_sum", "normal_map")
    spread = _spread(o32, o64, keys, hit)
    print("fp32-vs-fp64 spread of the oracle with its own sampler:", spread)
    for k in keys:
        tol = max(1e-4, 3.0 * spread[k])
        assert out[k].shape == tuple(z[f"{tag}.out.{k}"].shape)
        if k == "normal_map":
            # the normal of a ray that hits nothing (weight_sum ~ 0) is the direction of a vanishing sum: compare where it is defined
            assert_close(out[k].cpu()[hit], t(z[f"{tag}.out.{k}"])[hit], tol, "normal_map (rays with weight_sum > 0.01)")
        else:
            assert_close(out[k].cpu(), z[f"{tag}.out.{k}"], tol, k)


@pytest.mark.wgrad_independent
@pytest.mark.parametrize("tag", ["in", "out"])
def test_eval_render_given_reference_depths_golden(golden, tag):
    """The reference's OWN eval depths (G7 z_vals) through render(): every output of the reference's recorded render (G8) at 1e-4."""
    from i2sdf_amd import plumbing_conf
    z = golden("g7_g8_eval")
    sd = sd_from_npz(z, "sd.")
    sd["density.beta"] = torch.tensor(float(z[f"{tag}.beta_param"]))
    net = build(plumbing_conf(), sd)
    inp = _eval_rays(z[f"{tag}.t"])
    eng = net._engine_for("cuda:0")
    c, d, n = eng.ray_setup(inp["uv"].cuda(), inp["pose"].cuda(), inp["intrinsics"].cuda())
    zv = t(z[f"{tag}.z_vals"]).cuda()
    with torch.no_grad():
        out = net.render(cuda(inp), c, d, n, zv, zv[:, :1].contiguous())
    for k in ("rgb_values", "depth_values", "weight_sum"):
        assert_close(out[k].cpu(), z[f"{tag}.out.{k}"], 1e-4, k)
    hit = t(z[f"{tag}.out.weight_sum"]).reshape(-1) > 1e-2
    assert_close(out["normal_map"].cpu()[hit], t(z[f"{tag}.out.normal_map"])[hit], 1e-4, "normal_map (rays with weight_sum > 0.01)")


@pytest.mark.wgrad_independent
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
    assert_close(out["normal_map"].cpu()[hit], ref["normal_map"][hit], 1e-4, "normal_map (rays with weight_sum > 0.01)")      # measured 1.3e-5


def _g9_setup(z, light):
    from i2sdf_amd import plumbing_conf, I2SDFLoss
    sd = sd_from_npz(z, "sd.")
    net = build(plumbing_conf(skip=True, light=light), sd, train=True)
    inp = {k[3:]: t(z[k]) for k in z.files if k.startswith("in.")}
    gt = {k[3:]: t(z[k]) for k in z.files if k.startswith("gt.")}
    lk = {k: v for k, v in z["loss_kwargs"]}
    loss_fn = I2SDFLoss(**{k: (None if v == "None" else float(v)) for k, v in lk.items()})
    return sd, net, inp, gt, lk, loss_fn


def _oracle_lc(lk):
    return orc.LossCfg(eikonal_weight=float(lk["eikonal_weight"]), smooth_weight=float(lk["smooth_weight"]), smooth_iter=None,
                       depth_weight=float(lk["depth_weight"]), normal_weight=float(lk["normal_weight"]),
                       bubble_weight=float(lk["bubble_weight"]), light_mask_weight=float(lk.get("light_mask_weight", 0.0)))


@pytest.mark.parametrize("name,light", [("g9_train", False), ("g9_train_light", True)])
def test_train_step_vs_reference_golden(golden, name, light):
    """Sampler in the loop: forward + I2SDFLoss + backward with the reference's own recorded random draws; every output, the
    loss and every parameter gradient vs the reference's (fixture G9).  Individual inverse-CDF depths are ill-conditioned in the
    reference itself, so the bound per quantity is MEASURED in the test: the largest deviation from the fp32 oracle (= the
    reference, bitwise) among the fp64 oracle and two fp32 oracle runs with rounding-level weight noise, each with ITS sampler in
    the loop, x3, floor 1e-4; gradients additionally modulo backward-mask flips of ReLU units at zero within rounding
    (helpers.relu_flip_analysis).  The tight check with identical depths is test_train_step_given_reference_depths_golden."""
    from helpers import measured_spread, network_of, relu_flip_analysis, explain_by_relu_flips
    z = golden(name)
    sd, net, inp, gt, lk, loss_fn = _g9_setup(z, light)
    draws = {k[5:]: t(z[k]).cuda() for k in z.files if k.startswith("draw.")}
    out = net(cuda(inp), draws=draws)
    losses = loss_fn(out, cuda(gt), 10)
    net.zero_grad()
    losses["loss"].backward()
    ocfg = orc.plumbing_cfg(skip=True, light=light)
    ocfg.use_normal = True
    lc = _oracle_lc(lk)
    hit = t(z["out.weight_sum"]).reshape(-1) > 1e-2
    okeys = [k[4:] for k in z.files if k.startswith("out.")]

    def run(sd_, dt, what="all"):
        cast = lambda v: v.to(dt) if (torch.is_tensor(v) and v.dtype.is_floating_point) else v
        dr = orc.Draws(**{k[5:]: cast(t(z[k])) for k in z.files if k.startswith("draw.")})
        o, l, g_ = orc.training_step_grads({k_: cast(v) for k_, v in sd_.items()}, ocfg, {k_: cast(v) for k_, v in inp.items()},
                                           {k_: cast(v) for k_, v in gt.items()}, lc, dr, step=10)
        if what == "grads":
            return g_
        res = {"out." + k_: (o[k_][hit] if k_ == "normal_values" else o[k_]) for k_ in okeys}
        res["loss"] = l["loss"]
        res.update({"grad." + k_: v for k_, v in g_.items()})
        return res

    # weight noise of 1e-6: the HIP SDF forward agrees with fp64 to 5e-7 max-norm (test_gpu_sdf_forward.py), i.e. the sdf values the
    # sampler sees differ from the reference's at that level -- the noisy oracle members must be at least as far away
    spread = measured_spread(run, sd, n_perturbed=3, rel=1e-6)
    print("measured conditioning with the sampler in the loop:", {k: "%.1e" % v for k, v in spread.items() if not k.startswith("grad.")},
          "worst gradient %.1e" % max(v for k, v in spread.items() if k.startswith("grad.")))
    failures = []
    for k in okeys:
        tol = max(1e-4, 3.0 * spread["out." + k])
        assert out[k].shape == tuple(z["out." + k].shape), k
        a, b = (out[k].detach().cpu()[hit], t(z["out." + k])[hit]) if k == "normal_values" else (out[k].detach().cpu(), t(z["out." + k]))
        e = rel_max(a, b)
        print("  %-16s err %.2e  tol %.2e" % (k, e, tol))
        if k in ("grad_theta", "diff_norm"):
            # evaluated AT one sampler-chosen depth per ray (z_samples_eik): the max over the rays is the single worst-conditioned
            # depth pick, a heavy-tailed statistic -- require 97 % of the rows within the measured tolerance and the worst within 10x
            rows = (a.double() - b.double()).abs().reshape(a.shape[0], -1).max(1)[0] / float(b.double().abs().max())
            if float((rows <= tol).double().mean()) < 0.97 or e > 10 * tol:
                failures.append((k, e, tol, float((rows <= tol).double().mean())))
        elif e > tol:
            failures.append((k, e, tol))
    assert not failures, failures
    assert_close(losses["loss"].detach().cpu(), z["loss.loss"], max(1e-4, 3.0 * spread["loss"]), "loss")
    by_net = {}
    for k, v in spread.items():
        if k.startswith("grad."):
            by_net[network_of(k[5:])] = max(by_net.get(network_of(k[5:]), 0.0), v)
    _, cands, deltas = relu_flip_analysis(lambda: run(sd, torch.float32, "grads"), tau=1e-6)
    err, scale = {}, {}
    for n_, p in net.named_parameters():
        g = p.grad if p.grad is not None else torch.zeros_like(p)
        scale[n_] = float(np.abs(z["grad." + n_]).max())
        err[n_] = g.detach().cpu().double().reshape(-1) - t(z["grad." + n_]).double().reshape(-1)
        if scale[n_] == 0.0:
            assert float(g.abs().max()) == 0.0, n_
    raw = {n_: float(err[n_].abs().max()) / scale[n_] for n_ in err if scale[n_] > 0}
    chosen, res = explain_by_relu_flips(err, [{n_: d[n_].reshape(-1) for n_ in d} for d in deltas], scale)
    print(f"{len(cands)} ReLU units with |pre-activation| < 1e-6; flips used:", [(cands[i][0], cands[i][1], cands[i][2]) for i in chosen])
    bad = []
    for n_ in sorted(res, key=lambda k_: -raw[k_])[:6]:
        print("  grad %-40s raw err %.2e   after flips %.2e   oracle spread %.2e" % (n_, raw[n_], res[n_], spread["grad." + n_]))
    for n_, r in res.items():
        tol = max(1e-4, 3.0 * by_net[network_of(n_)])
        if r > tol:
            bad.append((n_, r, tol))
    assert len(chosen) <= 4 and not bad, (chosen, bad)


@pytest.mark.parametrize("name,light", [("g9_train", False), ("g9_train_light", True)])
def test_train_step_given_reference_depths_golden(golden, name, light):
    """The reference's OWN recorded depths (ref.z_vals / ref.z_eik of G9) and draws through render(): every output, every loss
    term and every parameter gradient against the reference's recorded numbers at 1e-4 (north_star)."""
    z = golden(name)
    sd, net, inp, gt, lk, loss_fn = _g9_setup(z, light)
    eng = net._engine_for("cuda:0")
    c, d, n = eng.ray_setup(inp["uv"].cuda(), inp["pose"].cuda(), inp["intrinsics"].cuda())
    out = net.render(cuda(inp), c, d, n, t(z["ref.z_vals"]).cuda(), t(z["ref.z_eik"]).cuda(),
                     draws={"eik_pts": t(z["draw.eik_pts"]).cuda(), "nbr_off": t(z["draw.nbr_off"]).cuda()})
    losses = loss_fn(out, cuda(gt), 10)
    net.zero_grad()
    losses["loss"].backward()
    hit = t(z["out.weight_sum"]).reshape(-1) > 1e-2
    for k in z.files:
        if k.startswith("out."):
            assert out[k[4:]].shape == tuple(z[k].shape), k
            if k.endswith("normal_values"):
                assert_close(out[k[4:]].detach().cpu()[hit], t(z[k])[hit], 1e-4, k + " (weight_sum > 0.01)")
            else:
                assert_close(out[k[4:]].detach().cpu(), z[k], 1e-4, k)
        if k.startswith("loss."):
            if float(np.abs(z[k])) > 0:
                assert_close(losses[k[5:]].detach().cpu(), z[k], 1e-4, k)
    worst = 0.0
    for n_, p in net.named_parameters():
        g = p.grad if p.grad is not None else torch.zeros_like(p)
        if np.abs(z["grad." + n_]).max() == 0:
            assert float(g.abs().max()) == 0.0, n_
        else:
            worst = max(worst, assert_close(g.cpu(), z["grad." + n_], 1e-4, "grad " + n_))
    print("worst relative parameter-gradient error vs the reference's recorded gradients", worst)


@pytest.mark.parametrize("name,light", [("g14_train_full", False), ("g14_train_full_light", True)])
def test_full_width_train_step_vs_reference_golden(golden, name, light):
    """synthetic.yml / synthetic_light_mask.yml networks (G14): the reference's depths and draws; every output and loss term of the
    reference's recorded step at 1e-4, and the gradient digest of the reference's own backward.

    Gradients.  Two things are ill-conditioned in the reference itself on a batch like this, and both are MEASURED in the test
    instead of being absorbed into a loose constant:
      * ~3 million ReLU units, ~20 of which have a pre-activation of zero within 1e-6: their backward mask (relu' = 0 or 1) is
        decided by rounding noise, and one flipped unit moves a bias gradient by ~1e-3 of its max-norm.  The test finds those
        units with the fp32 oracle (bitwise the reference's arithmetic), computes the exact gradient change of flipping each, and
        requires the difference to the reference's recorded gradients to be a 0/1 combination of at most 4 such flips plus a
        residual within tolerance;
      * rays that miss the geometry: alpha = 1 - exp(-E) underflows in fp32, the normal loss then normalises a vanishing sum
        (model/network/__init__.py:207-209).  The reference's recorded fp32 gradients differ from its own fp64 evaluation by 6e-4
        on the SDF net here; the residual tolerance per network is 3x the largest deviation among the fp64 oracle and two fp32
        oracle runs with rounding-level weight noise, never below 1e-4.
    The strict arithmetic check of the backward kernels against fp64 is test_train_step_given_depths_full_size /
    test_gpu_backward.py (well-conditioned seeds)."""
    from i2sdf_amd import synthetic_conf, I2SDFLoss
    from helpers import full_width_state_dict, measured_spread, memo, network_of, relu_flip_analysis, explain_by_relu_flips
    z = golden(name)
    ocfg, sd = full_width_state_dict(z, light)
    ocfg.use_normal = True
    sd["density.beta"] = torch.tensor(0.05)
    net = build(synthetic_conf(light), sd, train=True)
    inp = {k[3:]: t(z[k]) for k in z.files if k.startswith("in.")}
    gt = {k[3:]: t(z[k]) for k in z.files if k.startswith("gt.")}
    lk = {k: v for k, v in z["loss_kwargs"]}
    loss_fn = I2SDFLoss(**{k: (None if v == "None" else float(v)) for k, v in lk.items()})
    eng = net._engine_for("cuda:0")
    c, d, n = eng.ray_setup(inp["uv"].cuda(), inp["pose"].cuda(), inp["intrinsics"].cuda())
    out = net.render(cuda(inp), c, d, n, t(z["ref.z_vals"]).cuda(), t(z["ref.z_eik"]).cuda(),
                     draws={"eik_pts": t(z["draw.eik_pts"]).cuda(), "nbr_off": t(z["draw.nbr_off"]).cuda()})
    losses = loss_fn(out, cuda(gt), 10)
    net.zero_grad()
    losses["loss"].backward()
    hit = t(z["out.weight_sum"]).reshape(-1) > 1e-2
    for k in z.files:
        if k.startswith("out."):
            if k.endswith("normal_values"):
                assert_close(out[k[4:]].detach().cpu()[hit], t(z[k])[hit], 1e-4, k + " (weight_sum > 0.01)")
            else:
                assert_close(out[k[4:]].detach().cpu(), z[k], 1e-4, k)
        if k.startswith("loss.") and float(np.abs(z[k])) > 0:
            assert_close(losses[k[5:]].detach().cpu(), z[k], 1e-4, k)
    # ---- conditioning of the gradients on this fixture, measured with the oracle
    lc = _oracle_lc(lk)

    def run(sd_, dt, dev="cpu"):
        cast = lambda v: (v.to(dt) if v.dtype.is_floating_point else v).to(dev)
        dr = orc.Draws(eik_pts=cast(t(z["draw.eik_pts"])), nbr_off=cast(t(z["draw.nbr_off"])))
        _, _, g_ = orc.training_step_grads({k_: cast(v) for k_, v in sd_.items()}, ocfg, {k_: cast(v) for k_, v in inp.items()},
                                           {k_: cast(v) for k_, v in gt.items()}, lc, dr, step=10,
                                           z_override=(cast(t(z["ref.z_vals"])), cast(t(z["ref.z_eik"]))))
        return g_

    # oracle only: the same for both weight-gradient modes.  How far correct evaluations of the algorithm are apart does not depend on
    # where they run: the four of them as eager ops on the GPU (the mask analysis below stays on the host, fp32 = the reference's arithmetic)
    spread = memo(("g14 spread", name), lambda: measured_spread(lambda s_, d_: run(s_, d_, "cuda"), sd))
    by_net = {}
    for n_, v in spread.items():
        by_net[network_of(n_)] = max(by_net.get(network_of(n_), 0.0), v)
    print("measured conditioning (max deviation among fp64 / noisy-fp32 oracle runs) per network:", by_net)
    stride = int(z["grad_stride"])
    samp = lambda g_: (g_.detach().cpu().reshape(-1) if g_.numel() <= 1024 else g_.detach().cpu().reshape(-1)[::stride])
    def flips_of_the_oracle():
        r = relu_flip_analysis(lambda: run(sd, torch.float32), tau=1e-6)
        return r, relu_flip_analysis.pre
    (_, cands, deltas), oracle_pre = memo(("g14 flips", name), flips_of_the_oracle)
    err, scale = {}, {}
    for n_, p in net.named_parameters():
        g = p.grad if p.grad is not None else torch.zeros_like(p)
        scale[n_] = float(z["gmax." + n_])
        if scale[n_] == 0.0:
            assert float(g.abs().max()) == 0.0, n_
        err[n_] = samp(g).double() - t(z["gsample." + n_]).double().reshape(-1)
    raw = {n_: float(err[n_].abs().max()) / scale[n_] for n_ in err if scale[n_] > 0}
    chosen, res = explain_by_relu_flips(err, [{n_: samp(d[n_]) for n_ in d} for d in deltas], scale)
    print(f"{len(cands)} ReLU units with |pre-activation| < 1e-6; backward-mask flips that explain the difference to the reference:",
          [(cands[i][0], cands[i][1], cands[i][2], "%.1e" % cands[i][3]) for i in chosen])
    bad = []
    for n_ in sorted(res, key=lambda k_: -raw[k_])[:8]:
        print("  grad %-40s raw err %.2e   after flips %.2e   oracle spread %.2e" % (n_, raw[n_], res[n_], spread[n_]))
    for n_, r in res.items():
        tol = max(1e-4, 3.0 * by_net[network_of(n_)])
        if r > tol:
            bad.append((n_, r, tol))
    assert len(chosen) <= 4 and not bad, (chosen, bad)
    # ---- the explanation must be what the HIP path really did, not just a combination that happens to fit: read the library's own
    # saved radiance activations back (relu(a) > 0 is the mask its backward used) and compare mask by mask with the fp32 oracle run --
    # (i) masks differ ONLY at units whose pre-activation is zero within rounding, (ii) every flip the greedy search chose is one of them
    from helpers import hip_mask_flips
    node = out["rgb_values"].grad_fn                     # the autograd node of the render core keeps the saved tensors it used
    M_main = node.M_main
    rs_pm = eng.saved_to_point_major(node.rs, eng.blocked_points(1, M_main, node.rs.shape[1]))[:, :M_main]
    flips, worst = hip_mask_flips(rs_pm, oracle_pre)
    print(f"ReLU backward masks that differ between the HIP path and the fp32 oracle: {sorted(flips)} (largest |pre-activation| among them {worst:.1e})")
    assert worst < 1e-6, f"a backward mask differs at a unit whose pre-activation is {worst:.2e}: not a rounding-level flip"
    for i in chosen:
        assert (cands[i][0], cands[i][1], cands[i][2]) in flips, f"flip {cands[i]} explains the gradient difference but the HIP mask of that unit equals the oracle's"


@pytest.mark.wgrad_independent
def test_full_width_eval_vs_reference_golden(golden):
    """G15: synthetic.yml networks, eval.  (i) the reference's depths through render(): 1e-4; (ii) sampler in the loop: the
    iteration count is exact and outputs agree within the oracle's measured fp32-vs-fp64 spread."""
    from i2sdf_amd import synthetic_conf
    from helpers import full_width_state_dict
    z = golden("g15_eval_full")
    ocfg, sd = full_width_state_dict(z, False)
    sd["density.beta"] = torch.tensor(0.02)
    net = build(synthetic_conf(False), sd)
    inp = {k[3:]: t(z[k]) for k in z.files if k.startswith("in.")}
    eng = net._engine_for("cuda:0")
    c, d, n = eng.ray_setup(inp["uv"].cuda(), inp["pose"].cuda(), inp["intrinsics"].cuda())
    hit = t(z["out.weight_sum"]).reshape(-1) > 1e-2
    with torch.no_grad():
        out = net.render(cuda(inp), c, d, n, t(z["ref.z_vals"]).cuda(), t(z["ref.z_eik"]).cuda())
        for k in ("rgb_values", "depth_values", "weight_sum"):
            assert_close(out[k].cpu(), z["out." + k], 1e-4, k)
        assert_close(out["normal_map"].cpu()[hit], t(z["out.normal_map"])[hit], 1e-4, "normal_map (weight_sum > 0.01)")
        out = net(cuda(inp))
    assert int(net.last_sampler_iters.item()) == int(z["iters"])
    o32 = orc.network_forward(sd, ocfg, inp, training=False)
    o64 = orc.network_forward({k: v.double() for k, v in sd.items()}, ocfg, {k: v.double() for k, v in inp.items()}, training=False)
    keys = ("rgb_values", "depth_values", "weight_sum", "normal_map")
    spread = _spread(o32, o64, keys, hit)
    print("fp32-vs-fp64 spread of the oracle with its own sampler:", spread)
    for k in keys:
        tol = max(1e-4, 3.0 * spread[k])
        a, b = (out[k].cpu()[hit], t(z["out." + k])[hit]) if k == "normal_map" else (out[k].cpu(), t(z["out." + k]))
        assert_close(a, b, tol, k + " (sampler in the loop)")


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
    assert_close(out["normal_values"].detach().cpu()[hit], ref_out["normal_values"][hit], 1e-4, "normal_values (weight_sum > 0.01)")      # measured 7e-6
    assert_close(losses["loss"].detach().cpu(), ref_loss["loss"], 1e-5, "loss")
    worst = 0.0
    for n_, p in net.named_parameters():
        g = p.grad if p.grad is not None else torch.zeros_like(p)
        worst = max(worst, assert_close(g.cpu(), ref_g[n_], 1e-4, "grad " + n_))
    print("worst relative parameter-gradient error", worst)


def _oracle_fp64_on_gpu(sd, ocfg, inp, gt64, lc, d64, z_all, z_eik, step=10):
    """The fp64 restatement of a training step as stock PyTorch-ROCm eager ops on the GPU (the oracle is device-agnostic torch code; on
    the host's cores the same call takes ~8 s at 400 rays).  Returns CPU tensors."""
    D, g = torch.float64, (lambda v: v.cuda() if torch.is_tensor(v) else v)
    dr = orc.Draws(eik_pts=d64.eik_pts.cuda(), nbr_off=d64.nbr_off.cuda())
    out, losses, grads = orc.training_step_grads({k: v.to(D).cuda() for k, v in sd.items()}, ocfg, {k: v.to(D).cuda() for k, v in inp.items()},
                                                 {k: g(v) for k, v in gt64.items()}, lc, dr, step=step, z_override=(z_all.to(D).cuda(), z_eik.to(D).cuda()))
    c = lambda x: {k: (v.detach().cpu() if torch.is_tensor(v) else v) for k, v in x.items()}
    return c(out), c(losses), c(grads)


def test_train_step_bf16x3_matches_fp32_kernels(B=400):
    """The same training step (identical depths and draws) with every bf16x3 kernel enabled (the default) and with the plain
    fp32-MFMA kernels (`bf16x3: false`): outputs agree to 1e-5, and outputs and all parameter gradients of BOTH are within the 1e-4
    parity bar of the fp64 oracle.  Seeds / batch as chosen here are well conditioned.  Two things make ANY fp32 evaluation (the
    reference's included, bf16x3 or not) differ from fp64 by 1e-3 on unlucky batches, both measured with one-off probes in round 2:
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
    from helpers import memo
    ref_out, ref_loss, ref_g = memo(("x3 vs fp32 kernels: fp64 oracle", B), lambda: _oracle_fp64_on_gpu(sd, ocfg, inp, gt64, lc, d64, z_all, z_eik))
    for k in ("rgb_values", "depth_values", "weight_sum", "grad_theta"):
        assert_close(o3[k], o1[k], 1e-5, k + " (bf16x3 vs fp32 kernels)")
        assert_close(o3[k], ref_out[k], 1e-4, k + " (bf16x3 vs fp64)")
    assert abs(l3 - l1) <= 1e-6 * abs(l1)
    e3 = max(assert_close(g3[k], ref_g[k], 1e-4, "grad " + k + " (bf16x3 vs fp64)") for k in g1)
    e1 = max(assert_close(g1[k], ref_g[k], 1e-4, "grad " + k + " (fp32 kernels vs fp64)") for k in g1)
    print(f"worst relative parameter-gradient error vs fp64: bf16x3 kernels {e3:.2e}, fp32-MFMA kernels {e1:.2e}")


@pytest.mark.wgrad_independent
@pytest.mark.parametrize("B", [40, 400])
def test_wgrad_bf16x2_stays_inside_the_parity_bar(B):
    """I2SDF_OPT_WGRAD_BF16X2 (the default since round 4): the 256x256 weight-gradient blocks with two bf16 planes per operand and three products.
    Every parameter gradient of a full-width training step against the fp64 oracle: must stay within the 1e-4 bar, and the error
    is printed next to the bf16x3 (fp32-equivalent) form's.  B = 40 is the hard case (few points to average rounding over)."""
    from i2sdf_amd import synthetic_conf, I2SDFLoss
    ocfg = orc.synthetic_cfg(False)
    ocfg.use_normal = True
    sd = orc.perturb_params(orc.init_params(ocfg, seed=41), 0.03, seed=42)
    sd["density.beta"] = torch.tensor(0.05)
    inp = camera_inputs(B, (0.0, 0.0, -2.0), seed=5)
    gt = make_gt(B)
    dr = make_draws(ocfg, B, n_row=128, seed=2)
    cam, dirs, dn = orc.prepare_rays(inp["uv"], inp["pose"], inp["intrinsics"])
    z_all, z_eik = orc.sample_z_vals(sd, ocfg, dirs, cam, training=True, draws=dr, force_iters=1)
    D = torch.float64
    lc = orc.LossCfg(eikonal_weight=0.1, smooth_weight=0.01, smooth_iter=None, depth_weight=0.1, normal_weight=0.05)
    d64 = orc.Draws(eik_pts=dr.eik_pts.to(D), nbr_off=dr.nbr_off.to(D))
    gt64 = {k: (v.to(D) if v.dtype.is_floating_point else v) for k, v in gt.items()}
    _, _, ref_g = _oracle_fp64_on_gpu(sd, ocfg, inp, gt64, lc, d64, z_all, z_eik)
    worst = {}
    for x2 in (False, True):
        net = build(synthetic_conf(False), sd, train=True)
        eng = net._engine_for("cuda:0")
        eng.set_wgrad_bf16x2(x2)
        c, d, n = eng.ray_setup(inp["uv"].cuda(), inp["pose"].cuda(), inp["intrinsics"].cuda())
        out = net.render(cuda(inp), c, d, n, z_all.cuda(), z_eik.cuda(), draws={"eik_pts": dr.eik_pts.cuda(), "nbr_off": dr.nbr_off.cuda()})
        loss = I2SDFLoss(eikonal_weight=0.1, smooth_weight=0.01, smooth_iter=None, depth_weight=0.1, normal_weight=0.05)(out, cuda(gt), 10)["loss"]
        net.zero_grad()
        loss.backward()
        errs = {n_: rel_max(p.grad.cpu(), ref_g[n_]) for n_, p in net.named_parameters() if float(ref_g[n_].abs().max()) > 0}
        worst[x2] = max(errs.items(), key=lambda kv: kv[1])
        assert worst[x2][1] <= 1e-4, (x2, worst[x2])
    print(f"B={B}: worst parameter-gradient error vs fp64: bf16x3 wgrad {worst[False][1]:.2e} ({worst[False][0]}), bf16x2 wgrad {worst[True][1]:.2e} ({worst[True][0]})")
