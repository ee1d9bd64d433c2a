"""CPU: the oracle (oracle/i2sdf_oracle.py) against the committed golden vectors that
tests/golden/gen_golden.py produced by running the reference.
Tolerances: fp32 oracle vs fp32 reference, same op order up to reassociation -> 2e-5 max-norm relative
(the reference's own fp32-vs-fp64 noise floor is ~1e-6)."""
import numpy as np
import pytest
import torch

from oracle import i2sdf_oracle as orc
from helpers import t, sd_from_npz, assert_close

TOL = 2e-5


def test_g1_positional_encoding(golden):
    z = golden("g1_embed")
    x = t(z["x"])
    assert torch.equal(orc.positional_encode(x, 6), t(z["pe6"]))
    assert torch.equal(orc.positional_encode(x, 4), t(z["pe4"]))


@pytest.mark.parametrize("name,skip", [("g2_sdf", False), ("g3_sdf_skip", True)])
def test_g2_g3_sdf_forward_grad_double_backward(golden, name, skip):
    z = golden(name)
    cfg = orc.plumbing_cfg(skip=skip).sdf
    sd = sd_from_npz(z, "sd.")
    sd = {"implicit_network." + k: v for k, v in sd.items()}
    x = t(z["x"])
    assert_close(orc.sdf_forward(sd, cfg, x), z["out"], TOL, "sdf forward")
    params = {k: v.clone().requires_grad_(True) for k, v in sd.items()}
    sdf, feat, grad = orc.sdf_outputs(params, cfg, x, create_graph=True)
    assert_close(grad, z["grad"], TOL, "d sdf / d x")
    probe = ((grad.norm(2, dim=1) - 1) ** 2).sum() + sdf.sum() + (feat * t(z["feat_w"])).sum()
    assert_close(probe, z["probe"], TOL, "probe loss")
    probe.backward()
    for k, p in params.items():
        gk = "grad." + k[len("implicit_network."):]
        if gk in z.files:
            assert_close(p.grad, z[gk], 5e-5, gk)
    # analytic sweeps (what the HIP kernels implement) == autograd
    fw = orc.sdf_analytic_forward(sd, cfg, x)
    assert_close(fw["n"], z["grad"], TOL, "analytic n")
    eik = grad.detach()
    nbar = 2 * (eik.norm(2, dim=1, keepdim=True) - 1) * eik / eik.norm(2, dim=1, keepdim=True)
    grads, _, _ = orc.sdf_analytic_backward(sd, cfg, x, fw, torch.ones(x.shape[0], 1), t(z["feat_w"]), nbar)
    for k, gv in grads.items():
        gk = "grad." + k[len("implicit_network."):]
        if gk in z.files:
            assert_close(gv, z[gk], 5e-5, "analytic " + gk)
    # the form of the round-5 bf16x3 sweeps: G2 formed in sweep 2 from the stored G(hbar) -- the same gradients (fp64: to rounding)
    sd64 = {k: v.double() for k, v in sd.items()}
    fw64 = orc.sdf_analytic_forward(sd64, cfg, x.double())
    args64 = (torch.ones(x.shape[0], 1, dtype=torch.float64), t(z["feat_w"]).double(), nbar.double())
    ga, _, _ = orc.sdf_analytic_backward(sd64, cfg, x.double(), fw64, *args64)
    gb, _, _ = orc.sdf_analytic_backward(sd64, cfg, x.double(), fw64, *args64, g2_in_sweep2=True)
    for k in ga:
        assert_close(gb[k], ga[k], 1e-12, "G2 in sweep 2: " + k, floor=1e-30)
    g32, _, _ = orc.sdf_analytic_backward(sd, cfg, x, fw, torch.ones(x.shape[0], 1), t(z["feat_w"]), nbar, g2_in_sweep2=True)
    for k, gv in g32.items():
        gk = "grad." + k[len("implicit_network."):]
        if gk in z.files:
            assert_close(gv, z[gk], 5e-5, "analytic (G2 in sweep 2) " + gk)


def test_g4_radiance_net(golden):
    z = golden("g4_rgb")
    cfg = orc.plumbing_cfg().rgb
    sd = {"rendering_network." + k: v for k, v in sd_from_npz(z).items()}
    params = {k: v.clone().requires_grad_(True) for k, v in sd.items()}
    feat = t(z["feat"]).requires_grad_(True)
    rgb = orc.rgb_forward(params, cfg, t(z["dirs"]), feat)
    assert_close(rgb, z["rgb"], TOL, "rgb")
    (rgb * t(z["rgb_w"])).sum().backward()
    assert_close(feat.grad, z["feat_grad"], TOL, "feat grad")
    for k, p in params.items():
        assert_close(p.grad, z["grad." + k[len("rendering_network."):]], 5e-5, k)


def test_g5_density(golden):
    z = golden("g5_density")
    s = t(z["sdf"])
    for i, b in enumerate(z["betas"]):
        beta_p = torch.tensor(float(b), dtype=torch.float32, requires_grad=True)
        sr = s.clone().requires_grad_(True)
        sig = orc.laplace_density(sr, beta_p.abs() + 1e-4)
        assert_close(sig, z[f"sigma{i}"], 1e-6, "sigma")
        gs, gb = torch.autograd.grad(sig.sum(), [sr, beta_p])
        assert_close(gs, z[f"dsigma_ds{i}"], 1e-6, "dsigma/ds")
        assert_close(gb, z[f"dsum_dbeta{i}"], 1e-5, "dsigma/dbeta")
    so = torch.stack([s[:10, 0], s[10:20, 0]])
    assert_close(orc.laplace_density(so, t(z["beta_override"])), z["sigma_override"], 1e-6, "override")


def test_g6_volume_rendering_and_analytic_backward(golden):
    z = golden("g6_volume")
    zz, sdf = t(z["z"]), t(z["sdf"]).clone().requires_grad_(True)
    beta_p = t(z["beta_param"]).clone().requires_grad_(True)
    w, bg = orc.volume_weights(zz[:, :-1], zz[:, -1], sdf, beta_p.abs() + 1e-4)
    assert_close(w, z["weights"], TOL, "weights")
    assert_close(bg, z["bg_t"], TOL, "bg transmittance")
    gs, gb = torch.autograd.grad((w * t(z["w_w"])).sum() + bg.sum(), [sdf, beta_p])
    assert_close(gs, z["grad_sdf"], TOL, "grad sdf")
    assert_close(gb, z["grad_beta"], TOL, "grad beta")


def test_composite_backward_matches_autograd():
    g = torch.Generator().manual_seed(0)
    B, n = 24, 13
    zz = torch.sort(torch.rand(B, n + 1, generator=g, dtype=torch.float64) * 6, -1)[0]
    sdf = (torch.randn(B, n, generator=g, dtype=torch.float64) * 0.3).requires_grad_(True)
    rgb = torch.rand(B, n, 3, generator=g, dtype=torch.float64).requires_grad_(True)
    beta = torch.tensor(0.07, dtype=torch.float64, requires_grad=True)
    dn = torch.rand(B, generator=g, dtype=torch.float64) + 0.5
    o = orc.composite_forward(zz, sdf, rgb, None, dn, beta)
    gr, gd, gw = (torch.randn(B, 3, generator=g, dtype=torch.float64), torch.randn(B, generator=g, dtype=torch.float64),
                  torch.randn(B, 1, generator=g, dtype=torch.float64))
    L = (o["rgb"] * gr).sum() + (o["depth"] * gd).sum() + (o["wsum"] * gw).sum()
    a_s, a_c, a_b = torch.autograd.grad(L, [sdf, rgb, beta])
    s_bar, c_bar, b_bar = orc.composite_backward(zz, sdf.detach(), rgb.detach(), dn, beta.detach(), gr, gd, gw)
    assert_close(s_bar, a_s, 1e-10, "sdf bar")
    assert_close(c_bar, a_c, 1e-12, "rgb bar")
    assert_close(b_bar, a_b, 1e-10, "beta bar")


def test_g7b_error_bound(golden):
    z = golden("g7b_error_bound")
    zz, s, ds = t(z["z"]), t(z["sdf"]), t(z["d_star"])
    dists = zz[:, 1:] - zz[:, :-1]
    assert_close(orc.error_bound(torch.tensor(float(z["beta_scalar"])), s, dists, ds), z["eb_scalar"], TOL, "scalar beta")
    assert_close(orc.error_bound(t(z["beta_rows"]).unsqueeze(-1), s, dists, ds), z["eb_rows"], TOL, "row beta")


def _eval_inputs(tvec, B=1024, W=32, H=32, f=30.0):
    K = torch.eye(4)
    K[0, 0], K[1, 1], K[0, 2], K[1, 2] = f, f, W / 2, H / 2
    pose = torch.eye(4)
    pose[:3, 3] = t(tvec)
    ys, xs = torch.meshgrid(torch.arange(H), torch.arange(W), indexing="ij")
    uv = torch.stack([xs, ys], -1).float().reshape(1, -1, 2)[:, :B]
    return {"uv": uv, "intrinsics": K.unsqueeze(0), "pose": pose.unsqueeze(0)}


@pytest.mark.parametrize("tag", ["in", "out"])
def test_g7_g8_sampler_and_eval_forward(golden, tag):
    z = golden("g7_g8_eval")
    cfg = orc.plumbing_cfg()
    sd = sd_from_npz(z, "sd.")
    sd["density.beta"] = torch.tensor(float(z[f"{tag}.beta_param"]))
    inp = _eval_inputs(z[f"{tag}.t"])
    tr = orc.SamplerTrace()
    out = orc.network_forward(sd, cfg, inp, training=False, trace=tr)
    assert tr.iters == int(z[f"{tag}.iters"])
    assert_close(out["_z_vals"], z[f"{tag}.z_vals"], 1e-5, "z_vals")
    for k in ("rgb_values", "depth_values", "weight_sum"):
        assert_close(out[k], z[f"{tag}.out.{k}"], 1e-4, k)
    assert_close(out["normal_map"], z[f"{tag}.out.normal_map"], 2e-3, "normal_map")   # 3.8e-4 floor


@pytest.mark.parametrize("name,light", [("g9_train", False), ("g9_train_light", True)])
def test_g9_train_forward_loss_backward(golden, name, light):
    z = golden(name)
    cfg = orc.plumbing_cfg(skip=True, light=light)
    sd = sd_from_npz(z, "sd.")
    inp = {k[3:]: t(z[k]) for k in z.files if k.startswith("in.")}
    gt = {k[3:]: t(z[k]) for k in z.files if k.startswith("gt.")}
    dr = orc.Draws(**{k[5:]: t(z[k]) for k in z.files if k.startswith("draw.")})
    lk = {k: v for k, v in z["loss_kwargs"]}
    lc = orc.LossCfg(eikonal_weight=float(lk["eikonal_weight"]), smooth_weight=float(lk["smooth_weight"]), smooth_iter=None,
                     depth_weight=float(lk["depth_weight"]), normal_weight=float(lk["normal_weight"]),
                     bubble_weight=float(lk["bubble_weight"]), light_mask_weight=float(lk.get("light_mask_weight", 0.0)))
    out, losses, grads = orc.training_step_grads(sd, cfg, inp, gt, lc, dr, step=10)
    for k in z.files:
        if k.startswith("out."):
            tol = 2e-3 if k.endswith("normal_values") or k.endswith("diff_norm") else 1e-4
            assert_close(out[k[4:]], z[k], tol, k)
    for k in z.files:
        if k.startswith("loss."):
            assert_close(losses[k[5:]], z[k], 1e-4, k)
    for k in z.files:
        if k.startswith("grad."):
            assert_close(grads[k[5:]], z[k], 2e-3, k)


@pytest.mark.parametrize("name,light", [("g9_train", False), ("g9_train_light", True)])
def test_g9_train_step_given_reference_depths(golden, name, light):
    """Same as above but with the reference's OWN recorded depths (ref.z_vals): the ill-conditioned inverse-CDF depths are out
    of the comparison, and fp32 oracle vs fp32 reference must agree at fp32 rounding level on every output and gradient."""
    z = golden(name)
    cfg = orc.plumbing_cfg(skip=True, light=light)
    cfg.use_normal = True
    sd = sd_from_npz(z, "sd.")
    inp = {k[3:]: t(z[k]) for k in z.files if k.startswith("in.")}
    gt = {k[3:]: t(z[k]) for k in z.files if k.startswith("gt.")}
    dr = orc.Draws(eik_pts=t(z["draw.eik_pts"]), nbr_off=t(z["draw.nbr_off"]))
    lk = {k: v for k, v in z["loss_kwargs"]}
    lc = orc.LossCfg(eikonal_weight=float(lk["eikonal_weight"]), smooth_weight=float(lk["smooth_weight"]), smooth_iter=None,
                     depth_weight=float(lk["depth_weight"]), normal_weight=float(lk["normal_weight"]),
                     bubble_weight=float(lk["bubble_weight"]), light_mask_weight=float(lk.get("light_mask_weight", 0.0)))
    out, losses, grads = orc.training_step_grads(sd, cfg, inp, gt, lc, dr, step=10, z_override=(t(z["ref.z_vals"]), t(z["ref.z_eik"])))
    for k in z.files:
        if k.startswith("out."):
            assert_close(out[k[4:]], z[k], 2e-5, k)
        if k.startswith("loss."):
            assert_close(losses[k[5:]], z[k], 1e-5, k)
        if k.startswith("grad."):
            assert_close(grads[k[5:]], z[k], 5e-5, k)


@pytest.mark.parametrize("name,light", [("g14_train_full", False), ("g14_train_full_light", True)])
def test_g14_full_width_train_step(golden, name, light):
    """synthetic.yml / synthetic_light_mask.yml networks, the reference's depths and draws: outputs, loss, gradient digest."""
    from helpers import full_width_state_dict, assert_grad_digest
    z = golden(name)
    cfg, sd = full_width_state_dict(z, light)
    cfg.use_normal = True
    sd["density.beta"] = torch.tensor(0.05)
    inp = {k[3:]: t(z[k]) for k in z.files if k.startswith("in.")}
    gt = {k[3:]: t(z[k]) for k in z.files if k.startswith("gt.")}
    dr = orc.Draws(eik_pts=t(z["draw.eik_pts"]), nbr_off=t(z["draw.nbr_off"]))
    lk = {k: v for k, v in z["loss_kwargs"]}
    lc = orc.LossCfg(eikonal_weight=float(lk["eikonal_weight"]), smooth_weight=float(lk["smooth_weight"]), smooth_iter=None,
                     depth_weight=float(lk["depth_weight"]), normal_weight=float(lk["normal_weight"]),
                     bubble_weight=float(lk["bubble_weight"]), light_mask_weight=float(lk.get("light_mask_weight", 0.0)))
    out, losses, grads = orc.training_step_grads(sd, cfg, inp, gt, lc, dr, step=10, z_override=(t(z["ref.z_vals"]), t(z["ref.z_eik"])))
    for k in z.files:
        if k.startswith("out."):
            assert_close(out[k[4:]], z[k], 2e-5, k)
        if k.startswith("loss."):
            assert_close(losses[k[5:]], z[k], 1e-5, k)
    print("worst gradient-digest error", assert_grad_digest(z, grads, 5e-5))


def test_g15_full_width_eval(golden):
    from helpers import full_width_state_dict
    z = golden("g15_eval_full")
    cfg, sd = full_width_state_dict(z, False)
    sd["density.beta"] = torch.tensor(0.02)
    inp = {k[3:]: t(z[k]) for k in z.files if k.startswith("in.")}
    out = orc.network_forward(sd, cfg, inp, training=False, z_override=(t(z["ref.z_vals"]), t(z["ref.z_eik"])))
    for k in ("rgb_values", "depth_values", "weight_sum"):
        assert_close(out[k], z["out." + k], 2e-5, k)
    hit = t(z["out.weight_sum"]).reshape(-1) > 1e-2
    assert_close(out["normal_map"][hit], t(z["out.normal_map"])[hit], 1e-4, "normal_map (weight_sum > 0.01)")
    # and the sampler itself: iteration count exact
    tr = orc.SamplerTrace()
    cam, dirs, _ = orc.prepare_rays(inp["uv"], inp["pose"], inp["intrinsics"])
    orc.sample_z_vals(sd, cfg, dirs, cam, training=False, trace=tr)
    assert tr.iters == int(z["iters"])


def test_g10_camera(golden):
    z = golden("g10_camera")
    d, c = orc.get_camera_params(t(z["uv"]), t(z["pose"]), t(z["intrinsics"]))
    assert_close(d, z["ray_dirs"], 1e-6, "ray dirs")
    assert torch.equal(c, t(z["cam_loc"]))


def test_g10b_camera_quaternion(golden):
    z = golden("g10b_camera_quat")
    d, c = orc.get_camera_params(t(z["uv"]), t(z["pose"]), t(z["intrinsics"]))
    assert_close(d, z["ray_dirs"], 1e-6, "ray dirs (quaternion pose)")
    assert torch.equal(c, t(z["cam_loc"]))


def test_g12_sphere_intersections(golden):
    z = golden("g12_sphere")
    assert_close(orc.get_sphere_intersections(t(z["cam_loc"]), t(z["dirs"]), float(z["r"])), z["t"], 1e-6, "sphere intersections")
    with pytest.raises(ValueError):
        orc.get_sphere_intersections(torch.tensor([[5.0, 0, 0]]), torch.tensor([[0.0, 1, 0]]), 3.0)


def test_g13_batcher(golden):
    """ReconDataset.__getitem__ + collate_fn for 40 global pixel indices over 3 images."""
    z = golden("g13_batcher")
    tables = {k[4:]: t(z[k]) for k in z.files if k.startswith("tab.")}
    tidx, idx, sample, gt = orc.ray_batch(tables, [int(v) for v in z["img_res"]], t(z["tidx"]))
    assert torch.equal(idx, t(z["image_idx"]))
    for k in ("uv", "intrinsics", "pose"):
        assert torch.equal(sample[k], t(z["sample." + k])), k
    names = [k[3:] for k in z.files if k.startswith("gt.")]
    assert sorted(names) == sorted(gt)
    for k in names:
        assert torch.equal(gt[k], t(z["gt." + k])), k
    d, c = orc.get_camera_params(sample["uv"], sample["pose"], sample["intrinsics"])
    assert_close(d, z["ray_dirs"], 1e-6, "ray dirs")


def test_g11_loss(golden):
    z = golden("g11_loss")
    out = {k[4:]: t(z[k]) for k in z.files if k.startswith("out.")}
    gt = {k[3:]: t(z[k]) for k in z.files if k.startswith("gt.")}
    # config/synthetic.yml:15-23 at step 160000 (smooth on; bubble_weight still applies to surface_sdf)
    lc1 = orc.LossCfg(eikonal_weight=0.1, smooth_weight=0.01, smooth_iter=150000, depth_weight=0.1, normal_weight=0.05, bubble_weight=0.5)
    l1 = orc.i2sdf_loss(out, gt, lc1, 160000)
    lc2 = orc.LossCfg(eikonal_weight=0.1, smooth_weight=0.01, smooth_iter=150000, depth_weight=0.1, normal_weight=0.05, bubble_weight=0.5,
                      light_mask_weight=0.5)
    l2 = orc.i2sdf_loss(out, gt, lc2, 60000)
    for k in l1:
        assert_close(l1[k], z["synthetic." + k], 1e-6, "synthetic." + k)
        assert_close(l2[k], z["light." + k], 1e-6, "light." + k)


# ---- G16: marching-cubes grids and the bubble-PDF update -------------------------------------------------
def test_g16_grids_oracle_and_host_axes(golden):
    """The oracle's restatement of get_grid_uniform / get_grid AND the package's host-side axis helpers against the reference's
    own output (axes exactly; points exactly -- same numpy arithmetic)."""
    from i2sdf_amd import grid as G
    z = golden("g16_grid_pdf")
    res, bnd = int(z["uni.resolution"]), z["uni.boundary"]
    o = orc.get_grid_uniform(res, list(bnd))
    ax = G.uniform_axes(res, bnd)
    for a in range(3):
        assert np.array_equal(o["xyz"][a], z[f"uni.xyz{a}"]) and np.array_equal(ax.xyz[a], z[f"uni.xyz{a}"])
    assert torch.equal(o["grid_points"], t(z["uni.grid_points"]))
    for case in range(3):
        pts, res = t(z[f"al{case}.points"]), int(z[f"al{case}.resolution"])
        o = orc.get_grid(pts, res)
        ax = G.aligned_axes(pts, res)
        assert o["shortest_axis_index"] == ax.shortest_axis_index == int(z[f"al{case}.shortest_axis_index"]) == case
        assert o["shortest_axis_length"] == ax.shortest_axis_length == float(z[f"al{case}.shortest_axis_length"])
        for a in range(3):
            assert np.array_equal(o["xyz"][a], z[f"al{case}.xyz{a}"]), (case, a)
            assert np.array_equal(ax.xyz[a], z[f"al{case}.xyz{a}"]), (case, a)
        assert torch.equal(o["grid_points"], t(z[f"al{case}.grid_points"]))
        # explicit bounds instead of a point set
        ax2 = G.aligned_axes(None, res, pts.min(0).values.numpy(), pts.max(0).values.numpy())
        assert all(np.array_equal(ax2.xyz[a], ax.xyz[a]) for a in range(3))
        assert ax.shape_volume == tuple(z[f"al{case}.xyz{a}"].shape[0] for a in range(3))


def test_g16_update_pdf_oracle(golden):
    z = golden("g16_grid_pdf")
    links, idx = t(z["pdf.pointlinks"]), t(z["pdf.idx"])
    for tag, crit in (("rgb", "RGB"), ("rgb_mp", "RGB"), ("depth", "DEPTH"), ("depth_mp", "DEPTH")):
        pmax = None if np.isnan(z[f"pdf.{tag}.max"]) else float(z[f"pdf.{tag}.max"])
        pdf = torch.full((int(z["pdf.n_points"]),), -1.0)
        value = orc.pdf_error(crit, {"rgb_values": t(z["pdf.rgb_pred"]), "depth_values": t(z["pdf.depth_pred"])},
                              {"rgb": t(z["pdf.rgb_gt"]), "depth": t(z["pdf.depth_gt"])})
        orc.update_pdf(pdf, value, idx, links, pmax, float(z[f"pdf.{tag}.prune"]))
        assert torch.equal(pdf, t(z[f"pdf.{tag}.out"])), tag
        assert (pdf == -1).any() and (pdf == 0).any() == (tag.endswith("_mp"))      # untouched points stay; pruning only with a threshold
