"""GPU: fused I2SDFLoss (value + gradients in HIP) vs the fp64 oracle loss and the reference's golden values."""
import pytest
import torch

from helpers import assert_close, t

pytestmark = pytest.mark.gpu


def _rand_case(B, n_pc, light, seed):
    g = torch.Generator().manual_seed(seed)
    out = {"rgb_values": torch.rand(B, 3, generator=g), "depth_values": torch.rand(B, generator=g) * 3, "weight_sum": torch.rand(B, 1, generator=g),
           "grad_theta": torch.randn(2 * B, 3, generator=g), "diff_norm": torch.rand(B, generator=g),
           "normal_values": torch.nn.functional.normalize(torch.randn(B, 3, generator=g), dim=1), "surface_sdf": torch.randn(n_pc, 1, generator=g) * 0.1}
    gt = {"rgb": torch.rand(B, 3, generator=g), "depth": torch.rand(B, generator=g) * 3, "depth_mask": torch.rand(B, generator=g) > 0.3,
          "normal": torch.nn.functional.normalize(torch.randn(B, 3, generator=g), dim=1), "normal_mask": torch.rand(B, generator=g) > 0.3,
          "mask": (torch.rand(B, 1, generator=g) > 0.5).float()}
    if light:
        out["light_mask"] = torch.rand(B, 1, generator=g)
        gt["light_mask"] = (torch.rand(B, 1, generator=g) > 0.5).float()
    return out, gt


@pytest.mark.parametrize("light,kw,step", [
    (False, dict(eikonal_weight=0.1, smooth_weight=0.01, smooth_iter=150000, depth_weight=0.1, normal_weight=0.05, bubble_weight=0.5,
                 min_bubble_iter=50000, max_bubble_iter=150000), 160000),
    (True, dict(eikonal_weight=0.1, smooth_weight=0.01, smooth_iter=None, depth_weight=0.1, normal_weight=0.05, light_mask_weight=0.5,
                mask_weight=0.3), 10),
    (False, dict(eikonal_weight=0.1, depth_weight=0.0, normal_weight=0.0, angular_weight=0.0), 10),
])
def test_fused_loss_matches_oracle(light, kw, step):
    """values and the gradient w.r.t. every network output vs the fp64 oracle loss (oracle.i2sdf_loss + torch.autograd on CPU)."""
    from i2sdf_amd import I2SDFLoss
    from oracle import i2sdf_oracle as orc
    out, gt = _rand_case(777, 41, light, seed=3)
    fused = I2SDFLoss(**kw)
    lc = orc.LossCfg(**{k: v for k, v in kw.items() if k in orc.LossCfg.__dataclass_fields__})
    lc.smooth_iter = fused.smooth_iter
    o1 = {k: v.cuda().requires_grad_(True) for k, v in out.items()}
    o2 = {k: v.double().requires_grad_(True) for k, v in out.items()}
    gtc = {k: v.cuda() for k, v in gt.items()}
    gt64 = {k: (v.double() if v.dtype.is_floating_point else v) for k, v in gt.items()}
    l1, l2 = fused(o1, gtc, step), orc.i2sdf_loss(o2, gt64, lc, step)
    for k in l2:
        assert_close(l1[k].detach().cpu(), l2[k].detach(), 2e-6, k)
    (l1["loss"] * 1.7).backward()
    (l2["loss"] * 1.7).backward()
    for k in o1:
        g2 = o2[k].grad if o2[k].grad is not None else torch.zeros_like(o2[k])
        g1 = o1[k].grad if o1[k].grad is not None else torch.zeros_like(o1[k])
        if g2.abs().max() == 0:
            assert g1.abs().max() == 0, k
        else:
            assert_close(g1.cpu(), g2, 1e-5, "grad " + k)


def test_fused_loss_golden(golden):
    from i2sdf_amd import I2SDFLoss
    z = golden("g11_loss")
    out = {k[4:]: t(z[k]).cuda() for k in z.files if k.startswith("out.")}
    gt = {k[3:]: t(z[k]).cuda() for k in z.files if k.startswith("gt.")}
    kw = dict(eikonal_weight=0.1, smooth_weight=0.01, smooth_iter=150000, depth_weight=0.1, normal_weight=0.05, bubble_weight=0.5,
              min_bubble_iter=50000, max_bubble_iter=150000)
    l1 = I2SDFLoss(**kw)(out, gt, 160000)
    l2 = I2SDFLoss(light_mask_weight=0.5, **kw)(out, gt, 60000)
    for k in l1:
        assert_close(l1[k].cpu(), z["synthetic." + k], 2e-6, "synthetic." + k)
        assert_close(l2[k].cpu(), z["light." + k], 2e-6, "light." + k)


def test_extra_points_and_backward_seeds_match_the_torch_glue_bitwise():
    """i2sdf_extra_points / i2sdf_backward_seeds replace a multiply, two adds, a concatenation / three fills and two copies of torch
    glue (model/network/__init__.py:175-186 and autograd's accumulation): same bits."""
    import torch
    from i2sdf_amd import lib as L
    lib = L.load()
    dev = torch.device("cuda:0")
    g = torch.Generator().manual_seed(3)
    for B in (1, 77, 1024):
        cam, dirs = torch.randn(B, 3, generator=g).to(dev), torch.nn.functional.normalize(torch.randn(B, 3, generator=g), dim=1).to(dev)
        z, eik, off = (torch.rand(B, 1, generator=g) * 5).to(dev), (torch.rand(B, 3, generator=g) * 6 - 3).to(dev), ((torch.rand(B, 3, generator=g) - 0.5) * 0.01).to(dev)
        out = torch.full((3 * B + 5, 3), 7.0, device=dev)
        L.check(lib.i2sdf_extra_points(L.ptr(cam), L.ptr(dirs), L.ptr(z), L.ptr(eik), L.ptr(off), B, L.ptr(out), L.stream_ptr()), "extra_points")
        near = cam + z * dirs
        assert torch.equal(out[:3 * B], torch.cat([eik, near, near + off], 0))
        assert torch.equal(out[3 * B:], torch.full((5, 3), 7.0, device=dev)), "rows beyond 3B must be left alone"
    for (Mm, n_eik, n_pc, tail, n_beta, zero_main, with_e, with_s) in ((0, 0, 0, 0, 1, 0, False, False), (640, 96, 0, 32, 3, 0, True, False),
                                                                         (640, 96, 17, 15, 1, 1, True, True), (99328, 3072, 0, 0, 1, 0, True, False),
                                                                         (256, 30, 9, 3, 2, 1, False, True)):
        Ms = Mm + n_eik + n_pc + tail
        beta = torch.full((n_beta + 2,), 5.0, device=dev)
        sbar, nbar = torch.full((max(Ms, 1),), 3.0, device=dev), torch.full((max(Ms, 1), 3), 4.0, device=dev)
        ge = torch.randn(n_eik, 3, generator=g).to(dev) if with_e else None
        gs = torch.randn(n_pc, generator=g).to(dev) if with_s else None
        L.check(lib.i2sdf_backward_seeds(L.ptr(beta), n_beta, L.ptr(sbar), L.ptr(nbar), Mm, Ms, L.ptr(ge), n_eik, L.ptr(gs), n_pc, zero_main,
                                         L.stream_ptr()), "backward_seeds")
        assert torch.equal(beta, torch.cat([torch.zeros(n_beta, device=dev), torch.full((2,), 5.0, device=dev)]))
        es, en = torch.full((max(Ms, 1),), 3.0, device=dev), torch.full((max(Ms, 1), 3), 4.0, device=dev)
        if Ms > 0:
            es[Mm:Ms] = 0.0
            en[Mm:Ms] = 0.0
            if zero_main:
                en[:Mm] = 0.0
            if with_e:
                en[Mm:Mm + n_eik] = ge
            if with_s:
                es[Mm + n_eik:Mm + n_eik + n_pc] = gs
        assert torch.equal(sbar, es) and torch.equal(nbar, en), (Mm, n_eik, n_pc)
    # argument checks
    assert lib.i2sdf_backward_seeds(None, 1, None, None, 0, 0, None, 0, None, 0, 0, L.stream_ptr()) != 0
    assert lib.i2sdf_backward_seeds(L.ptr(beta), 1, L.ptr(sbar), L.ptr(nbar), 10, 12, None, 3, None, 0, 0, L.stream_ptr()) != 0      # n_eik > extra rows


# ---- round 6: loss + render backward fused (i2sdf_render_loss_backward) against the separate path, same process, same draws -------------
def _step(net, loss_fn, inp, gt, draws, step, fused, scale=1.0, extra=None):
    """one forward + loss + backward; returns (loss dict, flat gradient copy).  fused=False forces the separate path."""
    import os
    os.environ["I2SDF_FUSED_RENDER_LOSS"] = "1" if fused else "0"
    try:
        for p in net.parameters():
            p.grad = None
        out = net(inp, draws=draws)
        res = loss_fn(out, gt, step)
        tot = res["loss"] * scale
        if extra is not None:
            tot = tot + extra(out)
        tot.backward()
        g = torch.cat([p.grad.reshape(-1) for p in net.parameters()]).clone()
        return {k: v.detach().clone() for k, v in res.items()}, g, out
    finally:
        os.environ.pop("I2SDF_FUSED_RENDER_LOSS", None)


@pytest.mark.parametrize("light,n_pc,step,scale", [(False, 0, 10, 1.0), (True, 0, 10, 0.37), (False, 37, 60000, 1.0), (False, 0, 200000, 2.5)])
def test_fused_render_loss_equals_the_separate_path(light, n_pc, step, scale, wgrad_mode):
    """The module's default path -- I2SDFLoss recognises the outputs of a training render and runs loss + render backward as one fused library
    call -- against the separate entry points (I2SDF_FUSED_RENDER_LOSS=0): same reported values, same parameter gradients (the per-sample
    gradients are computed by the same device functions; sums are taken in a different but fixed order -> 1e-6), with a light head, with a
    bubble point cloud (step inside the bubble window), with the smoothness term active (step behind smooth_iter) and with an upstream
    gradient that is not 1."""
    from i2sdf_amd import I2SDFNetwork, I2SDFLoss, synthetic_conf
    from helpers import camera_inputs, make_gt
    conf = synthetic_conf(light)
    conf["use_normal"] = True
    torch.manual_seed(5)
    net = I2SDFNetwork(conf).cuda().train()
    with torch.no_grad():
        net.density.beta.fill_(0.05)
    B = 203
    inp = {k: v.cuda() for k, v in camera_inputs(B, (0.0, 0.0, -2.0), seed=11).items()}
    if n_pc:
        inp["pointcloud"] = (torch.rand(n_pc, 3, device="cuda") * 2 - 1) * 0.7
    gt = {k: v.cuda() for k, v in make_gt(B).items()}
    gt["depth_mask"][::3] = False
    gt["normal_mask"][1::4] = False
    if light:
        gt["light_mask"] = (torch.rand(B, 1, device="cuda") > 0.5).float()
    loss_fn = I2SDFLoss(eikonal_weight=0.1, smooth_weight=0.01, smooth_iter=150000, depth_weight=0.1, normal_weight=0.05, bubble_weight=0.5 if n_pc else 0.0,
                        min_bubble_iter=50000, max_bubble_iter=150000, light_mask_weight=0.5 if light else 0.0)
    eng = net._engine_for(torch.device("cuda:0"))
    draws = {k: v for k, v in eng.training_draws(B, 1234, "cuda", net.scene_bounding_sphere).items() if v is not None}
    net.force_iters = 2
    r_sep, g_sep, _ = _step(net, loss_fn, inp, gt, draws, step, fused=False, scale=scale)
    r_fus, g_fus, out = _step(net, loss_fn, inp, gt, draws, step, fused=True, scale=scale)
    assert getattr(out["rgb_values"], "_i2sdf_render", None) is not None
    for k in r_sep:
        assert_close(r_fus[k], r_sep[k], 2e-6, f"loss term {k}", floor=1e-6)
    assert torch.isfinite(g_fus).all()
    assert_close(g_fus, g_sep, 2e-6, "parameter gradients, fused vs separate")


def test_fused_render_loss_with_other_consumers_of_the_outputs(wgrad_mode):
    """Another differentiable term on the same outputs: the loss's seeds reach _RenderFn.backward summed with foreign gradients -- the fused
    path must notice (its placeholders do not arrive untouched), correct for the upstream scale and give what the separate path gives."""
    from i2sdf_amd import I2SDFNetwork, I2SDFLoss, synthetic_conf
    from helpers import camera_inputs, make_gt
    conf = synthetic_conf(False)
    conf["use_normal"] = True
    torch.manual_seed(6)
    net = I2SDFNetwork(conf).cuda().train()
    B = 64
    inp = {k: v.cuda() for k, v in camera_inputs(B, (0.0, 0.0, -2.0), seed=12).items()}
    gt = {k: v.cuda() for k, v in make_gt(B).items()}
    loss_fn = I2SDFLoss(eikonal_weight=0.1, smooth_weight=0.01, smooth_iter=None, depth_weight=0.1, normal_weight=0.05)
    eng = net._engine_for(torch.device("cuda:0"))
    draws = {k: v for k, v in eng.training_draws(B, 99, "cuda", net.scene_bounding_sphere).items() if v is not None}
    net.force_iters = 1
    for extra in (lambda o: 0.1 * o["rgb_values"].sum() + 0.05 * o["depth_values"].mean(),
                  lambda o: 0.02 * o["grad_theta"].pow(2).sum(),
                  lambda o: 0.3 * o["diff_norm"].sum() + 0.1 * o["normal_values"][:, 0].sum()):
        _, g_sep, _ = _step(net, loss_fn, inp, gt, draws, 10, fused=False, scale=0.7, extra=extra)
        _, g_fus, _ = _step(net, loss_fn, inp, gt, draws, 10, fused=True, scale=0.7, extra=extra)
        assert_close(g_fus, g_sep, 5e-6, "parameter gradients with a foreign term, fused vs separate")


def test_fused_render_loss_called_twice_on_the_same_outputs(wgrad_mode):
    """Two loss calls on one render's outputs (e.g. a logging call and the training call with other weights), backward through a weighted
    sum of both: the first call's prepared gradients are superseded by the second's -- its node must then give autograd real, scaled
    gradients -- and the result must equal the separate path's."""
    from i2sdf_amd import I2SDFNetwork, I2SDFLoss, synthetic_conf
    from helpers import camera_inputs, make_gt
    import os
    conf = synthetic_conf(False)
    conf["use_normal"] = True
    torch.manual_seed(8)
    net = I2SDFNetwork(conf).cuda().train()
    B = 48
    inp = {k: v.cuda() for k, v in camera_inputs(B, (0.0, 0.0, -2.0), seed=13).items()}
    gt = {k: v.cuda() for k, v in make_gt(B).items()}
    l1 = I2SDFLoss(eikonal_weight=0.1, depth_weight=0.1, normal_weight=0.05)
    l2 = I2SDFLoss(eikonal_weight=0.3, depth_weight=0.0, normal_weight=0.2, smooth_weight=0.05, smooth_iter=None)
    eng = net._engine_for(torch.device("cuda:0"))
    draws = {k: v for k, v in eng.training_draws(B, 5, "cuda", net.scene_bounding_sphere).items() if v is not None}
    net.force_iters = 1
    res = []
    for fused in (False, True):
        os.environ["I2SDF_FUSED_RENDER_LOSS"] = "1" if fused else "0"
        try:
            for p in net.parameters():
                p.grad = None
            out = net(inp, draws=draws)
            a, b = l1(out, gt, 10)["loss"], l2(out, gt, 10)["loss"]
            (0.6 * a + 1.7 * b).backward()
            res.append((torch.cat([p.grad.reshape(-1) for p in net.parameters()]).clone(), a.detach().clone(), b.detach().clone()))
        finally:
            os.environ.pop("I2SDF_FUSED_RENDER_LOSS", None)
    assert_close(res[1][1], res[0][1], 2e-6, "first loss value"); assert_close(res[1][2], res[0][2], 2e-6, "second loss value")
    assert_close(res[1][0], res[0][0], 5e-6, "parameter gradients of a weighted sum of two loss calls, fused vs separate")
