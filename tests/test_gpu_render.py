"""GPU: ray set-up and density/compositing kernels (forward + analytic backward) vs the oracle and the golden vectors."""
import pytest
import torch

from oracle import i2sdf_oracle as orc
from helpers import assert_close, t

pytestmark = pytest.mark.gpu


def _engine():
    from i2sdf_amd.config import NetConfig, plumbing_conf
    from i2sdf_amd.engine import RenderEngine
    return RenderEngine(NetConfig.from_conf(plumbing_conf()))


def test_ray_setup_golden(golden):
    z = golden("g10_camera")
    eng = _engine()
    cam, dirs, dn = eng.ray_setup(t(z["uv"]).cuda(), t(z["pose"]).cuda(), t(z["intrinsics"]).cuda())
    raw = t(z["ray_dirs"]).reshape(-1, 3)
    assert_close(dn.cpu(), raw.norm(dim=1), 1e-6, "||d||")
    assert_close(dirs.cpu(), torch.nn.functional.normalize(raw, dim=1), 1e-6, "dirs")
    assert torch.equal(cam.cpu(), t(z["cam_loc"]))


def test_ray_setup_eval_layout():
    eng = _engine()
    g = torch.Generator().manual_seed(0)
    P = 333
    uv = torch.rand(1, P, 2, generator=g) * 400
    K = torch.eye(4).unsqueeze(0).clone(); K[0, 0, 0] = 500; K[0, 1, 1] = 480; K[0, 0, 2] = 320; K[0, 1, 2] = 240; K[0, 0, 1] = 1.5
    pose = torch.eye(4).unsqueeze(0).clone(); pose[0, :3, 3] = torch.tensor([0.3, -0.2, 0.9])
    c0, d0, n0 = orc.prepare_rays(uv.double(), pose.double(), K.double())
    cam, dirs, dn = eng.ray_setup(uv.cuda(), pose.cuda(), K.cuda())
    assert_close(cam.cpu(), c0, 1e-7, "cam"); assert_close(dirs.cpu(), d0, 1e-6, "dirs"); assert_close(dn.cpu(), n0, 1e-6, "norm")


@pytest.mark.parametrize("n", [17, 97, 130])
def test_composite_forward_backward(n):
    eng = _engine()
    g = torch.Generator().manual_seed(n)
    B = 50
    zz = torch.sort(torch.rand(B, n + 1, generator=g) * 6, -1)[0]
    sdf = torch.randn(B, n, generator=g) * 0.3
    rgb = torch.rand(B, n, 3, generator=g)
    grd = torch.randn(B, n, 3, generator=g)
    lm = torch.rand(B, n, generator=g)
    dn = torch.rand(B, generator=g) + 0.5
    beta_p = torch.tensor([-0.07])                       # negative on purpose: beta = |beta_p| + beta_min
    D = torch.float64
    sd_, rg_, gr_, bp_ = (sdf.to(D).requires_grad_(True), rgb.to(D).requires_grad_(True), grd.to(D).requires_grad_(True),
                          beta_p.to(D).requires_grad_(True))
    beta = bp_.abs() + 1e-4
    ref = orc.composite_forward(zz.to(D), sd_, rg_, gr_, dn.to(D), beta)
    wdet = ref["w"].detach()
    nh = torch.nn.functional.normalize(gr_, dim=-1)
    normal_train = torch.nn.functional.normalize((wdet.unsqueeze(-1) * nh).sum(1), dim=-1)
    lmask_ref = (wdet * lm.to(D)).sum(1, keepdim=True)
    o = eng.composite_forward(beta_p.cuda(), zz.cuda(), sdf.reshape(-1).cuda(), rgb.reshape(-1, 3).cuda(), grd.reshape(-1, 3).cuda(),
                              lm.reshape(-1).cuda(), dn.cuda(), want_normal=True)
    assert_close(o["w"].cpu(), ref["w"], 1e-5, "weights")
    assert_close(o["rgb"].cpu(), ref["rgb"], 1e-5, "rgb")
    assert_close(o["depth"].cpu(), ref["depth"], 1e-5, "depth")
    assert_close(o["wsum"].cpu(), ref["wsum"], 1e-5, "wsum")
    assert_close(o["normal"].cpu(), ref["normal"], 1e-5, "normal")
    assert_close(o["lmask"].cpu(), lmask_ref, 1e-5, "lmask")
    g_rgb, g_d, g_w, g_n, g_l = (torch.randn(B, 3, generator=g), torch.randn(B, generator=g), torch.randn(B, 1, generator=g),
                                 torch.randn(B, 3, generator=g), torch.randn(B, 1, generator=g))
    lm_ = lm.to(D).requires_grad_(True)
    loss = ((ref["rgb"] * g_rgb.to(D)).sum() + (ref["depth"] * g_d.to(D)).sum() + (ref["wsum"] * g_w.to(D)).sum()
            + (normal_train * g_n.to(D)).sum() + ((wdet * lm_).sum(1, keepdim=True) * g_l.to(D)).sum())
    a_s, a_c, a_g, a_b, a_l = torch.autograd.grad(loss, [sd_, rg_, gr_, bp_, lm_])
    acc = torch.zeros(1, device="cuda")
    bo = eng.composite_backward(beta_p.cuda(), zz.cuda(), sdf.reshape(-1).cuda(), rgb.reshape(-1, 3).cuda(), grd.reshape(-1, 3).cuda(),
                                dn.cuda(), o["nsum"], g_rgb.cuda(), g_d.cuda(), g_w.reshape(-1).cuda(), g_n.cuda(), g_l.reshape(-1).cuda(),
                                beta_grad_accum=acc)
    assert_close(bo["sdf_bar"].cpu().reshape(B, n), a_s, 2e-5, "sdf_bar")
    assert_close(bo["rgb_bar"].cpu().reshape(B, n, 3), a_c, 1e-5, "rgb_bar")
    assert_close(bo["grad_bar"].cpu().reshape(B, n, 3), a_g, 2e-5, "grad_bar")
    assert_close(bo["lmask_bar"].cpu().reshape(B, n), a_l, 1e-5, "lmask_bar")
    assert_close(acc.cpu(), a_b, 1e-4, "beta grad")


def test_volume_rendering_golden(golden):
    z = golden("g6_volume")
    eng = _engine()
    zz = t(z["z"]); B, n = zz.shape[0], zz.shape[1] - 1
    o = eng.composite_forward(t(z["beta_param"]).reshape(1).cuda(), zz.cuda(), t(z["sdf"]).reshape(-1).cuda(), torch.zeros(B * n, 3).cuda(),
                              None, None, torch.ones(B).cuda(), want_normal=False)
    assert_close(o["w"].cpu(), z["weights"], 1e-5, "volume_rendering weights")
