"""GPU: training-mode forward kernels (SDF forward + explicit d sdf/dx chain + saved activations; radiance net)
vs the oracle's analytic restatement (fp64 arbiter)."""
import pytest
import torch

from oracle import i2sdf_oracle as orc
from helpers import assert_close, sd_from_npz, t

pytestmark = pytest.mark.gpu
TOL = 2e-5


def make_engine(conf, sd, parts=None):
    """parts: None = the engine's default (point ranges on, I2SDF_OPT_PARTS); 0 = full rounds + split-K tail workgroups."""
    from i2sdf_amd.config import NetConfig
    from i2sdf_amd.engine import RenderEngine
    eng = RenderEngine(NetConfig.from_conf(conf))
    if parts is not None:
        eng.set_parts(parts)
    eng.pack(eng.layout.flat_from_state_dict(sd).cuda())
    return eng


def dbl(sd):
    return {k: v.double() for k, v in sd.items()}


@pytest.mark.parametrize("which", ["synthetic", "light", "plumbing", "plumbing_skip"])
def test_sdf_forward_grad_and_saves(which):
    from i2sdf_amd.config import synthetic_conf, plumbing_conf
    if which == "synthetic":
        ocfg, conf = orc.synthetic_cfg(False), synthetic_conf(False)
    elif which == "light":
        ocfg, conf = orc.synthetic_cfg(True), synthetic_conf(True)
    elif which == "plumbing":
        ocfg, conf = orc.plumbing_cfg(False), plumbing_conf(False)
    else:
        ocfg, conf = orc.plumbing_cfg(True), plumbing_conf(True)
    sd = orc.perturb_params(orc.init_params(ocfg, seed=5), 0.05, seed=6)
    eng = make_engine(conf, sd)
    M = 777
    g = torch.Generator().manual_seed(1)
    x = (torch.rand(M, 3, generator=g) * 2 - 1) * 2.0
    fw = orc.sdf_analytic_forward(dbl(sd), ocfg.sdf, x.double())
    out = eng.sdf_forward_grad(points=x.cuda())
    L = ocfg.sdf.n_lin
    assert_close(out["sdf"].cpu(), fw["sdf"], TOL, "sdf")
    assert_close(out["feat"][:M].cpu(), fw["feat"], TOL, "feature")
    assert_close(out["grad"].cpu(), fw["n"], TOL, "d sdf/dx")
    hs_pm, ab_pm = eng.saved_to_point_major(out["hs"], out["blk"]), eng.saved_pm("abars", out["abars"], out["abars"].shape[1])
    for l in range(L - 1):
        ref_h = orc.softplus100(fw["a"][l])
        w = ref_h.shape[1]
        assert_close(hs_pm[l, :M, :w].cpu(), ref_h, TOL, f"h_{l+1}")
        assert_close(ab_pm[l, :M, :w].cpu(), fw["abar"][l], TOL, f"abar_{l}")


def test_sdf_forward_grad_ray_mode_matches_point_mode():
    from i2sdf_amd.config import plumbing_conf
    ocfg = orc.plumbing_cfg(True)
    sd = orc.perturb_params(orc.init_params(ocfg, seed=5), 0.05, seed=6)
    eng = make_engine(plumbing_conf(True), sd)
    g = torch.Generator().manual_seed(2)
    B, n = 37, 11
    cam = torch.randn(B, 3, generator=g) * 0.3
    dirs = torch.nn.functional.normalize(torch.randn(B, 3, generator=g), dim=1)
    z = torch.sort(torch.rand(B, n + 1, generator=g) * 4, -1)[0]
    zc = z.cuda()
    pts = (cam.cuda().unsqueeze(1) + zc[:, :n].unsqueeze(2) * dirs.cuda().unsqueeze(1)).reshape(-1, 3)
    a = eng.sdf_forward_grad(points=pts)
    b = eng.sdf_forward_grad(rays=(cam.cuda(), dirs.cuda(), zc[:, :n].contiguous()))
    for k in ("sdf", "grad"):
        assert_close(b[k], a[k], 1e-6, k)


@pytest.mark.parametrize("which", ["synthetic", "light", "plumbing"])
def test_rgb_forward(which):
    from i2sdf_amd.config import synthetic_conf, plumbing_conf
    if which == "plumbing":
        ocfg, conf = orc.plumbing_cfg(False), plumbing_conf(False)
    else:
        ocfg, conf = orc.synthetic_cfg(which == "light"), synthetic_conf(which == "light")
    sd = orc.perturb_params(orc.init_params(ocfg, seed=7), 0.05, seed=8)
    eng = make_engine(conf, sd)
    g = torch.Generator().manual_seed(3)
    B, n = 29, 13
    M, F = B * n, ocfg.rgb.feature_size
    dirs = torch.nn.functional.normalize(torch.randn(B, 3, generator=g), dim=1)
    feat = torch.randn(M, F, generator=g)
    ref = orc.rgb_forward(dbl(sd), ocfg.rgb, dirs.double().unsqueeze(1).repeat(1, n, 1).reshape(-1, 3), feat.double())
    Mp = eng.pad_rows(M)
    featp = torch.zeros(Mp, F)
    featp[:M] = feat
    rgb, rs, _ = eng.rgb_forward(dirs.cuda(), n, featp.cuda(), M)
    assert_close(rgb.cpu(), ref, TOL, "rgb")


def test_rgb_forward_golden(golden):
    from i2sdf_amd.config import plumbing_conf
    z = golden("g4_rgb")
    ocfg = orc.plumbing_cfg()
    sd = orc.init_params(ocfg)
    sd.update({"rendering_network." + k: v for k, v in sd_from_npz(z, "sd.").items()})
    eng = make_engine(plumbing_conf(), sd)
    M = z["feat"].shape[0]
    featp = torch.zeros(eng.pad_rows(M), 64)
    featp[:M] = t(z["feat"])
    rgb, _, _ = eng.rgb_forward(t(z["dirs"]).cuda(), 1, featp.cuda(), M)
    assert_close(rgb.cpu(), z["rgb"], 2e-5, "RenderingNetwork.forward")


@pytest.mark.parametrize("parts", [0, 4])
def test_rgb_forward_split_k_tail(parts):
    """M = 256 full workgroups + a short tail.  parts = 0: the tail runs through the split-K kernel (ksplit.h); parts = 4: four point
    ranges of full workgroups on their own streams (I2SDF_OPT_PARTS), ragged last workgroup.  Both against the oracle."""
    from i2sdf_amd.config import synthetic_conf
    ocfg = orc.synthetic_cfg(False)
    sd = orc.perturb_params(orc.init_params(ocfg, seed=7), 0.05, seed=8)
    eng = make_engine(synthetic_conf(False), sd, parts=parts)
    g = torch.Generator().manual_seed(9)
    n = 7
    B = (256 * 128 + 777 + n - 1) // n
    M = B * n
    dirs = torch.nn.functional.normalize(torch.randn(B, 3, generator=g), dim=1)
    feat = torch.randn(M, 256, generator=g)
    Mp = eng.pad_rows(M)
    featp = torch.zeros(Mp, 256)
    featp[:M] = feat
    rgb, rs, pev = eng.rgb_forward(dirs.cuda(), n, featp.cuda(), M)
    idx = torch.cat([torch.arange(0, 300), torch.arange(256 * 128 - 200, M)])          # some bulk points + the whole tail
    ref = orc.rgb_forward(dbl(sd), ocfg.rgb, dirs.double()[idx // n], feat.double()[idx])
    assert_close(rgb.cpu()[idx], ref, TOL, "rgb (bulk sample + split-K tail)")
    # saved activations of the tail (consumed by the backward / weight-gradient kernels)
    x = torch.cat([orc.positional_encode(dirs.double()[idx // n], 4), feat.double()[idx]], -1)
    W0 = orc.effective_weight(dbl(sd), "rendering_network.lin0")
    r1 = torch.relu(x @ W0.t() + sd["rendering_network.lin0.bias"].double())
    rs_pm = eng.saved_to_point_major(rs, eng.blocked_points(1, M, Mp))
    assert_close(rs_pm[0].cpu()[idx], r1, TOL, "r_1")
    assert_close(pev.cpu()[idx][:, :27], orc.positional_encode(dirs.double()[idx // n], 4), 1e-6, "PE(view)")


@pytest.mark.parametrize("light,parts", [(False, 0), (True, 0), (False, 4)])
def test_sdf_forward_grad_split_k_tail(light, parts):
    """256 full workgroups + a short tail (parts = 0: split-K workgroups; parts = 4: point ranges of full workgroups): sdf, feature,
    d sdf/dx and every saved tensor of the tail."""
    from i2sdf_amd.config import synthetic_conf
    ocfg = orc.synthetic_cfg(light)
    sd = orc.perturb_params(orc.init_params(ocfg, seed=5), 0.05, seed=6)
    eng = make_engine(synthetic_conf(light), sd, parts=parts)
    g = torch.Generator().manual_seed(11)
    M = 256 * 128 + 601
    x = (torch.rand(M, 3, generator=g) * 2 - 1) * 2.0
    out = eng.sdf_forward_grad(points=x.cuda())
    idx = torch.cat([torch.arange(0, 200), torch.arange(256 * 128 - 100, M)])
    fw = orc.sdf_analytic_forward(dbl(sd), ocfg.sdf, x.double()[idx])
    L = ocfg.sdf.n_lin
    assert_close(out["sdf"].cpu()[idx], fw["sdf"], TOL, "sdf")
    assert_close(out["feat"].cpu()[idx], fw["feat"], TOL, "feature")
    assert_close(out["grad"].cpu()[idx], fw["n"], TOL, "d sdf/dx")
    assert_close(out["pe"].cpu()[idx][:, :39], fw["p"], 1e-6, "PE")
    hs_pm, ab_pm = eng.saved_to_point_major(out["hs"], out["blk"]), eng.saved_pm("abars", out["abars"], out["abars"].shape[1])
    assert out["blk"] in ((0, 256 * 128) if parts == 0 else (out["Mp"],)), "blocked prefix = the points of the full workgroups"
    for l in range(L - 1):
        ref_h = orc.softplus100(fw["a"][l])
        wd = ref_h.shape[1]
        assert_close(hs_pm[l].cpu()[idx][:, :wd], ref_h, TOL, f"h_{l+1}")
        assert_close(ab_pm[l].cpu()[idx][:, :wd], fw["abar"][l], TOL, f"abar_{l}")
