"""GPU: MLP backward kernels (sweeps + weight-gradient GEMMs + weight-norm backward) vs the oracle's autograd in fp64."""
import pytest
import torch

from oracle import i2sdf_oracle as orc
from helpers import assert_close, sd_from_npz, t
from test_gpu_train_forward import make_engine, dbl

pytestmark = pytest.mark.gpu


def _cfgs(which):
    from i2sdf_amd.config import synthetic_conf, plumbing_conf
    if which == "synthetic":
        return orc.synthetic_cfg(False), synthetic_conf(False)
    if which == "light":
        return orc.synthetic_cfg(True), synthetic_conf(True)
    if which == "plumbing":
        return orc.plumbing_cfg(False), plumbing_conf(False)
    return orc.plumbing_cfg(True), plumbing_conf(True)


@pytest.mark.parametrize("which,M", [("plumbing", 300), ("plumbing_skip", 1500), ("synthetic", 2100), ("light", 640)])
def test_sdf_backward_param_grads(which, M):
    """probe loss = sum(sdf*sw) + sum(feat*fw) + sum((|grad|-1)^2): exercises s-bar, f-bar and the double backward."""
    ocfg, conf = _cfgs(which)
    sd = orc.perturb_params(orc.init_params(ocfg, seed=11), 0.05, seed=12)
    eng = make_engine(conf, sd)
    flat = eng.layout.flat_from_state_dict(sd).cuda()
    g = torch.Generator().manual_seed(5)
    x = (torch.rand(M, 3, generator=g) * 2 - 1) * 1.5
    F = ocfg.sdf.feature_size
    sw, fw_ = torch.randn(M, 1, generator=g), torch.randn(M, F, generator=g) * 0.1
    m_f = M - 37                                            # the last 37 points have no feature gradient (like eikonal points)
    fw_[m_f:] = 0
    # oracle (fp64 autograd)
    params = {k: v.double().requires_grad_(True) for k, v in sd.items() if k.startswith("implicit_network")}
    sdf, feat, grad = orc.sdf_outputs(params, ocfg.sdf, x.double(), create_graph=True)
    loss = (sdf * sw.double()).sum() + (feat * fw_.double()).sum() + ((grad.norm(2, dim=1) - 1) ** 2).sum()
    names = list(params)
    ref = dict(zip(names, torch.autograd.grad(loss, [params[k] for k in names], allow_unused=True)))
    # HIP
    fwd = eng.sdf_forward_grad(points=x.cuda())
    n = fwd["grad"]
    nn = n.norm(dim=1, keepdim=True)
    nbar = 2 * (nn - 1) * n / nn
    Mp = fwd["Mp"]
    fbar = torch.zeros(Mp, F, device="cuda")
    fbar[:M] = fw_.cuda()
    bw = eng.sdf_backward(fwd, sbar=sw.reshape(-1).cuda(), fbar=fbar, m_fbar=m_f, nbar=nbar)
    gflat = torch.zeros_like(flat)
    eng.weight_grads(flat, gflat, fwd, bw, M_main=m_f, fbar=fbar)
    got = eng.layout.state_dict_from_flat(gflat.cpu())
    for k in names:
        r = ref[k] if ref[k] is not None else torch.zeros_like(params[k])
        assert_close(got[k], r, 1e-4, k)


def test_sdf_backward_golden(golden):
    """Against the reference's own parameter gradients (fixture G3: skip net, probe loss with double backward)."""
    from i2sdf_amd.config import plumbing_conf
    z = golden("g3_sdf_skip")
    ocfg = orc.plumbing_cfg(skip=True)
    sd = orc.init_params(ocfg)
    sd.update({"implicit_network." + k: v for k, v in sd_from_npz(z, "sd.").items()})
    eng = make_engine(plumbing_conf(skip=True), sd)
    flat = eng.layout.flat_from_state_dict(sd).cuda()
    x = t(z["x"])
    M = x.shape[0]
    fwd = eng.sdf_forward_grad(points=x.cuda())
    assert_close(fwd["grad"].cpu(), z["grad"], 2e-5, "gradient")
    n = fwd["grad"]
    nn = n.norm(dim=1, keepdim=True)
    nbar = 2 * (nn - 1) * n / nn
    fbar = torch.zeros(fwd["Mp"], 64, device="cuda")
    fbar[:M] = t(z["feat_w"]).cuda()
    bw = eng.sdf_backward(fwd, sbar=torch.ones(M, device="cuda"), fbar=fbar, m_fbar=M, nbar=nbar)
    gflat = torch.zeros_like(flat)
    eng.weight_grads(flat, gflat, fwd, bw, M_main=M, fbar=fbar)
    got = eng.layout.state_dict_from_flat(gflat.cpu())
    for k in z.files:
        if k.startswith("grad."):
            assert_close(got["implicit_network." + k[5:]], z[k], 1e-4, k)


@pytest.mark.parametrize("which", ["plumbing", "synthetic", "light"])
def test_rgb_backward_param_and_feature_grads(which):
    ocfg, conf = _cfgs(which)
    sd = orc.perturb_params(orc.init_params(ocfg, seed=13), 0.05, seed=14)
    eng = make_engine(conf, sd)
    flat = eng.layout.flat_from_state_dict(sd).cuda()
    g = torch.Generator().manual_seed(6)
    B, n = 31, 9
    M, F = B * n, ocfg.rgb.feature_size
    x = (torch.rand(M, 3, generator=g) * 2 - 1)
    dirs = torch.nn.functional.normalize(torch.randn(B, 3, generator=g), dim=1)
    cw = torch.randn(M, 3, generator=g)
    params = {k: v.double().requires_grad_(True) for k, v in sd.items() if not k.startswith("light") and k != "density.beta"}
    o = orc.sdf_forward(params, ocfg.sdf, x.double())
    feat = o[:, 1:]
    rgb = orc.rgb_forward(params, ocfg.rgb, dirs.double().unsqueeze(1).repeat(1, n, 1).reshape(-1, 3), feat)
    loss = (rgb * cw.double()).sum()
    names = list(params)
    ref = dict(zip(names, torch.autograd.grad(loss, [params[k] for k in names], allow_unused=True)))
    fwd = eng.sdf_forward_grad(points=x.cuda())
    rgb_h, rs, pev = eng.rgb_forward(dirs.cuda(), n, fwd["feat"], M)
    assert_close(rgb_h.cpu(), rgb, 2e-5, "rgb")
    gar, ga_last, fbar = eng.rgb_backward(rgb_h, cw.cuda(), rs, M)
    bw = eng.sdf_backward(fwd, sbar=None, fbar=fbar, m_fbar=M, nbar=None)
    gflat = torch.zeros_like(flat)
    eng.weight_grads(flat, gflat, fwd, bw, M_main=M, fbar=fbar, rgb_fw={"pev": pev, "rs": rs}, rgb_bw={"gar": gar, "ga_last": ga_last})
    got = eng.layout.state_dict_from_flat(gflat.cpu())
    for k in names:
        r = ref[k] if ref[k] is not None else torch.zeros_like(params[k])
        assert_close(got[k], r, 1e-4, k)


@pytest.mark.parametrize("which,parts", [("synthetic", 0), ("light", 0), ("synthetic", 4)])
def test_backward_split_k_tail(which, parts):
    """256 full workgroups + a short tail: the tail of both backward kernels runs as split-K workgroups (ksplit.h) with parts = 0, as
    full workgroups of the last of four point ranges with parts = 4 (I2SDF_OPT_PARTS).
    The probe loss only weights the tail and a few bulk points, so the oracle differentiates just those."""
    ocfg, conf = _cfgs(which)
    sd = orc.perturb_params(orc.init_params(ocfg, seed=21), 0.05, seed=22)
    eng = make_engine(conf, sd, parts=parts)
    flat = eng.layout.flat_from_state_dict(sd).cuda()
    g = torch.Generator().manual_seed(23)
    n = 7
    B = (256 * 128 + 500 + n - 1) // n
    M, F = B * n, ocfg.rgb.feature_size
    x = (torch.rand(M, 3, generator=g) * 2 - 1) * 1.5
    dirs = torch.nn.functional.normalize(torch.randn(B, 3, generator=g), dim=1)
    idx = torch.cat([torch.arange(0, 150), torch.arange(256 * 128 - 90, M)])
    sel = torch.zeros(M, 1)
    sel[idx] = 1
    cw = torch.randn(M, 3, generator=g) * sel
    sw = torch.randn(M, 1, generator=g) * sel
    m_f = M - 29
    # oracle on the selected points only
    params = {k: v.double().requires_grad_(True) for k, v in sd.items() if not k.startswith("light") and k != "density.beta"}
    sdf, feat, grad = orc.sdf_outputs(params, ocfg.sdf, x.double()[idx], create_graph=True)
    with_rgb = (idx < m_f).double().unsqueeze(1)
    rgb = orc.rgb_forward(params, ocfg.rgb, dirs.double()[idx // n], feat)
    loss = (rgb * cw.double()[idx] * with_rgb).sum() + (sdf * sw.double()[idx]).sum() + ((grad.norm(2, dim=1) - 1) ** 2).sum()
    names = list(params)
    ref = dict(zip(names, torch.autograd.grad(loss, [params[k] for k in names], allow_unused=True)))
    # HIP
    fwd = eng.sdf_forward_grad(points=x.cuda())
    rgb_h, rs, pev = eng.rgb_forward(dirs.cuda(), n, fwd["feat"], M)
    cw_d = cw.clone()
    cw_d[m_f:] = 0
    gar, ga_last, fbar = eng.rgb_backward(rgb_h, cw_d.cuda(), rs, M)
    nvec = fwd["grad"]
    nn = nvec.norm(dim=1, keepdim=True)
    nbar = 2 * (nn - 1) * nvec / nn * sel.cuda()
    bw = eng.sdf_backward(fwd, sbar=sw.reshape(-1).cuda(), fbar=fbar, m_fbar=m_f, nbar=nbar)
    gflat = torch.zeros_like(flat)
    eng.weight_grads(flat, gflat, fwd, bw, M_main=m_f, fbar=fbar, rgb_fw={"pev": pev, "rs": rs}, rgb_bw={"gar": gar, "ga_last": ga_last})
    got = eng.layout.state_dict_from_flat(gflat.cpu())
    for k in names:
        r = ref[k] if ref[k] is not None else torch.zeros_like(params[k])
        assert_close(got[k], r, 1e-4, k)
