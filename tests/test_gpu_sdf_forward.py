"""GPU: weight packing + fused SDF-MLP forward (HIP, through the C ABI) vs the CPU oracle."""
import numpy as np
import pytest
import torch

from oracle import i2sdf_oracle as orc
from helpers import assert_close, sd_from_npz, t

pytestmark = pytest.mark.gpu


def _engine(conf, sd):
    from i2sdf_amd.config import NetConfig
    from i2sdf_amd.engine import RenderEngine
    cfg = NetConfig.from_conf(conf)
    eng = RenderEngine(cfg)
    flat = eng.layout.flat_from_state_dict(sd).cuda()
    eng.pack(flat)
    return eng


@pytest.mark.parametrize("light", [False, True])
@pytest.mark.parametrize("M", [1, 100, 4096 + 17])
def test_sdf_forward_full_size(light, M):
    from i2sdf_amd.config import synthetic_conf
    ocfg = orc.synthetic_cfg(light)
    sd = orc.perturb_params(orc.init_params(ocfg, seed=3), 0.05, seed=4)
    eng = _engine(synthetic_conf(light), sd)
    g = torch.Generator().manual_seed(M)
    x = (torch.rand(M, 3, generator=g) * 2 - 1) * 2.5
    ref = orc.sdf_forward({k: v.double() for k, v in sd.items()}, ocfg.sdf, x.double())   # fp64 arbiter
    sdf = eng.sdf_forward(x.cuda())
    assert_close(sdf.cpu(), ref[:, :1], 1e-5, "sdf")
    sdf2, feat = eng.sdf_forward(x.cuda(), want_features=True)
    assert_close(sdf2.cpu(), ref[:, :1], 1e-5, "sdf (full)")
    assert_close(feat.cpu(), ref[:, 1:], 1e-5, "feature")


@pytest.mark.parametrize("name,skip", [("g2_sdf", False), ("g3_sdf_skip", True)])
def test_sdf_forward_golden(golden, name, skip):
    """Directly against the reference's own outputs (committed fixture)."""
    from i2sdf_amd.config import plumbing_conf
    z = golden(name)
    ocfg = orc.plumbing_cfg(skip=skip)
    sd = orc.init_params(ocfg)
    sd.update({"implicit_network." + k: v for k, v in sd_from_npz(z, "sd.").items()})
    eng = _engine(plumbing_conf(skip=skip), sd)
    sdf, feat = eng.sdf_forward(t(z["x"]).cuda(), want_features=True)
    out = torch.cat([sdf, feat], 1).cpu()
    assert_close(out, z["out"], 2e-5, "ImplicitNetwork.forward")


@pytest.mark.parametrize("which", ["synthetic", "light", "plumbing", "plumbing_skip"])
def test_sdf_forward_bf16x3(which):
    """The bf16x3 split-arithmetic forward (256-wide nets: 16-point waves, x3h.h; 64-wide: x3.h) against the fp64 oracle at the SAME bar as the fp32 MFMA kernel, and
    against the fp32 kernel itself."""
    from i2sdf_amd.config import synthetic_conf, plumbing_conf
    if which in ("synthetic", "light"):
        ocfg, conf = orc.synthetic_cfg(which == "light"), synthetic_conf(which == "light")
    else:
        ocfg, conf = orc.plumbing_cfg(skip=which.endswith("skip")), plumbing_conf(skip=which.endswith("skip"))
    sd = orc.perturb_params(orc.init_params(ocfg, seed=3), 0.05, seed=4)
    eng = _engine(conf, sd)
    M = 256 * 128 + 333
    g = torch.Generator().manual_seed(17)
    x = (torch.rand(M, 3, generator=g) * 2 - 1) * 2.5
    eng.set_sdf_forward_bf16x3(False)
    f32 = eng.sdf_forward(x.cuda()).cpu()
    eng.set_sdf_forward_bf16x3(True)
    x3 = eng.sdf_forward(x.cuda()).cpu()
    idx = torch.cat([torch.arange(0, 3000), torch.arange(M - 500, M)])
    ref = orc.sdf_forward({k: v.double() for k, v in sd.items()}, ocfg.sdf, x.double()[idx])[:, :1]
    e32 = assert_close(f32[idx], ref, 1e-5, "sdf (fp32 MFMA)")
    e3 = assert_close(x3[idx], ref, 1e-5, "sdf (bf16x3)")
    print(f"max-norm relative error vs fp64: fp32 MFMA {e32:.2e}, bf16x3 {e3:.2e}")
    assert_close(x3, f32, 1e-5, "bf16x3 vs fp32 kernel, all points")


@pytest.mark.parametrize("light", [False, True])
def test_implicit_network_call_returns_257_columns(light):
    """`net.implicit_network(x)` -- what the unchanged meshing callers use (model/eval/recon.py:51,90; utils/plots.py:52) -- returns
    [sdf | feature] rows; since round 6 on the 16-point-wave bf16x3 kernel of the training forward (mlp_fwd.hip: launch_sdf_fwd).  All 257
    columns against the fp64 oracle at 1e-5, at a size with several workgroups and a ragged tail, both arithmetic forms, and the sdf
    column against the sdf-only kernel (the sampler's / i2sdf_sdf_grid's)."""
    from i2sdf_amd import I2SDFNetwork
    from i2sdf_amd.config import synthetic_conf
    ocfg = orc.synthetic_cfg(light)
    sd = orc.perturb_params(orc.init_params(ocfg, seed=3), 0.05, seed=4)
    net = I2SDFNetwork(synthetic_conf(light))
    net.load_state_dict(sd)
    net = net.cuda().eval()
    M = 5 * 128 + 77
    g = torch.Generator().manual_seed(7)
    x = (torch.rand(M, 3, generator=g) * 2 - 1) * 2.5
    ref = orc.sdf_forward({k: v.double() for k, v in sd.items()}, ocfg.sdf, x.double())
    eng = net._engine_for(torch.device("cuda:0"))
    outs = {}
    for x3 in (True, False):
        eng.set_sdf_forward_bf16x3(x3)
        out = net.implicit_network(x.cuda())
        assert out.shape == (M, 257) and out.is_contiguous()
        assert_close(out.cpu(), ref, 1e-5, f"implicit_network(x), bf16x3={x3}")
        outs[x3] = out
    eng.set_sdf_forward_bf16x3(True)
    assert_close(outs[True][:, :1].cpu(), net.implicit_network.get_sdf_vals(x.cuda()).cpu(), 1e-5, "sdf column vs the sdf-only kernel")
    assert_close(outs[True].cpu(), outs[False].cpu(), 1e-5, "bf16x3 vs fp32 MFMA")
