"""CPU: configuration parsing, parameter layout / initialisation, the module's state_dict contract, the loss module."""
import os

import numpy as np
import pytest
import torch

from oracle import i2sdf_oracle as orc
from helpers import assert_close, sd_from_npz, t


def test_config_matches_reference_yaml_when_available():
    yaml = pytest.importorskip("yaml")
    path = "/root/reference/config"
    if not os.path.isdir(path):
        pytest.skip("reference not present")
    from i2sdf_amd.config import NetConfig, synthetic_conf
    for name, light in (("synthetic.yml", False), ("synthetic_light_mask.yml", True)):
        conf = yaml.safe_load(open(os.path.join(path, name)))["model"]
        a, b = NetConfig.from_conf(conf), NetConfig.from_conf(synthetic_conf(light))
        assert a == b


def test_layer_shapes_match_survey_appendix_b():
    from i2sdf_amd.config import NetConfig, synthetic_conf
    c = NetConfig.from_conf(synthetic_conf())
    assert [d for d in c.sdf.dims] == [(256, 39), (256, 256), (256, 256), (217, 256), (256, 256), (256, 256), (256, 256), (256, 256), (257, 256)]
    assert c.sdf.skip_layer == 4 and c.sdf.pe_dim == 39
    assert [d for d in c.rgb.dims] == [(256, 283), (256, 256), (256, 256), (256, 256), (3, 256)]
    c3 = NetConfig.from_conf(synthetic_conf(True))
    assert c3.sdf.skip_layer == 3 and len(c3.sdf.dims) == 7 and c3.sdf.dims[2] == (217, 256)
    assert c3.light.dims == [(128, 256), (1, 128)]
    assert (c.sdf.dims == orc.synthetic_cfg().sdf.layer_shapes()) and (c.rgb.dims == orc.synthetic_cfg().rgb.layer_shapes())


def test_unsupported_configs_raise():
    from i2sdf_amd.config import NetConfig, synthetic_conf
    c = synthetic_conf(); c["bg_network"] = {}
    with pytest.raises(NotImplementedError):
        NetConfig.from_conf(c)
    c = synthetic_conf(); c["rendering_network"]["mode"] = "idr"
    with pytest.raises(NotImplementedError):
        NetConfig.from_conf(c)
    c = synthetic_conf(); c["implicit_network"]["embed_type"] = "fourier"
    with pytest.raises(NotImplementedError):
        NetConfig.from_conf(c)


def test_param_layout_roundtrip_and_order(golden):
    from i2sdf_amd.config import NetConfig, plumbing_conf
    from i2sdf_amd.params import ParamLayout
    z = golden("g9_train_light")
    lay = ParamLayout(NetConfig.from_conf(plumbing_conf(skip=True, light=True)))
    ref_keys = [k[3:] for k in z.files if k.startswith("sd.")]
    assert [n for n, _, _ in lay.entries] == ref_keys           # the reference's state_dict order
    sd = sd_from_npz(z, "sd.")
    flat = lay.flat_from_state_dict(sd)
    back = lay.state_dict_from_flat(flat)
    for k in sd:
        assert torch.equal(back[k].reshape(sd[k].shape), sd[k])
    offs = [o for _, o, _ in lay.entries]
    assert offs == sorted(offs) and lay.n_params == flat.numel()


def test_reference_init_scheme():
    from i2sdf_amd import I2SDFNetwork, synthetic_conf
    torch.manual_seed(0)
    net = I2SDFNetwork(synthetic_conf())
    sd = net.state_dict()
    assert torch.count_nonzero(sd["implicit_network.lin0.weight_v"][:, 3:]) == 0            # mlp.py:61
    assert torch.count_nonzero(sd["implicit_network.lin4.weight_v"][:, -36:]) == 0           # mlp.py:66
    assert torch.all(sd["implicit_network.lin8.bias"] == -0.6)                               # mlp.py:58, bias 0.6
    assert abs(sd["implicit_network.lin8.weight_v"].mean().item() - np.sqrt(np.pi) / 16) < 1e-4
    assert_close(sd["implicit_network.lin2.weight_g"], sd["implicit_network.lin2.weight_v"].norm(dim=1, keepdim=True), 1e-6, "g = ||v||")
    b = 1 / np.sqrt(283)
    assert sd["rendering_network.lin0.weight_v"].abs().max() <= b + 1e-6
    assert float(sd["density.beta"]) == pytest.approx(0.1)
    # the oracle's own initialiser draws from the same distributions
    o = orc.init_params(orc.synthetic_cfg())
    assert set(o) == set(sd) and all(tuple(o[k].shape) == tuple(sd[k].shape) for k in sd)


def test_module_surface():
    from i2sdf_amd import I2SDFNetwork, synthetic_conf
    net = I2SDFNetwork(synthetic_conf(True))
    assert net.rendering_network.mode == "nerf" and net.use_light and not net.use_bg
    groups = net.get_param_groups(5e-4)
    assert len(groups) == 1 and groups[0]["lr"] == 5e-4
    assert sum(p.numel() for p in groups[0]["params"]) == 635965
    assert float(net.density.get_beta()) == pytest.approx(0.1 + 1e-4)
    sdf = torch.linspace(-1, 1, 11)
    assert_close(net.density(sdf), orc.laplace_density(sdf, net.density.get_beta().detach()), 1e-7, "LaplaceDensity.forward")


def test_loss_module_has_no_cpu_path():
    """The product loss is the fused HIP entry point only: CPU tensors raise (the CPU restatement is oracle.i2sdf_loss, pinned by
    G11 in test_oracle_golden.py); the constructor keeps the reference's signature and its smooth_iter rule (:300-302)."""
    from i2sdf_amd import I2SDFLoss
    lf = I2SDFLoss(eikonal_weight=0.1, smooth_weight=0.01, smooth_iter=150000, depth_weight=0.1, normal_weight=0.05, bubble_weight=0.5,
                   min_bubble_iter=50000, max_bubble_iter=200000)                               # config/synthetic.yml:15-23 shape
    assert lf.smooth_iter == 200000 and lf.angular_weight == 0.05
    out = {"rgb_values": torch.rand(4, 3), "depth_values": torch.rand(4), "weight_sum": torch.rand(4, 1)}
    with pytest.raises(RuntimeError, match="no eager-torch or CPU fallback"):
        lf(out, {"rgb": torch.rand(4, 3)}, 0)


def test_unsupported_shapes_are_refused_at_construction_with_the_reason():
    """Widths / sampler capacities the kernels are not instantiated for raise when the config is read, not at the first launch."""
    import copy
    import pytest
    from i2sdf_amd.config import NetConfig, synthetic_conf
    for mutate, needle in (
            (lambda c: c["implicit_network"].update(dims=[128] * 8), "hidden width 128"),
            (lambda c: c["implicit_network"].update(dims=[512] * 8), "hidden width 512"),
            (lambda c: c["rendering_network"].update(dims=[128] * 4), "radiance width 128"),
            (lambda c: c["ray_sampler"].update(N_samples_eval=256), "N_samples_eval=256"),
            (lambda c: c["ray_sampler"].update(max_total_iters=8), "max_total_iters=8"),
            (lambda c: c["implicit_network"].update(multires=10), "multires")):
        conf = copy.deepcopy(synthetic_conf())
        mutate(conf)
        with pytest.raises(NotImplementedError) as e:
            NetConfig.from_conf(conf)
        assert needle in str(e.value), (needle, str(e.value))
    NetConfig.from_conf(synthetic_conf())          # the shipped shapes pass
    NetConfig.from_conf(synthetic_conf(True))


def test_knockout_patches_match_the_current_sources():
    """scripts/ab/knockout_build.py patches a COPY of csrc by exact-string replacement (the shipped headers carry no knock-out code): every
    pattern must match the current sources exactly once, or the tool would time an unpatched kernel."""
    import importlib.util
    import os
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    spec = importlib.util.spec_from_file_location("knockout_build", os.path.join(root, "scripts", "ab", "knockout_build.py"))
    m = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(m)
    for name, patches in m.PATCHES.items():
        for fn, old, new in patches:
            assert open(os.path.join(m.CSRC, fn)).read().count(old) == 1, (name, fn)
    assert "I2SDF_ABL" not in open(os.path.join(m.CSRC, "common.h")).read()
