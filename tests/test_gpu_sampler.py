"""GPU: the error-bounded sampler (device loop) vs the oracle / the reference's golden z_vals.

Individual depths are ill-conditioned where the inverse CDF is flat (pdf = weights + 1e-5 with weights ~ 0): a 1-ulp
change of an SDF value moves such a sample by ~1e-3 while the rendering is unaffected (those samples carry ~zero
weight); see DESIGN.md.  So depths are compared with a robust criterion (median / 99th percentile + bounded outlier
fraction), the iteration count exactly, and the rendered outputs through the end-to-end tests."""
import numpy as np
import pytest
import torch

from oracle import i2sdf_oracle as orc
from helpers import sd_from_npz, t, make_draws
from test_gpu_train_forward import make_engine

pytestmark = pytest.mark.gpu


def frac_off(z, ref, far=6.0):
    err = (z.detach().cpu().double() - torch.as_tensor(ref).double()).abs() / far
    return (err > 1e-4).double().mean().item(), err.median().item()


def robust_z_check(z, ref, far=6.0, what="", noise=0.0):
    """`noise` = fraction of depths by which the oracle itself moves between fp32 and fp64 arithmetic (same inputs):
    the HIP path must not be further from the reference than twice that, plus 2 %."""
    z = z.detach().cpu().double()
    ref = torch.as_tensor(ref).double()
    assert z.shape == ref.shape, (z.shape, ref.shape)
    assert torch.isfinite(z).all()
    assert (z[:, 1:] >= z[:, :-1]).all(), "rows must be sorted"
    frac_bad, med = frac_off(z, ref, far)
    assert med <= 1e-6, f"{what}: median |dz|/far {med:.2e}"
    assert frac_bad <= 2 * noise + 0.02, f"{what}: {frac_bad*100:.2f}% of depths differ by more than 1e-4*far (fp32 noise {noise*100:.2f}%)"
    return frac_bad


def _eval_rays(tvec, B=1024, W=32, H=32, f=30.0):
    K = torch.eye(4); K[0, 0], K[1, 1], K[0, 2], K[1, 2] = f, f, W / 2, H / 2
    pose = torch.eye(4); pose[:3, 3] = torch.as_tensor(tvec)
    ys, xs = torch.meshgrid(torch.arange(H), torch.arange(W), indexing="ij")
    uv = torch.stack([xs, ys], -1).float().reshape(1, -1, 2)[:, :B]
    return {"uv": uv, "intrinsics": K.unsqueeze(0), "pose": pose.unsqueeze(0)}


@pytest.mark.parametrize("tag", ["in", "out"])
def test_sampler_eval_vs_reference_golden(golden, tag):
    from i2sdf_amd.config import plumbing_conf
    z = golden("g7_g8_eval")
    sd = sd_from_npz(z, "sd.")
    sd["density.beta"] = torch.tensor(float(z[f"{tag}.beta_param"]))
    eng = make_engine(plumbing_conf(), sd)
    flat = eng.layout.flat_from_state_dict(sd).cuda()
    inp = _eval_rays(z[f"{tag}.t"])
    cam, dirs, _ = eng.ray_setup(inp["uv"].cuda(), inp["pose"].cuda(), inp["intrinsics"].cuda())
    zo, zeik, iters = eng.sample_rays(flat, cam, dirs, training=False)
    assert int(iters.item()) == int(z[f"{tag}.iters"])
    robust_z_check(zo, z[f"{tag}.z_vals"], what=f"z_vals[{tag}]")


# (the last case: the BASELINE batch width -- 1024 rays, five iterations -- against the oracle's own sampler in fp64)
@pytest.mark.parametrize("which,tvec,beta,B", [("synthetic", (0.1, -0.2, 0.3), 0.1, 256), ("synthetic", (0.0, 0.0, -2.0), 0.02, 256),
                                               ("light", (0.0, 0.0, -2.0), 0.02, 200), ("synthetic", (0.0, 0.0, -2.0), 0.02, 1024)])
@pytest.mark.parametrize("planes", [2, 3])
def test_sampler_eval_full_size(which, tvec, beta, B, planes):
    """planes: split planes per operand of the sampler's sdf-only passes -- 2 = I2SDF_OPT_SAMPLER_BF16X2 (the default since round 6),
    3 = the fp32-equivalent form.  Both must reproduce the oracle's iteration count exactly and keep the same depth bars."""
    from i2sdf_amd.config import synthetic_conf
    from helpers import camera_inputs
    light = which == "light"
    ocfg = orc.synthetic_cfg(light)
    sd = orc.init_params(ocfg, seed=21)
    sd["density.beta"] = torch.tensor(beta)
    eng = make_engine(synthetic_conf(light), sd)
    eng.set_sampler_bf16x2(planes == 2)
    flat = eng.layout.flat_from_state_dict(sd).cuda()
    inp = camera_inputs(B, tvec, train_layout=False)
    cam_o, dirs_o, _ = orc.prepare_rays(inp["uv"], inp["pose"], inp["intrinsics"])
    tr = orc.SamplerTrace()
    z_ref, _ = orc.sample_z_vals(sd, ocfg, dirs_o, cam_o, training=False, trace=tr)
    z64, _ = orc.sample_z_vals({k: v.double() for k, v in sd.items()}, ocfg, dirs_o.double(), cam_o.double(), training=False)
    noise, _ = frac_off(z_ref, z64)
    cam, dirs, _ = eng.ray_setup(inp["uv"].cuda(), inp["pose"].cuda(), inp["intrinsics"].cuda())
    zo, zeik, iters = eng.sample_rays(flat, cam, dirs, training=False)
    assert int(iters.item()) == tr.iters
    bad = robust_z_check(zo, z64, what="z_vals", noise=noise)
    print(f"planes={planes}: {bad * 100:.3f}% of depths off by > 1e-4 far (the fp32 oracle itself vs fp64: {noise * 100:.3f}%)")


@pytest.mark.parametrize("force", [0, 1, 3])
def test_sampler_train_with_draws(force):
    from i2sdf_amd.config import plumbing_conf
    from helpers import camera_inputs
    ocfg = orc.plumbing_cfg(skip=True)
    sd = orc.perturb_params(orc.init_params(ocfg, seed=23), 0.05, seed=24)
    sd["density.beta"] = torch.tensor(0.03)
    eng = make_engine(plumbing_conf(skip=True), sd)
    flat = eng.layout.flat_from_state_dict(sd).cuda()
    B = 333
    inp = camera_inputs(B, (0.0, 0.2, -1.8), W=32, H=32, f=30.0, seed=3)
    cam_o, dirs_o, _ = orc.prepare_rays(inp["uv"], inp["pose"], inp["intrinsics"])
    # row length that extra_idx indexes depends on the iteration count: take it from the oracle run
    tr = orc.SamplerTrace()
    dr = make_draws(ocfg, B, n_row=32, seed=1)
    z_probe, _ = orc.sample_z_vals(sd, ocfg, dirs_o, cam_o, training=True, draws=dr, force_iters=force or None, trace=tr)
    dr = make_draws(ocfg, B, n_row=32 * tr.iters, seed=1)      # same strat_u/cdf_u (same seed order), valid extra_idx
    tr = orc.SamplerTrace()
    z_ref, zeik_ref = orc.sample_z_vals(sd, ocfg, dirs_o, cam_o, training=True, draws=dr, force_iters=force or None, trace=tr)
    cam, dirs, _ = eng.ray_setup(inp["uv"].cuda(), inp["pose"].cuda(), inp["intrinsics"].cuda())
    zo, zeik, iters = eng.sample_rays(flat, cam, dirs, training=True, strat_u=dr.strat_u.cuda(), cdf_u=dr.cdf_u.cuda(),
                                      extra_idx=dr.extra_idx.repeat(ocfg.sampler.max_total_iters, 1).cuda(), eik_idx=dr.eik_idx.cuda(), force_iters=force)
    assert int(iters.item()) == tr.iters
    d64 = orc.Draws(strat_u=dr.strat_u.double(), cdf_u=dr.cdf_u.double(), extra_idx=dr.extra_idx, eik_idx=dr.eik_idx)
    z64, _ = orc.sample_z_vals({k: v.double() for k, v in sd.items()}, ocfg, dirs_o.double(), cam_o.double(), training=True, draws=d64,
                               force_iters=force or None)
    noise, _ = frac_off(z_ref, z64)
    robust_z_check(zo, z64, what="z_vals(train)", noise=noise)
    got = torch.gather(zo.cpu(), 1, dr.eik_idx.long().unsqueeze(-1))
    assert torch.equal(got, zeik.cpu()), "z_eik must be z_vals[eik_idx]"


def test_error_bound_golden(golden):
    """S1 in isolation: ErrorBoundSampler.get_error_bound (ray_sampler.py:243-251) vs the reference's recorded values (G7b),
    scalar beta and one beta per ray, with the fixture's explicit d*."""
    from i2sdf_amd.config import plumbing_conf
    z = golden("g7b_error_bound")
    eng = make_engine(plumbing_conf(), orc.init_params(orc.plumbing_cfg(), seed=0))
    zr, sr, ds = t(z["z"]).cuda(), t(z["sdf"]).cuda(), t(z["d_star"]).cuda()
    eb = eng.error_bound(zr, sr, torch.tensor(float(z["beta_scalar"])), d_star=ds)
    np.testing.assert_allclose(eb.cpu().numpy(), z["eb_scalar"], rtol=2e-5, atol=1e-30)
    eb = eng.error_bound(zr, sr, t(z["beta_rows"]).cuda(), d_star=ds)
    np.testing.assert_allclose(eb.cpu().numpy(), z["eb_rows"], rtol=2e-5, atol=1e-30)


def test_d_star_four_cases():
    """Theorem-1 bound d* per interval (ray_sampler.py:99-114): rows built so that every branch occurs -- first_cond
    (a^2+b^2<=c^2 -> |d_i|), second_cond (-> |d_{i+1}|), the Heron height of the triangle, a sign change (-> 0), and the
    degenerate b+c<=a case (-> 0) -- vs the oracle's restatement, elementwise; then the error bound that uses it."""
    from i2sdf_amd.config import plumbing_conf
    eng = make_engine(plumbing_conf(), orc.init_params(orc.plumbing_cfg(), seed=0))
    g = torch.Generator().manual_seed(5)
    B, n = 64, 150                                         # n > 128: rows of 3 samples per lane
    zr = torch.sort(torch.rand(B, n, generator=g) * 6.0, -1)[0]
    sr = torch.randn(B, n, generator=g) * 0.5
    sr[0] = sr[0].abs() + 5.0                              # |d| >> interval: first/second cond everywhere
    sr[1, ::2] *= 0.01                                     # alternating tiny/large: second_cond then first_cond
    sr[2] = sr[2].abs() * 0.02 + 1e-3                      # |d| << interval: b + c - a <= 0 -> 0
    sr[3] = 0.3 + 0.01 * torch.rand(n, generator=g)        # nearly equal, same sign: Heron branch
    sr[4, 10] = 0.0                                        # sign(0) * sign(x) != 1 -> 0
    ref_ds = orc.d_star_bound(zr, sr)
    first = ((zr[:, 1:] - zr[:, :-1]) ** 2 + sr[:, :-1] ** 2 <= sr[:, 1:] ** 2)
    second = ((zr[:, 1:] - zr[:, :-1]) ** 2 + sr[:, 1:] ** 2 <= sr[:, :-1] ** 2)
    same = sr[:, 1:].sign() * sr[:, :-1].sign() == 1
    heron = ~first & ~second & (sr[:, :-1].abs() + sr[:, 1:].abs() - (zr[:, 1:] - zr[:, :-1]) > 0)
    for name, m in (("first", first & ~second & same), ("second", second & same), ("heron", heron & same), ("sign change", ~same),
                    ("degenerate", ~first & ~second & ~heron & same)):
        assert int(m.sum()) > 0, f"case '{name}' does not occur in the test rows"
    beta = torch.linspace(0.02, 0.4, B)
    eb, ds = eng.error_bound(zr.cuda(), sr.cuda(), beta.cuda(), want_d_star=True)
    ds = ds.cpu()
    # the Heron height is a difference of nearly equal products: compare it relative to the interval's scale
    scale = torch.maximum(torch.maximum(sr[:, :-1].abs(), sr[:, 1:].abs()), zr[:, 1:] - zr[:, :-1])
    assert float(((ds - ref_ds).abs() / scale).max()) <= 5e-5
    assert torch.equal(ds == 0, ref_ds == 0)               # branch selection is exact
    ref_eb = orc.error_bound(beta.unsqueeze(-1), sr, zr[:, 1:] - zr[:, :-1], ref_ds)
    np.testing.assert_allclose(eb.cpu().numpy(), ref_eb.numpy(), rtol=1e-3, atol=1e-30)
