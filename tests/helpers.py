"""Shared test helpers (CPU + GPU tests)."""
import numpy as np
import torch

from oracle import i2sdf_oracle as orc


def t(a, dtype=None):
    x = torch.from_numpy(np.asarray(a))
    return x.to(dtype) if dtype is not None else x


def sd_from_npz(z, prefix="sd.", strip="", dtype=None):
    out = {}
    for k in z.files:
        if k.startswith(prefix):
            name = k[len(prefix):]
            if strip and name.startswith(strip):
                name = name[len(strip):]
            out[name] = t(z[k], dtype)
    return out


def rel_max(a, b):
    """max-norm relative error |a-b|_inf / max(|b|_inf, tiny) -- the SURVEY 8(d) criterion."""
    a, b = torch.as_tensor(a, dtype=torch.float64), torch.as_tensor(b, dtype=torch.float64)
    return float((a - b).abs().max() / b.abs().max().clamp_min(1e-30))


def assert_close(a, b, tol, what=""):
    a_ = torch.as_tensor(a).detach().cpu()
    b_ = torch.as_tensor(b).detach().cpu()
    assert a_.shape == b_.shape, f"{what}: shape {tuple(a_.shape)} vs {tuple(b_.shape)}"
    assert torch.isfinite(a_.double()).all(), f"{what}: non-finite values"
    err = rel_max(a_, b_)
    assert err <= tol, f"{what}: max-norm relative error {err:.3e} > {tol:.1e}"
    return err


def camera_inputs(B, t_xyz, W=640, H=480, f=600.0, seed=0, train_layout=True, dtype=torch.float32):
    """BASELINE.md section 3 synthetic cameras: R=I, fx=fy=f, c=(W/2,H/2), integer pixels, seeded."""
    g = torch.Generator().manual_seed(seed)
    K = torch.eye(4, dtype=dtype)
    K[0, 0], K[1, 1], K[0, 2], K[1, 2] = f, f, W / 2, H / 2
    pose = torch.eye(4, dtype=dtype)
    pose[:3, 3] = torch.tensor(t_xyz, dtype=dtype)
    if train_layout:
        uv = torch.stack([torch.randint(0, W, (B,), generator=g), torch.randint(0, H, (B,), generator=g)], -1).to(dtype).reshape(B, 1, 2)
        return {"uv": uv, "intrinsics": K.repeat(B, 1, 1), "pose": pose.repeat(B, 1, 1)}
    idx = torch.randperm(W * H, generator=g)[:B]
    uv = torch.stack([idx % W, idx // W], -1).to(dtype).reshape(1, B, 2)
    return {"uv": uv, "intrinsics": K.unsqueeze(0), "pose": pose.unsqueeze(0)}


def make_draws(cfg: orc.NetCfg, B, n_row, seed=0, dtype=torch.float32):
    """A full set of training draws (see oracle.Draws). n_row = row length extra_idx indexes into."""
    g = torch.Generator().manual_seed(seed)
    sc = cfg.sampler
    R = cfg.scene_bounding_sphere
    return orc.Draws(
        strat_u=torch.rand(B, sc.N_samples_eval, generator=g, dtype=dtype),
        cdf_u=torch.rand(B, sc.N_samples, generator=g, dtype=dtype),
        extra_idx=torch.randperm(n_row, generator=g)[: sc.N_samples_extra],
        eik_idx=torch.randint(sc.N_samples + sc.N_samples_extra + 2, (B,), generator=g),
        eik_pts=(torch.rand(B, 3, generator=g, dtype=dtype) * 2 - 1) * R,
        nbr_off=(torch.rand(B, 3, generator=g, dtype=dtype) * 2 - 1) * 0.005,
    )


def make_gt(B, seed=0, dtype=torch.float32, light=False):
    g = torch.Generator().manual_seed(seed + 100)
    gt = {"rgb": torch.rand(B, 3, generator=g, dtype=dtype), "depth": torch.rand(B, generator=g, dtype=dtype) * 3,
          "depth_mask": torch.ones(B, dtype=torch.bool),
          "normal": torch.nn.functional.normalize(torch.randn(B, 3, generator=g, dtype=dtype), dim=1),
          "normal_mask": torch.ones(B, dtype=torch.bool)}
    if light:
        gt["light_mask"] = (torch.rand(B, 1, generator=g) > 0.5).to(dtype)
    return gt


def full_width_state_dict(z, light=False):
    """The weights of the full-width fixtures (G14/G15) are not stored: rebuild them exactly as gen_golden.py did
    (oracle.init_params + perturb_params, deterministic CPU generator) and check the stored per-tensor checksums."""
    cfg = orc.synthetic_cfg(light)
    sd = orc.perturb_params(orc.init_params(cfg, seed=int(z["init_seed"])), float(z["perturb_scale"]), seed=int(z["perturb_seed"]))
    chk = torch.stack([v.double().sum() for v in sd.values()])
    assert torch.allclose(chk[:-1], t(z["sd_checksum"])[:-1], rtol=0, atol=1e-9), "rebuilt weights differ from the fixture's"
    return cfg, sd


def assert_grad_digest(z, grads, tol, stride_key="grad_stride"):
    """grads: {name: tensor}.  Compares the fixture's gradient digest (whole small tensors, strided sample of large ones) with
    the max-norm relative criterion, the denominator being the FULL reference tensor's max |g| (stored)."""
    stride = int(z[stride_key])
    worst = 0.0
    for k in z.files:
        if not k.startswith("gsample."):
            continue
        n = k[len("gsample."):]
        g = torch.as_tensor(grads[n]).detach().cpu().double().reshape(-1)
        ref = t(z[k]).double().reshape(-1)
        mine = g if g.numel() <= 1024 else g[::stride]
        assert mine.shape == ref.shape, n
        gmax = float(z["gmax." + n])
        if gmax == 0.0:
            assert float(g.abs().max()) == 0.0, f"grad {n}: reference is exactly zero"
            continue
        err = float((mine - ref).abs().max()) / gmax
        assert err <= tol, f"grad {n}: max-norm relative error {err:.3e} > {tol:.1e}"
        worst = max(worst, err)
        # the sum over the whole tensor guards the elements the sample skips (loose: cancellation)
        assert abs(float(g.sum()) - float(z["gsum." + n])) <= 50 * tol * gmax * max(1.0, g.numel() ** 0.5), f"grad {n}: sum"
    return worst
