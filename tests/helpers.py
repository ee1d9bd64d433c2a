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


def assert_close(a, b, tol, what="", floor=0.0):
    """max-norm relative error <= tol.  `floor`: lower bound of the denominator, for quantities with a natural O(1) scale whose
    reference can be ~0 on a particular batch (e.g. the colours of rays that all miss the geometry)."""
    a_ = torch.as_tensor(a).detach().cpu()
    b_ = torch.as_tensor(b).detach().cpu()
    assert a_.shape == b_.shape, f"{what}: shape {tuple(a_.shape)} vs {tuple(b_.shape)}"
    assert torch.isfinite(a_.double()).all(), f"{what}: non-finite values"
    err = rel_max(a_, b_) if floor <= 0 else float((a_.double() - b_.double()).abs().max() / max(float(b_.double().abs().max()), floor))
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


def perturbed_weights(sd, rel=2e-7, seed=0):
    """Another fp32 evaluation of the same network, emulated: every weight direction tensor times (1 + rel * N(0,1)).  rel is at
    fp32 rounding level, so a quantity that moves by more than that under this perturbation is ill-conditioned IN THE REFERENCE
    (e.g. a ReLU whose pre-activation is zero within rounding: its backward mask is then decided by rounding noise)."""
    g = torch.Generator().manual_seed(seed)
    out = {}
    for k, v in sd.items():
        out[k] = v * (1.0 + rel * torch.randn(v.shape, generator=g, dtype=v.dtype)) if k.endswith("weight_v") else v.clone()
    return out


def measured_spread(run, sd, n_perturbed=2, rel=2e-7):
    """run(sd, dtype) -> dict name -> tensor (outputs / losses / gradients of the ORACLE).  Returns name -> the largest max-norm
    relative deviation from the plain fp32 run among: the fp64 run and `n_perturbed` fp32 runs with rounding-level weight noise.
    This is how far two correct evaluations of the reference algorithm are apart on this input: the conditioning of each quantity."""
    base = run(sd, torch.float32)
    members = [run({k: v.double() for k, v in sd.items()}, torch.float64)]
    members += [run(perturbed_weights(sd, rel=rel, seed=100 + i), torch.float32) for i in range(n_perturbed)]
    spread = {}
    for k, b in base.items():
        b64 = b.detach().double()
        den = float(b64.abs().max())
        if den == 0.0:
            spread[k] = 0.0
            continue
        spread[k] = max(float((m[k].detach().double() - b64).abs().max()) / den for m in members)
    return spread


def network_of(name):
    return name.split(".")[0]


def relu_flip_analysis(grads_fn, tau=1e-6, max_candidates=64):
    """ReLU units of the radiance net whose pre-activation is zero within fp32 rounding, and what flipping each one's BACKWARD
    MASK does to the gradient.

    grads_fn() -> {name: grad} runs the fp32 oracle's training step (bitwise the reference's arithmetic, see
    test_oracle_vs_reference.py).  The derivative of relu at a pre-activation of +-1e-7 is 1 or 0 depending on the sign rounding
    noise left behind, so two correct fp32 evaluations (the reference's and any other) can legitimately differ by exactly such
    flips; each flip changes the gradient by a fixed, computable vector.  Returns (base, candidates, deltas): candidates = list of
    (layer, point, unit, pre_activation); deltas[i] = {name: grad with candidate i's mask flipped  -  base grad}.
    relu_flip_analysis.pre holds the oracle's pre-activations {layer: (points, units)} of the last call (hip_mask_flips below)."""
    rec = {}

    def record(a, l):
        rec[l] = a.detach()
        return torch.relu(a)

    with orc.relu_hook(record):
        base = grads_fn()
    relu_flip_analysis.pre = dict(rec)
    cands = []
    for l, a in rec.items():
        idx = (a.abs() < tau).nonzero()
        for m, u in idx.tolist():
            cands.append((l, m, u, float(a[m, u])))
    cands.sort(key=lambda c: abs(c[3]))
    cands = cands[:max_candidates]
    deltas = []
    for (l, m, u, _) in cands:
        def forced(a, layer, l=l, m=m, u=u):
            mask = (a.detach() > 0)
            if layer == l:
                mask = mask.clone()
                mask[m, u] = ~mask[m, u]
            return a * mask.to(a.dtype)

        with orc.relu_hook(forced):
            g = grads_fn()
        deltas.append({k: g[k] - base[k] for k in base})
    return base, cands, deltas


def hip_mask_flips(rs_point_major, pre, tau=1e-6):
    """Which backward masks does the HIP path REALLY have differently from the fp32 oracle run?  rs_point_major: the library's saved
    post-ReLU activations of the radiance net, (layers, points, units) in point-major order (relu(a) > 0 is the mask its backward
    uses); pre: the oracle's pre-activations {layer: (points, units)}.  Returns (flips = set of (layer, point, unit) with different
    masks, worst = the largest |oracle pre-activation| among them): a mask may only differ where rounding decides it."""
    flips, worst = set(), 0.0
    for l, a in pre.items():
        a = a.detach().cpu()
        hip = rs_point_major[l][: a.shape[0], : a.shape[1]].cpu() > 0
        diff = (hip != (a > 0)).nonzero()
        for m, u in diff.tolist():
            flips.add((l, m, u))
            worst = max(worst, abs(float(a[m, u])))
    return flips, worst


def explain_by_relu_flips(err, deltas, scale, max_flips=4):
    """err, deltas[i]: {name: 1-D tensor} (same sampling), scale: {name: max |reference grad|}.  Greedy search for the set of
    candidate flips (each applied whole: a backward mask is 0 or 1) that explains err: repeatedly apply the flip that lowers the
    scaled max-norm of the residual the most, while it lowers it by at least a third.  Deterministic (no least squares: most
    candidate columns are nearly zero).  Returns (chosen indices, residual {name: max-norm relative error})."""
    names = [n for n in err if scale[n] > 0]
    e = torch.cat([err[n].double() / scale[n] for n in names])
    cols = [torch.cat([d[n].double() / scale[n] for n in names]) for d in deltas]
    chosen = []
    for _ in range(max_flips):
        cur = float(e.abs().max())
        best, best_val = None, cur
        for i, c in enumerate(cols):
            if i in chosen or float(c.abs().max()) < 1e-7:
                continue
            v = float((e - c).abs().max())
            if v < best_val:
                best, best_val = i, v
        if best is None or best_val > cur * (2.0 / 3.0):
            break
        chosen.append(best)
        e = e - cols[best]
    res, off = {}, 0
    for n in names:
        k = err[n].numel()
        res[n] = float(e[off:off + k].abs().max())
        off += k
    return chosen, res


_MEMO = {}


def memo(key, fn):
    """Results of the ORACLE that do not depend on the weight-gradient mode under test (tests/conftest.py runs the gradient modules once
    per mode): computed by the first parametrisation, reused by the second.  Only oracle-side values go in here -- never anything the
    library produced."""
    if key not in _MEMO:
        _MEMO[key] = fn()
    return _MEMO[key]
