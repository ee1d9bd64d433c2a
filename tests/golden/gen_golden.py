#!/usr/bin/env python3
"""Generate the committed golden vectors (tests/golden/*.npz) by RUNNING THE REFERENCE in this container.

    python tests/golden/gen_golden.py          # needs /root/reference; writes tests/golden/g*.npz

The reference (pure Python/PyTorch) is imported through tests/golden/ref_import.py with stub modules for its
off-path dependencies.  Only data (inputs, parameters of tiny networks, expected outputs) is written; no
reference source travels.  Vector inventory follows SURVEY.md section 8(c) G1..G11.
"""
import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import ref_import  # noqa: E402

torch.set_num_threads(4)


def npify(d):
    return {k: (v.detach().cpu().numpy() if torch.is_tensor(v) else np.asarray(v)) for k, v in d.items()}


def save(name, **arrs):
    path = os.path.join(HERE, name + ".npz")
    np.savez_compressed(path, **npify(arrs))
    print(f"wrote {path}  {os.path.getsize(path)/1024:.1f} KB")


def sd_arrays(module, prefix="sd."):
    return {prefix + k: v for k, v in module.state_dict().items()}


def perturb_(module, scale=0.05, seed=1):
    g = torch.Generator().manual_seed(seed)
    with torch.no_grad():
        for n, p in module.named_parameters():
            if n.endswith("density.beta") or n == "beta":
                continue
            ref = p.abs().mean().clamp_min(1e-3)
            p.add_(torch.randn(p.shape, generator=g) * scale * ref)


def small_conf(skip=False, light=False):
    cfg = ref_import.load_cfg("synthetic.yml").model
    cfg.feature_vector_size = 64
    cfg.implicit_network.dims = [64, 64, 64] if skip else [64, 64]
    cfg.implicit_network.skip_in = [2] if skip else []
    cfg.rendering_network.dims = [64, 64]
    cfg.ray_sampler.N_samples = 16
    cfg.ray_sampler.N_samples_eval = 32
    cfg.ray_sampler.N_samples_extra = 8
    if light:
        cfg.light_network = type(cfg)({"dims": [32], "weight_norm": True})
    cfg.use_normal = True
    return cfg


def camera_batch(B, t, W=32, H=32, f=30.0, seed=0, train_layout=True, skew=0.0):
    g = torch.Generator().manual_seed(seed)
    K = torch.eye(4)
    K[0, 0], K[1, 1], K[0, 2], K[1, 2], K[0, 1] = f, f, W / 2, H / 2, skew
    pose = torch.eye(4)
    pose[:3, 3] = torch.tensor(t)
    if train_layout:
        uv = torch.stack([torch.randint(0, W, (B,), generator=g), torch.randint(0, H, (B,), generator=g)], -1).float().reshape(B, 1, 2)
        return {"uv": uv, "intrinsics": K.repeat(B, 1, 1), "pose": pose.repeat(B, 1, 1)}
    ys, xs = torch.meshgrid(torch.arange(H), torch.arange(W), indexing="ij")
    uv = torch.stack([xs, ys], -1).float().reshape(1, -1, 2)[:, :B]
    return {"uv": uv, "intrinsics": K.unsqueeze(0), "pose": pose.unsqueeze(0)}


class DrawRecorder:
    """Record every random draw the reference makes during one training forward."""

    def __init__(self):
        self.log = []

    def __enter__(self):
        self._rand, self._randperm, self._randint = torch.rand, torch.randperm, torch.randint
        self._uniform, self._nprandint = torch.Tensor.uniform_, np.random.randint
        rec = self

        def rand(*a, **k):
            out = rec._rand(*a, **k); rec.log.append(("rand", out.clone())); return out

        def randperm(*a, **k):
            out = rec._randperm(*a, **k); rec.log.append(("randperm", out.clone())); return out

        def randint(*a, **k):
            out = rec._randint(*a, **k); rec.log.append(("randint", out.clone())); return out

        def uniform_(self_t, *a, **k):
            out = rec._uniform(self_t, *a, **k); rec.log.append(("uniform", out.clone())); return out

        def nprandint(*a, **k):
            out = rec._nprandint(*a, **k); rec.log.append(("np.randint", torch.tensor(out))); return out

        torch.rand, torch.randperm, torch.randint = rand, randperm, randint
        torch.Tensor.uniform_ = uniform_
        np.random.randint = nprandint
        return self

    def __exit__(self, *exc):
        torch.rand, torch.randperm, torch.randint = self._rand, self._randperm, self._randint
        torch.Tensor.uniform_ = self._uniform
        np.random.randint = self._nprandint


def record_z(full):
    """Wrap the model's sampler so that the (z_vals, z_samples_eik) it returns are recorded (model/network/__init__.py:95)."""
    rec = []
    orig = full.ray_sampler.get_z_vals

    def wrapped(*a, **k):
        zv, ze = orig(*a, **k)
        rec.append((zv.detach().clone(), ze.detach().clone()))
        return zv, ze

    full.ray_sampler.get_z_vals = wrapped
    return rec


def grad_digest(named_grads, stride=61):
    """Full-width gradients are 3.2 MB: keep every tensor of <= 1024 elements whole and a strided sample of the larger ones,
    plus each tensor's max |g| (the denominator of the max-norm relative error) and its sum."""
    out = {}
    for n, g_ in named_grads:
        f = g_.detach().reshape(-1)
        out["gsample." + n] = f if f.numel() <= 1024 else f[::stride]
        out["gmax." + n] = f.abs().max()
        out["gsum." + n] = f.double().sum()
    return out


def full_width(ref_model, I2SDFLoss):
    """G14/G15: the shipped synthetic.yml / synthetic_light_mask.yml networks end to end.  The 800 955 weights are not stored:
    both sides build them with oracle.init_params(seed) + perturb_params (deterministic CPU torch generator) and the reference
    model loads that state_dict, so the fixture holds inputs, the reference's own depths/draws, outputs, loss and a gradient digest."""
    sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))
    from oracle import i2sdf_oracle as orc
    gg = torch.Generator().manual_seed(1414)
    for name, light, yml in (("g14_train_full", False, "synthetic.yml"), ("g14_train_full_light", True, "synthetic_light_mask.yml")):
        ocfg = orc.synthetic_cfg(light)
        sd = orc.perturb_params(orc.init_params(ocfg, seed=141), 0.03, seed=142)
        sd["density.beta"] = torch.tensor(0.05)
        cfg = ref_import.load_cfg(yml).model
        cfg.use_normal = True
        full = ref_model.I2SDFNetwork(cfg)
        full.load_state_dict(sd)
        full.train()
        B = 32
        inp = camera_batch(B, (0.0, 0.0, -2.0), W=640, H=480, f=600.0, seed=14)
        gt = {"rgb": torch.rand(B, 3, generator=gg), "depth": torch.rand(B, generator=gg) * 3, "depth_mask": torch.rand(B, generator=gg) > 0.2,
              "normal": torch.nn.functional.normalize(torch.randn(B, 3, generator=gg), dim=1), "normal_mask": torch.rand(B, generator=gg) > 0.1}
        lkw = dict(eikonal_weight=0.1, smooth_weight=0.01, smooth_iter=None, depth_weight=0.1, normal_weight=0.05)
        if light:
            gt["light_mask"] = (torch.rand(B, 1, generator=gg) > 0.5).float()
            lkw["light_mask_weight"] = 0.5
        inp["pointcloud"] = torch.rand(24, 3, generator=gg) * 2 - 1
        lkw["bubble_weight"] = 0.5
        torch.manual_seed(1)
        np.random.seed(0)
        zrec = record_z(full)
        with DrawRecorder() as rec:
            out = full(inp)
        losses = I2SDFLoss(**lkw)(out, gt, 10)
        full.zero_grad()
        losses["loss"].backward()
        kinds = [k for k, _ in rec.log]
        assert kinds == ["rand", "rand", "randperm", "randint", "uniform", "uniform", "np.randint"], kinds
        arrs = {"in." + k: v for k, v in inp.items()}
        arrs.update({"gt." + k: v for k, v in gt.items()})
        arrs.update({"out." + k: v for k, v in out.items()})
        arrs.update({"loss." + k: v for k, v in losses.items()})
        arrs.update(grad_digest([(n, p.grad if p.grad is not None else torch.zeros_like(p)) for n, p in full.named_parameters()]))
        arrs.update({"draw.eik_pts": rec.log[4][1], "draw.nbr_off": rec.log[5][1]})
        arrs["ref.z_vals"], arrs["ref.z_eik"] = zrec[0]
        arrs["loss_kwargs"] = np.array(sorted(lkw.items(), key=lambda kv: kv[0]), dtype=object).astype(str)
        arrs["init_seed"], arrs["perturb_seed"], arrs["perturb_scale"], arrs["grad_stride"] = np.int64(141), np.int64(142), np.float32(0.03), np.int64(61)
        arrs["sd_checksum"] = torch.stack([v.double().sum() for v in sd.values()])      # guards the "same weights on both sides" premise
        save(name, **arrs)
    # G15: eval render (k data dependent), full width, with the reference's depths
    ocfg = orc.synthetic_cfg(False)
    sd = orc.perturb_params(orc.init_params(ocfg, seed=151), 0.03, seed=152)
    sd["density.beta"] = torch.tensor(0.02)
    cfg = ref_import.load_cfg("synthetic.yml").model
    cfg.use_normal = True
    full = ref_model.I2SDFNetwork(cfg)
    full.load_state_dict(sd)
    full.eval()
    P = 64
    idx = torch.randperm(640 * 480, generator=gg)[:P]
    K = torch.eye(4); K[0, 0] = K[1, 1] = 600.0; K[0, 2], K[1, 2] = 320.0, 240.0
    pose = torch.eye(4); pose[:3, 3] = torch.tensor([0.0, 0.0, -2.0])
    inp = {"uv": torch.stack([idx % 640, idx // 640], -1).float().reshape(1, P, 2), "intrinsics": K.unsqueeze(0), "pose": pose.unsqueeze(0)}
    zrec = record_z(full)
    calls = []
    orig = full.implicit_network.get_sdf_vals
    full.implicit_network.get_sdf_vals = lambda pts, _o=orig: (calls.append(pts.shape[0]), _o(pts))[1]
    out = full(inp)
    arrs = {"in." + k: v for k, v in inp.items()}
    arrs.update({"out." + k: v for k, v in out.items()})
    arrs["ref.z_vals"], arrs["ref.z_eik"] = zrec[0]
    arrs["iters"] = np.int64(len(calls))
    arrs["init_seed"], arrs["perturb_seed"], arrs["perturb_scale"] = np.int64(151), np.int64(152), np.float32(0.03)
    arrs["sd_checksum"] = torch.stack([v.double().sum() for v in sd.values()])
    save("g15_eval_full", **arrs)


def n4_fixtures():
    """G16: the reference's marching-cubes grids (utils/plots.py get_grid_uniform / get_grid) and its bubble-PDF update
    (ReconstructionTrainer.update_pdf, model/trainer/recon.py:142-152, fed as in :195-199)."""
    import importlib
    import types
    ref_import.import_reference()
    plots = importlib.import_module("utils.plots")
    trainer = importlib.import_module("model.trainer.recon").ReconstructionTrainer
    cuda_was = torch.Tensor.cuda
    torch.Tensor.cuda = lambda self, *a, **k: self          # the grid builders end in .cuda(); this container has no device
    try:
        arrs = {}
        gu = plots.get_grid_uniform(5, [-1.5, 1.5])
        arrs.update({"uni.resolution": 5, "uni.boundary": np.array([-1.5, 1.5]), "uni.grid_points": gu["grid_points"],
                     **{f"uni.xyz{a}": gu["xyz"][a] for a in range(3)}})
        g = torch.Generator().manual_seed(7)
        for case, (scale, res) in enumerate((((0.4, 1.0, 1.7), 6), ((1.2, 0.35, 2.0), 7), ((1.5, 0.9, 0.25), 5))):
            pts = torch.randn(60, 3, generator=g) * torch.tensor(scale) + torch.tensor([0.1, -0.2, 0.05])
            ga = plots.get_grid(pts, res)
            assert ga["shortest_axis_index"] == case
            arrs.update({f"al{case}.points": pts, f"al{case}.resolution": res, f"al{case}.grid_points": ga["grid_points"],
                         f"al{case}.shortest_axis_length": ga["shortest_axis_length"], f"al{case}.shortest_axis_index": ga["shortest_axis_index"],
                         **{f"al{case}.xyz{a}": ga["xyz"][a] for a in range(3)}})
    finally:
        torch.Tensor.cuda = cuda_was
    # bubble PDF: 3 images of 12x10 pixels, ~70 % of the pixels carry a point
    n_img, HW = 3, 120
    has = torch.rand(n_img * HW, generator=g) < 0.7
    links = -torch.ones(n_img * HW, dtype=torch.long)
    links[has] = torch.arange(int(has.sum()))
    n_pts = int(has.sum())
    idx = torch.randperm(n_img * HW, generator=g)[:150]
    rgb_pred, rgb_gt = torch.rand(150, 3, generator=g) * 1.6 - 0.3, torch.rand(150, 3, generator=g) * 1.4 - 0.2
    d_pred, d_gt = torch.rand(150, generator=g) * 4, torch.rand(150, generator=g) * 4
    arrs.update({"pdf.pointlinks": links, "pdf.idx": idx, "pdf.rgb_pred": rgb_pred, "pdf.rgb_gt": rgb_gt, "pdf.depth_pred": d_pred,
                 "pdf.depth_gt": d_gt, "pdf.n_points": n_pts})
    for tag, crit, pmax, prune in (("rgb", "RGB", None, 0.0), ("rgb_mp", "RGB", 0.3, 0.12), ("depth", "DEPTH", None, 0.0),
                                   ("depth_mp", "DEPTH", 1.5, 0.4)):
        self = types.SimpleNamespace(bubble_activated=True, train_dataset=types.SimpleNamespace(pointlinks=links),
                                     pdf=torch.full((n_pts,), -1.0), pdf_max=pmax, pdf_prune=prune)
        if crit == "RGB":                                    # the expressions of model/trainer/recon.py:196 / :199, evaluated by torch here
            value = (rgb_pred.detach().clamp(0, 1) - rgb_gt.clamp(0, 1)).abs().mean(dim=-1)
        else:
            value = (d_pred.detach() - d_gt).abs()
        trainer.update_pdf(self, value, idx)
        arrs.update({f"pdf.{tag}.out": self.pdf, f"pdf.{tag}.max": np.nan if pmax is None else pmax, f"pdf.{tag}.prune": prune})
    save("g16_grid_pdf", **arrs)


def main():
    if sys.argv[1:] == ["n4"]:
        return n4_fixtures()
    ref_model, ref_utils = ref_import.import_reference()
    from model.network.embedder import get_embedder
    from model.network.mlp import ImplicitNetwork, RenderingNetwork
    from model.network.density import LaplaceDensity
    from model.network import I2SDFLoss

    # ---- G1 positional encoding -------------------------------------------------------
    g = torch.Generator().manual_seed(0)
    x = (torch.rand(64, 3, generator=g) * 2 - 1) * 3.0
    e6, d6 = get_embedder("positional", input_dims=3, multires=6)
    e4, d4 = get_embedder("positional", input_dims=3, multires=4)
    save("g1_embed", x=x, pe6=e6(x), pe4=e4(x))

    # ---- G2/G3 SDF nets: forward, d sdf/dx, parameter grads of a probe loss (double backward) ----
    for name, skip in (("g2_sdf", False), ("g3_sdf_skip", True)):
        torch.manual_seed(0)
        cfg = small_conf(skip)
        net = ImplicitNetwork(64, 0.0, **cfg.implicit_network)
        perturb_(net)
        x = (torch.rand(256, 3, generator=g) * 2 - 1) * 1.5
        out = net(x.clone())
        sdf, feat, grad = net.get_outputs(x.clone())
        gw = torch.randn(feat.shape, generator=g)
        probe = ((grad.norm(2, dim=1) - 1) ** 2).sum() + sdf.sum() + (feat * gw).sum()
        net.zero_grad()
        probe.backward()
        arrs = dict(x=x, out=out, grad=grad, feat_w=gw, probe=probe)
        arrs.update(sd_arrays(net))
        arrs.update({"grad." + n: p.grad for n, p in net.named_parameters() if p.grad is not None})
        save(name, **arrs)

    # ---- G4 radiance net ----------------------------------------------------------------
    torch.manual_seed(0)
    cfg = small_conf()
    rnet = RenderingNetwork(64, **cfg.rendering_network)
    dirs = torch.nn.functional.normalize(torch.randn(128, 3, generator=g), dim=1)
    feat = torch.randn(128, 64, generator=g)
    rgb = rnet(None, None, dirs, feat.clone().requires_grad_(True))
    gw = torch.randn(rgb.shape, generator=g)
    featr = feat.clone().requires_grad_(True)
    rgb2 = rnet(None, None, dirs, featr)
    (rgb2 * gw).sum().backward()
    arrs = dict(dirs=dirs, feat=feat, rgb=rgb, rgb_w=gw, feat_grad=featr.grad)
    arrs.update(sd_arrays(rnet))
    arrs.update({"grad." + n: p.grad for n, p in rnet.named_parameters()})
    save("g4_rgb", **arrs)

    # ---- G5 Laplace density --------------------------------------------------------------
    dens = LaplaceDensity(params_init={"beta": 0.1}, beta_min=1e-4)
    s = torch.cat([torch.linspace(-2, 2, 41), torch.tensor([0.0, 1e-6, -1e-6, 50.0, -50.0])]).reshape(-1, 1)
    rows = {}
    for i, b in enumerate([0.1, 0.01, 1.0]):
        with torch.no_grad():
            dens.beta.fill_(b)
        sr = s.clone().requires_grad_(True)
        sig = dens(sr)
        gs, gb = torch.autograd.grad(sig.sum(), [sr, dens.beta])
        rows[f"sigma{i}"], rows[f"dsigma_ds{i}"], rows[f"dsum_dbeta{i}"] = sig, gs, gb
    # explicit beta override, per-row (sampler form)
    bo = torch.tensor([[0.05], [0.2]])
    rows["sigma_override"] = dens(torch.stack([s[:10, 0], s[10:20, 0]]), beta=bo)
    save("g5_density", sdf=s, betas=np.array([0.1, 0.01, 1.0]), beta_override=bo, **rows)

    # ---- G6 volume rendering -------------------------------------------------------------
    torch.manual_seed(0)
    cfg = small_conf()
    full = ref_model.I2SDFNetwork(cfg)
    z = torch.sort(torch.rand(32, 18, generator=g) * 6.0, -1)[0]
    sdf = (torch.randn(32 * 17, 1, generator=g) * 0.3).requires_grad_(True)
    w, bg_t = full.volume_rendering(z[:, :-1], z[:, -1], sdf)
    ww = torch.randn(w.shape, generator=g)
    gs, gb = torch.autograd.grad((w * ww).sum() + bg_t.sum(), [sdf, full.density.beta])
    save("g6_volume", z=z, sdf=sdf, beta_param=full.density.beta, weights=w, bg_t=bg_t, w_w=ww, grad_sdf=gs, grad_beta=gb)

    # ---- G7 sampler pieces + full eval get_z_vals ---------------------------------------
    sampler = full.ray_sampler
    # d* four triangle cases + error bound on random rows
    zr = torch.sort(torch.rand(16, 24, generator=g) * 6.0, -1)[0]
    sr = torch.randn(16, 24, generator=g) * 0.5
    sr[0, :] = sr[0, :].abs() + 5.0       # first_cond/second_cond dominated rows
    sr[1, ::2] *= 0.01
    # reproduce d_star through the reference by calling get_z_vals internals is impossible; use its formula via a probe model
    # -> G7a is pinned through the full get_z_vals below; G7b pins get_error_bound directly:
    dists = zr[:, 1:] - zr[:, :-1]
    dstar = torch.rand(16, 23, generator=g) * 0.3
    eb_scalar = sampler.get_error_bound(torch.tensor(0.05), full, sr.reshape(-1, 1), zr, dists, dstar)
    eb_rows = sampler.get_error_bound(torch.linspace(0.01, 0.5, 16).unsqueeze(-1), full, sr.reshape(-1, 1), zr, dists, dstar)
    save("g7b_error_bound", z=zr, sdf=sr, d_star=dstar, beta_scalar=np.float32(0.05), beta_rows=torch.linspace(0.01, 0.5, 16),
         eb_scalar=eb_scalar, eb_rows=eb_rows)

    arrs = {}
    for tag, t, beta_p in (("in", (0.1, -0.2, 0.3), 0.1), ("out", (0.0, 0.0, -2.0), 0.02)):
        torch.manual_seed(0)
        full = ref_model.I2SDFNetwork(small_conf())
        full.eval()
        with torch.no_grad():
            full.density.beta.fill_(beta_p)
        inp = camera_batch(1024, t, train_layout=False)
        dirs_raw, cam = ref_utils.get_camera_params(inp["uv"], inp["pose"], inp["intrinsics"])
        cam_f = cam.unsqueeze(1).repeat(1, 1024, 1).reshape(-1, 3)
        dirs = torch.nn.functional.normalize(dirs_raw.reshape(-1, 3), dim=1)
        calls = []
        orig = full.implicit_network.get_sdf_vals
        full.implicit_network.get_sdf_vals = lambda pts, _o=orig: (calls.append(pts.shape[0]), _o(pts))[1]
        zv, zeik = sampler.__class__.get_z_vals(full.ray_sampler, dirs, cam_f, full)
        arrs.update({f"{tag}.z_vals": zv, f"{tag}.iters": np.int64(len(calls)), f"{tag}.beta_param": np.float32(beta_p),
                     f"{tag}.t": np.array(t, dtype=np.float32)})
        if tag == "in":
            arrs.update(sd_arrays(full))
        out = full(inp)
        arrs.update({f"{tag}.out.{k}": v for k, v in out.items()})          # ---- G8 end-to-end eval forward
    save("g7_g8_eval", **arrs)

    # ---- G9 end-to-end train forward + loss + backward with captured draws ---------------
    for name, light in (("g9_train", False), ("g9_train_light", True)):
        torch.manual_seed(0)
        cfg = small_conf(skip=True, light=light)
        full = ref_model.I2SDFNetwork(cfg)
        perturb_(full)
        full.train()
        with torch.no_grad():
            full.density.beta.fill_(0.05)
        B = 96
        inp = camera_batch(B, (0.0, 0.2, -1.8), seed=3)
        gt = {"rgb": torch.rand(B, 3, generator=g), "depth": torch.rand(B, generator=g) * 3, "depth_mask": torch.rand(B, generator=g) > 0.2,
              "normal": torch.nn.functional.normalize(torch.randn(B, 3, generator=g), dim=1), "normal_mask": torch.rand(B, generator=g) > 0.1}
        lkw = dict(eikonal_weight=0.1, smooth_weight=0.01, smooth_iter=None, depth_weight=0.1, normal_weight=0.05)
        if light:
            gt["light_mask"] = (torch.rand(B, 1, generator=g) > 0.5).float()
            lkw["light_mask_weight"] = 0.5
        pc = (torch.rand(40, 3, generator=g) * 2 - 1)
        inp["pointcloud"] = pc
        lkw["bubble_weight"] = 0.5
        loss_fn = I2SDFLoss(**lkw)
        np.random.seed(0)
        zrec = record_z(full)
        with DrawRecorder() as rec:
            out = full(inp)
        losses = loss_fn(out, gt, 10)
        full.zero_grad()
        losses["loss"].backward()
        kinds = [k for k, _ in rec.log]
        assert kinds == ["rand", "rand", "randperm", "randint", "uniform", "uniform", "np.randint"], kinds
        arrs = {"in." + k: v for k, v in inp.items()}
        arrs.update({"gt." + k: v for k, v in gt.items()})
        arrs.update({"out." + k: v for k, v in out.items()})
        arrs.update({"loss." + k: v for k, v in losses.items()})
        arrs.update(sd_arrays(full))
        arrs.update({"grad." + n: (p.grad if p.grad is not None else torch.zeros_like(p)) for n, p in full.named_parameters()})
        arrs.update({"draw.strat_u": rec.log[0][1], "draw.cdf_u": rec.log[1][1], "draw.extra_idx": rec.log[2][1][:8],
                     "draw.eik_idx": rec.log[3][1], "draw.eik_pts": rec.log[4][1], "draw.nbr_off": rec.log[5][1]})
        arrs["loss_kwargs"] = np.array(sorted(lkw.items(), key=lambda kv: kv[0]), dtype=object).astype(str)
        arrs["ref.z_vals"], arrs["ref.z_eik"] = zrec[0]          # the reference's own depths: feeds the 1e-4 end-to-end test
        save(name, **arrs)

    # ---- G10 get_camera_params with skew ---------------------------------------------------
    inp = camera_batch(64, (0.3, -0.1, 0.7), W=640, H=480, f=600.0, seed=5, skew=2.5)
    R = torch.linalg.qr(torch.randn(3, 3, generator=g))[0]
    inp["pose"][:, :3, :3] = R
    dirs_raw, cam = ref_utils.get_camera_params(inp["uv"], inp["pose"], inp["intrinsics"])
    save("g10_camera", uv=inp["uv"], pose=inp["pose"], intrinsics=inp["intrinsics"], ray_dirs=dirs_raw, cam_loc=cam)

    # ---- G11 loss values for fixed outputs --------------------------------------------------
    B = 50
    outd = {"rgb_values": torch.rand(B, 3, generator=g), "depth_values": torch.rand(B, generator=g) * 3, "weight_sum": torch.rand(B, 1, generator=g),
            "grad_theta": torch.randn(2 * B, 3, generator=g), "diff_norm": torch.rand(B, generator=g),
            "normal_values": torch.nn.functional.normalize(torch.randn(B, 3, generator=g), dim=1), "surface_sdf": torch.randn(30, 1, generator=g) * 0.1,
            "light_mask": torch.rand(B, 1, generator=g)}
    gt = {"rgb": torch.rand(B, 3, generator=g), "depth": torch.rand(B, generator=g) * 3, "depth_mask": torch.rand(B, generator=g) > 0.3,
          "normal": torch.nn.functional.normalize(torch.randn(B, 3, generator=g), dim=1), "normal_mask": torch.rand(B, generator=g) > 0.3,
          "light_mask": (torch.rand(B, 1, generator=g) > 0.5).float()}
    arrs = {"out." + k: v for k, v in outd.items()}
    arrs.update({"gt." + k: v for k, v in gt.items()})
    cfgl = ref_import.load_cfg("synthetic.yml").loss
    l1 = I2SDFLoss(**cfgl)(outd, gt, 160000)         # shipped config, past smooth_iter, after bubble
    cfgl2 = ref_import.load_cfg("synthetic_light_mask.yml").loss
    l2 = I2SDFLoss(**cfgl2)(outd, gt, 60000)         # light-mask config, inside the bubble window
    arrs.update({"synthetic." + k: v for k, v in l1.items()})
    arrs.update({"light." + k: v for k, v in l2.items()})
    save("g11_loss", **arrs)


    # ---- G10b quaternion poses, G12 sphere intersections -----------------------------------
    g = torch.Generator().manual_seed(77)            # fresh stream: fixtures above stay byte-identical
    inp = camera_batch(64, (0.3, -0.1, 0.7), W=640, H=480, f=600.0, seed=5, skew=2.5)
    q = torch.randn(6, 4, generator=g)
    pose7 = torch.cat([q, torch.randn(6, 3, generator=g)], dim=1)
    uvq = torch.rand(6, 5, 2, generator=g) * 32
    dirs_q, cam_q = ref_utils.get_camera_params(uvq, pose7, inp["intrinsics"][:6])
    save("g10b_camera_quat", uv=uvq, pose=pose7, intrinsics=inp["intrinsics"][:6], ray_dirs=dirs_q, cam_loc=cam_q)
    o = torch.randn(200, 3, generator=g) * 0.8                                   # inside the r=3 sphere
    d = torch.nn.functional.normalize(torch.randn(200, 3, generator=g), dim=1)
    save("g12_sphere", cam_loc=o, dirs=d, r=np.float32(3.0), t=ref_utils.get_sphere_intersections(o, d, r=3.0))

    # ---- G13 ReconDataset batching (N2): __getitem__ + collate_fn on hand-filled tables ------
    import importlib
    ReconDataset = importlib.import_module("dataset.train_dataset").ReconDataset
    ds = object.__new__(ReconDataset)                  # the constructor only reads image files; the batching code reads attributes
    n_img, H, W = 3, 6, 8
    ds.n_images, ds.img_res, ds.total_pixels = n_img, [H, W], H * W
    uv_np = np.mgrid[0:H, 0:W].astype(np.int32)        # same three lines as the constructor (train_dataset.py:67-70)
    uv_np = np.flip(uv_np, axis=0).copy()
    ds.uv = torch.from_numpy(uv_np).float().reshape(2, -1).transpose(1, 0)
    cb = camera_batch(n_img, (0.2, -0.1, -2.0), W=W, H=H, f=9.0, seed=3, skew=0.3)
    ds.intrinsics_all, ds.pose_all = cb["intrinsics"].clone(), cb["pose"].clone()
    ds.intrinsics_all[:, 0, 0] += torch.arange(n_img).float()          # distinct cameras
    ds.pose_all[:, :3, 3] += torch.randn(n_img, 3, generator=g) * 0.1
    ds.rgb_images = torch.rand(n_img, H * W, 3, generator=g)
    ds.use_mask, ds.mask_images = True, (torch.rand(n_img, H * W, 1, generator=g) > 0.2).float()
    ds.use_lightmask, ds.lightmask_images = True, (torch.rand(n_img, H * W, 1, generator=g) > 0.7).float()
    ds.use_depth, ds.use_bubble = True, False
    ds.depth_images, ds.depth_masks = torch.rand(n_img, H * W, generator=g) * 4, torch.rand(n_img, H * W, generator=g) > 0.3
    ds.use_normal = True
    ds.normal_images = torch.nn.functional.normalize(torch.randn(n_img, H * W, 3, generator=g), dim=-1)
    ds.normal_masks = torch.rand(n_img, H * W, generator=g) > 0.3
    tidx = torch.randperm(n_img * H * W, generator=g)[:40]
    t_out, i_out, sample, gtb = ds.collate_fn([ds[int(i)] for i in tidx])
    dirs_b, cam_b = ref_utils.get_camera_params(sample["uv"], sample["pose"], sample["intrinsics"])
    arrs = {"tab." + k: getattr(ds, k) for k in ("intrinsics_all", "pose_all", "rgb_images", "mask_images", "lightmask_images",
                                                  "depth_images", "depth_masks", "normal_images", "normal_masks")}
    arrs.update({"sample." + k: v for k, v in sample.items()})
    arrs.update({"gt." + k: v for k, v in gtb.items()})
    save("g13_batcher", img_res=np.array([H, W]), tidx=t_out, image_idx=i_out, ray_dirs=dirs_b, cam_loc=cam_b, **arrs)

    # ---- G14 / G15 full-width (synthetic.yml shapes) end-to-end fixtures ---------------------------
    full_width(ref_model, I2SDFLoss)

    # ---- G16 marching-cubes grids + bubble-PDF update ---------------------------------
    n4_fixtures()


if __name__ == "__main__":
    main()
