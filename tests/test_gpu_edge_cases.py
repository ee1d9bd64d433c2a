"""GPU: ragged / degenerate inputs through the module and the C ABI (sizes that are not multiples of any tile, a single
ray, an empty point cloud, masked ground truth), plus the N3/N4 helpers (chunked image render, grid SDF evaluation)."""
import pytest
import torch

from oracle import i2sdf_oracle as orc
from helpers import assert_close, camera_inputs, make_gt

pytestmark = pytest.mark.gpu


def _net(train, skip=True, light=False):
    from i2sdf_amd import I2SDFNetwork, plumbing_conf
    conf = plumbing_conf(skip=skip, light=light)
    conf["use_normal"] = True
    ocfg = orc.plumbing_cfg(skip=skip, light=light)
    sd = orc.perturb_params(orc.init_params(ocfg, seed=51), 0.05, seed=52)
    sd["density.beta"] = torch.tensor(0.05)
    net = I2SDFNetwork(conf)
    net.load_state_dict(sd)
    return net.cuda().train(train), ocfg, sd


@pytest.mark.parametrize("B", [1, 3, 5, 130])
def test_tiny_and_ragged_batches_train(B):
    from i2sdf_amd import I2SDFLoss
    net, ocfg, sd = _net(True)
    inp = camera_inputs(B, (0.0, 0.2, -1.8), W=32, H=32, f=30.0, seed=B)
    gt = make_gt(B)
    if B > 2:      # an all-false mask gives nan in the reference too (mean over an empty selection)
        gt["depth_mask"][::2] = False
        gt["normal_mask"][1::3] = False
    out = net({k: v.cuda() for k, v in inp.items()})
    assert out["rgb_values"].shape == (B, 3) and out["grad_theta"].shape == (2 * B, 3) and out["diff_norm"].shape == (B,)
    loss = I2SDFLoss(eikonal_weight=0.1, depth_weight=0.1, normal_weight=0.05)(out, {k: v.cuda() for k, v in gt.items()}, 0)["loss"]
    loss.backward()
    for n, p in net.named_parameters():
        assert p.grad is not None and torch.isfinite(p.grad).all(), n
    assert torch.isfinite(loss)


def test_pointcloud_sizes():
    net, ocfg, sd = _net(True)
    B = 17
    inp = {k: v.cuda() for k, v in camera_inputs(B, (0.0, 0.2, -1.8), W=32, H=32, f=30.0, seed=1).items()}
    for n_pc in (1, 129):
        inp["pointcloud"] = (torch.rand(n_pc, 3, device="cuda") * 2 - 1)
        out = net(inp)
        assert out["surface_sdf"].shape == (n_pc, 1)
        ref = orc.sdf_forward(sd, ocfg.sdf, inp["pointcloud"].cpu())[:, :1]
        assert_close(out["surface_sdf"].detach().cpu(), ref, 2e-5, "surface_sdf")
        out["surface_sdf"].abs().mean().backward()


def test_predict_only_and_no_grad_training_mode():
    net, ocfg, sd = _net(True, light=True)
    inp = {k: v.cuda() for k, v in camera_inputs(33, (0.0, 0.2, -1.8), W=32, H=32, f=30.0, seed=2).items()}
    out = net(inp, predict_only=True)
    assert set(out) == {"rgb_values", "depth_values", "weight_sum", "light_mask"}
    with torch.no_grad():
        out = net(inp)                                        # bubble-PDF initialisation style call (model/trainer/recon.py:194)
    assert "grad_theta" in out and not out["rgb_values"].requires_grad


def test_empty_point_set_is_a_noop():
    net, _, _ = _net(False)
    out = net.implicit_network(torch.zeros(0, 3, device="cuda"))
    assert out.shape == (0, 65)


def test_render_image_equals_manual_chunk_loop_and_grid_sdf():
    net, ocfg, sd = _net(False)
    P = 700
    inp = {k: v.cuda() for k, v in camera_inputs(P, (0.0, 0.2, -1.8), W=32, H=32, f=30.0, seed=4, train_layout=False).items()}
    full = net.render_image(inp, split_n_pixels=256)
    parts = []
    with torch.no_grad():
        for lo in range(0, P, 256):
            d = dict(inp); d["uv"] = inp["uv"][:, lo:lo + 256].contiguous()
            parts.append(net(d))
    for k in full:
        assert full[k].shape[0] == P
        assert torch.equal(full[k].reshape(P, -1), torch.cat([p[k] for p in parts], 0).reshape(P, -1)), k
    pts = (torch.rand(5000, 3) * 2 - 1) * 1.5
    got = net.sdf_grid(pts.cuda(), chunk=1024)
    assert_close(got.cpu(), orc.sdf_forward(sd, ocfg.sdf, pts)[:, 0], 2e-5, "sdf_grid")


def test_first_call_under_inference_mode_then_train():
    """Lightning 1.9's validate/test loops run under torch.inference_mode() and may be the FIRST callers of the module (fit sanity
    check, `implicit_network(batch)` in the mesh test step, model/eval/recon.py:89-90): the flat parameter buffer and the engine
    built there must be ordinary tensors, and training afterwards must work."""
    from i2sdf_amd import I2SDFLoss
    net, ocfg, sd = _net(False)
    inp = {k: v.cuda() for k, v in camera_inputs(40, (0.0, 0.2, -1.8), W=32, H=32, f=30.0, seed=9, train_layout=False).items()}
    pts = (torch.rand(300, 3) * 2 - 1).cuda()
    with torch.inference_mode():
        o = net.implicit_network(pts)
        out = net(inp)
    assert o.shape == (300, 65) and out["rgb_values"].shape == (40, 3)
    assert not net._flat.is_inference()
    assert_close(o.cpu()[:, 0], orc.sdf_forward(sd, ocfg.sdf, pts.cpu())[:, 0], 2e-5, "sdf under inference_mode")
    net.train()
    tin = {k: v.cuda() for k, v in camera_inputs(24, (0.0, 0.2, -1.8), W=32, H=32, f=30.0, seed=10).items()}
    gt = {k: v.cuda() for k, v in make_gt(24).items()}
    opt = torch.optim.Adam(net.get_param_groups(1e-3), eps=1e-15)
    w0 = net._flat.clone()
    loss = I2SDFLoss(eikonal_weight=0.1, depth_weight=0.1, normal_weight=0.05)(net(tin), gt, 0)["loss"]
    loss.backward()
    opt.step()
    assert torch.isfinite(loss) and float((net._flat - w0).abs().max()) > 0


def test_ray_batcher_rejects_bad_indices():
    from i2sdf_amd import RayBatcher
    n, H, W = 2, 4, 5
    K = torch.eye(4).repeat(n, 1, 1); pose = torch.eye(4).repeat(n, 1, 1)
    rb = RayBatcher(K, pose, [H, W], rgb_images=torch.rand(n, H * W, 3))
    with pytest.raises(IndexError):
        rb.batch(torch.tensor([0, n * H * W]))                         # host indices: validated before the launch
    with pytest.raises(IndexError):
        rb.batch(torch.tensor([-1, 3]))
    t_, i_, sample, gt = rb.batch(torch.tensor([0, 7, n * H * W + 5, -3], device="cuda"))   # device indices: clamped + counted
    assert rb.bad_indices() == 2
    assert torch.isfinite(gt["rgb"]).all() and int(i_.max()) <= n - 1 and int(i_.min()) >= 0


# ---- the padding-row contract of the per-point workspaces (include/i2sdf.h; round 5 made the saved-tensor stores of padding lanes unconditional) ----
CANARY = 0x7FC0BEEF          # a quiet NaN with a payload nobody computes


def _canary_engine(conf, sd):
    """An engine whose per-point workspaces carry a canary block of 128 rows behind their last row (engine._ws is the one allocation site)."""
    import torch
    from test_gpu_train_forward import make_engine
    eng = make_engine(conf, sd)
    eng._canaries = []

    def ws(*shape, device=None):
        n = 1
        for s_ in shape:
            n *= s_
        tail = 128 * shape[-1]
        buf = torch.empty(n + tail, dtype=torch.float32, device=device)
        buf[n:].view(torch.int32).fill_(CANARY)
        eng._canaries.append((shape, buf, n))
        return buf[:n].view(*shape)

    eng._ws = ws
    return eng


def _poison_padding(t, M, blocked, p24_layers=None):
    """NaN into rows M..Mp of a saved (layers, Mp, 256) tensor (rows in the blocked layout for the first `blocked` points; layers flagged in
    p24_layers hold packed 24-bit records, csrc/x3.h P24: the padding points' upper halves become 0x7fc0 = NaN, their mid bytes 0xff)."""
    Lr, Mp, H = t.shape
    if M >= Mp:
        return
    nan = float("nan")
    if p24_layers is not None and any(p24_layers):
        assert blocked >= Mp
        nb = Mp // 32
        for l in range(Lr):
            if not p24_layers[l]:
                _poison_padding(t[l:l + 1], M, blocked)
                continue
            raw = t[l].reshape(-1)[: nb * 6144].view(torch.int32).reshape(nb, 16, 384)
            hi = raw[:, :, :256].reshape(nb, 16, 32, 8)       # [blk][kc][point][2 lanes x 4 dwords]
            mid = raw[:, :, 256:].reshape(nb, 16, 32, 4)
            b0 = M // 32
            hi[b0, :, M % 32:] = 0x7fc07fc0; mid[b0, :, M % 32:] = -1
            hi[b0 + 1:] = 0x7fc07fc0; mid[b0 + 1:] = -1
        return
    if blocked >= Mp:
        v = t.view(Lr, Mp // 32, 16, 32, 16)
        b0 = M // 32
        v[:, b0, :, M % 32:, :] = nan
        v[:, b0 + 1:] = nan
    else:
        assert blocked == 0
        t[:, M:] = nan


@pytest.mark.parametrize("M", [1, 127, 129, 51201])
def test_padding_rows_may_be_written_but_nothing_beyond_them_and_nobody_reads_them(M, wgrad_mode):
    """Forward with saves, d sdf/dx chain, radiance forward / backward, both sweeps and the weight gradients at point counts around the
    128-point workgroup: (1) no kernel writes behind row Mp of any workspace (canary block), (2) the padding rows M..Mp are never READ --
    poisoned with NaN between the producers and every consumer, all gradients stay finite and bitwise equal to the unpoisoned run."""
    from i2sdf_amd.config import synthetic_conf
    ocfg, conf = orc.synthetic_cfg(False), synthetic_conf(False)
    sd = orc.perturb_params(orc.init_params(ocfg, seed=61), 0.05, seed=62)
    n = 1
    g = torch.Generator().manual_seed(M)
    x = ((torch.rand(M, 3, generator=g) * 2 - 1) * 1.2).cuda()
    dirs = torch.nn.functional.normalize(torch.randn(M, 3, generator=g), dim=1).cuda()
    cw = torch.randn(M, 3, generator=g).cuda()
    sw = torch.randn(M, generator=g).cuda()
    results = []
    for poison in (False, True):
        eng = _canary_engine(conf, sd)
        flat = eng.layout.flat_from_state_dict(sd).cuda()
        fwd = eng.sdf_forward_grad(points=x)
        rgb, rs, pev = eng.rgb_forward(dirs, n, fwd["feat"], M)
        Mp = fwd["Mp"]
        assert Mp == (M + 127) // 128 * 128
        blk_s, blk_r = eng.blocked_points(0, M, Mp), eng.blocked_points(1, M, Mp)
        s24 = eng.saves24_points(M, Mp) == Mp          # abars / gus / gas as packed 24-bit records (gus: all layers but the unused slot 0 and the last)
        nl = fwd["abars"].shape[0]
        if poison:
            _poison_padding(fwd["hs"], M, blk_s)
            _poison_padding(fwd["abars"], M, blk_s, [True] * nl if s24 else None)
            _poison_padding(rs, M, blk_r)
            fwd["feat"][M:] = float("nan"); fwd["pe"][M:] = float("nan"); pev[M:] = float("nan")
        gar, ga_last, fbar = eng.rgb_backward(rgb, cw, rs, M)
        nvec = fwd["grad"]
        nn_ = nvec.norm(dim=1, keepdim=True)
        nbar = 2 * (nn_ - 1) * nvec / nn_
        if poison:
            _poison_padding(gar, M, blk_r)
            fbar[M:] = float("nan"); ga_last[M:] = float("nan")
        bw = eng.sdf_backward(fwd, sbar=sw, fbar=fbar, m_fbar=M, nbar=nbar)
        if poison:
            _poison_padding(bw["gus"], M, blk_s, ([False] + [True] * (nl - 1) + [False]) if s24 else None)
            _poison_padding(bw["gas"], M, blk_s, [True] * nl if s24 else None)
            for k in ("gpbar", "ga_last4", "ones4"):
                bw[k][M:] = float("nan")
        gflat = torch.zeros_like(flat)
        eng.weight_grads(flat, gflat, fwd, bw, M_main=M, fbar=fbar, rgb_fw={"pev": pev, "rs": rs}, rgb_bw={"gar": gar, "ga_last": ga_last})
        torch.cuda.synchronize()
        assert len(eng._canaries) >= 14
        for shape, buf, n_used in eng._canaries:
            tail = buf[n_used:].view(torch.int32)
            assert bool((tail == CANARY).all()), f"workspace {shape}: {int((tail != CANARY).sum())} words written behind row Mp = {Mp} (M = {M})"
        assert torch.isfinite(gflat).all(), f"M = {M}, poisoned = {poison}: a consumer read a padding row"
        assert torch.isfinite(fwd["sdf"]).all() and torch.isfinite(fwd["grad"]).all() and torch.isfinite(rgb).all()
        results.append(gflat.clone())
    assert torch.equal(results[0], results[1]), "gradients depend on the contents of the padding rows"
