"""GPU: SURVEY 8(f) row N4 -- the SDF volume for marching cubes (i2sdf_sdf_grid) and the bubble PDF (i2sdf_pdf_update, the initial
sweep) against the oracle and the reference's own grids / PDF updates (tests/golden/g16_grid_pdf.npz)."""
import numpy as np
import pytest
import torch

from oracle import i2sdf_oracle as orc
from helpers import t, assert_close, make_draws
from test_gpu_edge_cases import _net

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("case", [0, 1, 2])
def test_sdf_volume_on_the_references_grids(golden, case):
    """Flat order == the reference's grid_points order; volume order == its reshape(ny,nx,nz).transpose(1,0,2); chunk boundaries
    and rank slabs do not matter; the points themselves never exist on the host side of this call."""
    from i2sdf_amd import grid as G
    z = golden("g16_grid_pdf")
    net, ocfg, sd = _net(False)
    pts = t(z[f"al{case}.grid_points"])
    ax = G.aligned_axes(t(z[f"al{case}.points"]), int(z[f"al{case}.resolution"]))
    want = orc.sdf_forward(sd, ocfg.sdf, pts)[:, 0]
    flat = net.sdf_volume(ax, order="meshgrid", chunk=1000)                    # several ragged chunks
    assert flat.shape == (pts.shape[0],)
    assert_close(flat.cpu(), want, 2e-5, "flat grid sdf")
    vol = net.sdf_volume(ax, chunk=1 << 20)
    assert tuple(vol.shape) == ax.shape_volume
    assert torch.equal(vol, orc.grid_volume(flat, ax.xyz).contiguous())        # same points, same kernel: bit-identical
    slabs = [net.sdf_volume(ax, order="meshgrid", chunk=512, rank=r, world_size=3) for r in range(3)]
    assert torch.equal(torch.cat(slabs), flat)


def test_sdf_volume_uniform_and_aligned_transform(golden):
    from i2sdf_amd import grid as G
    z = golden("g16_grid_pdf")
    net, ocfg, sd = _net(False, skip=False)
    ax = G.uniform_axes(int(z["uni.resolution"]), z["uni.boundary"])
    pts = t(z["uni.grid_points"])
    got = net.sdf_volume(ax, order="meshgrid")
    assert_close(got.cpu(), orc.sdf_forward(sd, ocfg.sdf, pts)[:, 0], 2e-5, "uniform grid")
    # the PCA-aligned fine grid of model/eval/recon.py:75-90: grid in the eigen-frame, evaluated at vecs^T p + s_mean
    g = torch.Generator().manual_seed(3)
    q, _ = torch.linalg.qr(torch.randn(3, 3, generator=g))
    s_mean = torch.tensor([0.05, -0.1, 0.2])
    ax = G.aligned_axes(t(z["al1.points"]) * 0.5, 9)
    o = orc.get_grid(t(z["al1.points"]) * 0.5, 9)
    world = orc.align_grid_points(o["grid_points"], q, s_mean)
    got = net.sdf_volume(ax, rot=q.t(), trans=s_mean, order="meshgrid", chunk=700)
    assert_close(got.cpu(), orc.sdf_forward(sd, ocfg.sdf, world)[:, 0], 2e-5, "aligned grid")
    only_shift = net.sdf_volume(ax, trans=s_mean, order="meshgrid")
    assert_close(only_shift.cpu(), orc.sdf_forward(sd, ocfg.sdf, o["grid_points"] + s_mean)[:, 0], 2e-5, "shifted grid")


def test_sdf_grid_argument_checks():
    net, _, _ = _net(False)
    a = torch.linspace(-1, 1, 4, device="cuda")
    eng = net._engine_for(a.device)
    assert eng.sdf_grid(a, a, a, count=0).shape == (0,)
    from i2sdf_amd.lib import I2SDFError
    with pytest.raises(I2SDFError):
        eng.sdf_grid(a, a, a, first=60, count=10)             # past the end of the 64-point grid
    with pytest.raises(I2SDFError):
        eng.sdf_grid(a, a, a, order=7)


@pytest.mark.parametrize("tag,crit", [("rgb", "RGB"), ("rgb_mp", "RGB"), ("depth", "DEPTH"), ("depth_mp", "DEPTH")])
def test_pdf_update_matches_the_reference(golden, tag, crit):
    from i2sdf_amd import BubblePDF
    z = golden("g16_grid_pdf")
    pmax = None if np.isnan(z[f"pdf.{tag}.max"]) else float(z[f"pdf.{tag}.max"])
    prune = float(z[f"pdf.{tag}.prune"])
    n_pts = int(z["pdf.n_points"])
    bp = BubblePDF(torch.zeros(n_pts, 3), t(z["pdf.pointlinks"]), crit, pmax, prune)
    bp.pdf.fill_(-1.0)
    out = {"rgb_values": t(z["pdf.rgb_pred"]).cuda(), "depth_values": t(z["pdf.depth_pred"]).cuda()}
    gt = {"rgb": t(z["pdf.rgb_gt"]).cuda(), "depth": t(z["pdf.depth_gt"]).cuda()}
    bp.update_pdf(out, gt, t(z["pdf.idx"]).cuda())
    want, got = t(z[f"pdf.{tag}.out"]), bp.pdf.cpu()
    assert torch.equal(got == -1, want == -1)                                 # exactly the linked points of the batch were written
    # values agree to an ulp of the 3-term mean; the prune decision may only differ for values within an ulp of the threshold
    near = ((want - prune).abs() < 1e-6) | ((got - prune).abs() < 1e-6)
    assert near.sum() <= 1
    assert torch.allclose(got[~near], want[~near], rtol=0, atol=2e-7), float((got - want).abs().max())
    assert bp.bad_indices() == 0
    bp.update_pdf(out, gt, torch.full((150,), 10 ** 6, dtype=torch.int64))    # out-of-range pixels: skipped and counted
    assert bp.bad_indices() == 150 and torch.equal(bp.pdf.cpu(), got)


def test_initialize_bubble_pdf_sweep_vs_oracle():
    """The initial sweep (model/trainer/recon.py:172-199): every pixel of every image through model.forward(data, True) in TRAINING
    mode under no_grad, error scattered into the PDF -- against the oracle doing the same split by split with the same draws."""
    from i2sdf_amd import BubblePDF, RayBatcher
    net, ocfg, sd = _net(True)
    n_img, H, W, split = 2, 10, 12, 50
    g = torch.Generator().manual_seed(5)
    K = torch.eye(4).repeat(n_img, 1, 1)
    K[:, 0, 0] = K[:, 1, 1] = 12.0
    K[:, 0, 2], K[:, 1, 2] = W / 2, H / 2
    pose = torch.eye(4).repeat(n_img, 1, 1)
    pose[0, :3, 3], pose[1, :3, 3] = torch.tensor([0.0, 0.2, -1.8]), torch.tensor([0.3, -0.1, -1.6])
    tables = {"intrinsics_all": K, "pose_all": pose, "rgb_images": torch.rand(n_img, H * W, 3, generator=g),
              "depth_images": torch.rand(n_img, H * W, generator=g) * 3, "depth_masks": torch.rand(n_img, H * W, generator=g) > 0.25}
    links = -torch.ones(n_img * H * W, dtype=torch.long)
    valid = tables["depth_masks"].reshape(-1)
    links[valid] = torch.arange(int(valid.sum()))
    n_pts = int(valid.sum())
    rb = RayBatcher(K, pose, [H, W], rgb_images=tables["rgb_images"], depth_images=tables["depth_images"], depth_masks=tables["depth_masks"])
    n_row = ocfg.sampler.N_samples_eval + ocfg.sampler.N_samples

    def draws(i, lo, n):
        return make_draws(ocfg, n, n_row=n_row, seed=1000 * i + lo)

    for crit, pmax, prune in (("DEPTH", 1.0, 0.2), ("RGB", None, 0.05)):
        bp = BubblePDF(torch.zeros(n_pts, 3), links, crit, pmax, prune)
        bp.initialize_bubble_pdf(net, rb, split, draws_for=lambda i, lo, n: {k: getattr(draws(i, lo, n), k).cuda() for k in
                                                                             ("strat_u", "cdf_u", "extra_idx", "eik_idx", "eik_pts", "nbr_off")})
        want = torch.zeros(n_pts)
        vals = torch.zeros(n_img * H * W)
        for i in range(n_img):
            for lo in range(0, H * W, split):
                n = min(split, H * W - lo)
                tidx = torch.arange(i * H * W + lo, i * H * W + lo + n)
                _, _, sample, gt = orc.ray_batch(tables, [H, W], tidx)
                out = orc.network_forward(sd, ocfg, sample, True, draws(i, lo, n), predict_only=True)
                v = orc.pdf_error(crit, out, gt)
                vals[tidx] = v
                orc.update_pdf(want, v, tidx, links, pmax, prune)
        got = bp.pdf.cpu()
        # the error is a difference of O(1) quantities: compare at the 1e-4 bar relative to the largest error, except where the
        # prune threshold / clamp sits inside that band
        v_pts = vals[valid]
        band = 1e-4 * float(v_pts.abs().max())
        edge = (v_pts - prune).abs() < 2 * band
        assert edge.float().mean() < 0.05
        assert float((got - want)[~edge].abs().max()) <= band, (crit, float((got - want)[~edge].abs().max()), band)
        assert (got > 0).sum() > n_pts // 4
        s = bp.sample_bubble(16)
        assert s.shape == (16, 3) and float(bp.sample_count.sum()) == 16
