"""GPU: the HBM-resident ray batcher, quaternion poses and sphere intersections vs the golden vectors
(reference ReconDataset.__getitem__/collate_fn, get_camera_params, get_sphere_intersections) and the oracle."""
import pytest
import torch

from oracle import i2sdf_oracle as orc
from helpers import assert_close, t

pytestmark = pytest.mark.gpu


def _engine():
    from i2sdf_amd.config import NetConfig, plumbing_conf
    from i2sdf_amd.engine import RenderEngine
    return RenderEngine(NetConfig.from_conf(plumbing_conf()))


def _batcher(z):
    from i2sdf_amd.batcher import RayBatcher
    tab = {k[4:]: t(z[k]) for k in z.files if k.startswith("tab.")}
    return RayBatcher(tab["intrinsics_all"], tab["pose_all"], [int(v) for v in z["img_res"]], rgb_images=tab["rgb_images"],
                      mask_images=tab["mask_images"], lightmask_images=tab["lightmask_images"], depth_images=tab["depth_images"],
                      depth_masks=tab["depth_masks"], normal_images=tab["normal_images"], normal_masks=tab["normal_masks"])


def test_ray_batch_golden(golden):
    """Same batch as the reference's dataset + collate_fn produce: indices, sample dict, every ground-truth entry bit-exact;
    the rays equal get_camera_params on the reference's per-ray K / pose stacks."""
    z = golden("g13_batcher")
    rb = _batcher(z)
    tidx, idx, sample, gt = rb.batch(t(z["tidx"]))
    assert torch.equal(tidx.cpu(), t(z["tidx"])) and torch.equal(idx.cpu(), t(z["image_idx"]))
    assert torch.equal(sample["uv"].cpu(), t(z["sample.uv"]))
    assert torch.equal(sample["intrinsics"].cpu(), t(z["sample.intrinsics"]))          # gathered lazily from the tables
    assert torch.equal(sample["pose"].cpu(), t(z["sample.pose"]))
    names = [k[3:] for k in z.files if k.startswith("gt.")]
    assert sorted(names) == sorted(gt)
    for k in names:
        ref = t(z["gt." + k])
        assert gt[k].dtype == ref.dtype and gt[k].shape == ref.shape, k
        assert torch.equal(gt[k].cpu(), ref), k
    raw = t(z["ray_dirs"]).reshape(-1, 3)
    r = sample["rays"]
    assert_close(r["dnorm"].cpu(), raw.norm(dim=1), 1e-6, "||d||")
    assert_close(r["dirs"].cpu(), torch.nn.functional.normalize(raw, dim=1), 1e-6, "dirs")
    assert torch.equal(r["cam_loc"].cpu(), t(z["cam_loc"]))


def test_ray_batch_matches_ray_setup_and_network_input(golden):
    """The batcher's rays are bit-identical to i2sdf_ray_setup on the expanded sample, so a network fed with the compact
    sample renders exactly what it renders from the reference-layout dict."""
    from i2sdf_amd.config import plumbing_conf
    from i2sdf_amd.network import I2SDFNetwork
    z = golden("g13_batcher")
    rb = _batcher(z)
    _, _, sample, _ = rb.batch(t(z["tidx"]))
    cam, dirs, dn = _engine().ray_setup(sample["uv"], sample["pose"], sample["intrinsics"])
    r = sample["rays"]
    assert torch.equal(cam, r["cam_loc"]) and torch.equal(dirs, r["dirs"]) and torch.equal(dn, r["dnorm"])
    net = I2SDFNetwork(plumbing_conf()).cuda().eval()
    with torch.no_grad():
        a = net(sample)
        b = net({k: sample[k] for k in ("uv", "intrinsics", "pose")})
    for k in a:
        assert torch.equal(a[k], b[k]), k


def test_ray_batch_epoch_is_a_partition():
    """epoch(): every pixel of every image exactly once; two ranks get disjoint halves (no collective)."""
    from i2sdf_amd.batcher import RayBatcher
    n, H, W = 2, 5, 7
    K = torch.eye(4).repeat(n, 1, 1); K[:, 0, 0] = K[:, 1, 1] = 10.0
    rb = RayBatcher(K, torch.eye(4).repeat(n, 1, 1), [H, W], rgb_images=torch.rand(n, H * W, 3))
    seen = torch.cat([b[0] for b in rb.epoch(16, generator=torch.Generator().manual_seed(1))])
    assert torch.equal(seen.sort().values.cpu(), torch.arange(n * H * W))
    halves = [torch.cat([b[0] for b in rb.epoch(16, generator=torch.Generator().manual_seed(1), rank=r, world_size=2)]) for r in range(2)]
    both = torch.cat(halves).sort().values.cpu()
    assert torch.equal(both, torch.arange(n * H * W)) and halves[0].shape == halves[1].shape
    assert len(list(rb.epoch(16, generator=torch.Generator().manual_seed(1), drop_last=True))) == (n * H * W) // 16
    tidx, idx, sample, gt = rb.batch(torch.empty(0, dtype=torch.int64))               # empty batch
    assert sample["uv"].shape == (0, 1, 2) and gt["rgb"].shape == (0, 3)


def test_ray_setup_quaternion_golden(golden):
    z = golden("g10b_camera_quat")
    cam, dirs, dn = _engine().ray_setup(t(z["uv"]).cuda(), t(z["pose"]).cuda(), t(z["intrinsics"]).cuda())
    raw = t(z["ray_dirs"]).reshape(-1, 3)
    P = z["uv"].shape[1]
    assert_close(dn.cpu(), raw.norm(dim=1), 2e-6, "||d||")
    assert_close(dirs.cpu(), torch.nn.functional.normalize(raw, dim=1), 2e-6, "dirs")
    assert torch.equal(cam.cpu(), t(z["cam_loc"]).repeat_interleave(P, dim=0))
    # quaternion tables in the batcher
    from i2sdf_amd.batcher import RayBatcher
    rb = RayBatcher(t(z["intrinsics"]), t(z["pose"]), [4, 4])
    _, _, s, _ = rb.batch(torch.arange(6 * 16))
    d0, c0 = orc.get_camera_params(orc.pixel_uv(4, 4).unsqueeze(0).repeat(6, 1, 1).double(), t(z["pose"]).double(), t(z["intrinsics"]).double())
    assert_close(s["rays"]["dirs"].cpu(), torch.nn.functional.normalize(d0.reshape(-1, 3), dim=1), 2e-6, "batcher dirs (quaternion)")


def test_sphere_intersections_golden(golden):
    z = golden("g12_sphere")
    eng = _engine()
    out = eng.sphere_intersections(t(z["cam_loc"]).cuda(), t(z["dirs"]).cuda(), float(z["r"]))
    assert_close(out.cpu(), z["t"], 1e-6, "sphere intersections")
    with pytest.raises(ValueError):                       # the reference prints and exit()s here (rend_util.py:220-222)
        eng.sphere_intersections(torch.tensor([[5.0, 0, 0]]).cuda(), torch.tensor([[0.0, 1, 0]]).cuda(), 3.0)
