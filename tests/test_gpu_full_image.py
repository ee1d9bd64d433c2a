"""GPU: BASELINE config 4 at its full size -- the 640x480 view (307 200 rays) of the synthetic.yml networks rendered by ONE
`i2sdf_render_image(..., 12000)` call: 26 chunks, the last one ragged (7 200 rays), one workspace reused by all of them.
Replaces the reference's chunk loop utils.split_input -> model(chunk) -> utils.merge_output (utils/__init__.py:35-84,
model/eval/recon.py:161-182).

  * properties of all 307 200 rays (finite, ranges, sorted depths, per-chunk iteration counts inside 1..max_total_iters);
  * bit-equality with the Python chunk loop over `net(chunk)` -- same kernels, same chunk composition -- per-chunk sampler
    iteration counts included;
  * the first, a middle and the ragged last chunk against the fp64 oracle on a ray subset, with the depths the library chose
    (1e-4 max-norm relative), and for the ragged last chunk the oracle's OWN sampler must stop after the same number of iterations
    (the convergence test is chunk-global, ray_sampler.py:151: the chunk composition is what decides it)."""
import pytest
import torch

from oracle import i2sdf_oracle as orc
from helpers import assert_close

pytestmark = pytest.mark.gpu
D = torch.float64
W, H, CHUNK = 640, 480, 12000


def _setup():
    from i2sdf_amd import I2SDFNetwork, synthetic_conf
    conf = dict(synthetic_conf(False))
    conf["use_normal"] = True
    ocfg = orc.synthetic_cfg(False)
    ocfg.use_normal = True
    sd = orc.perturb_params(orc.init_params(ocfg, seed=81), 0.03, seed=82)
    sd["density.beta"] = torch.tensor(0.02)
    net = I2SDFNetwork(conf)
    net.load_state_dict(sd)
    net = net.cuda().eval()
    K = torch.eye(4); K[0, 0] = K[1, 1] = 600.0; K[0, 2], K[1, 2] = W / 2, H / 2          # BASELINE.md camera (ii)
    pose = torch.eye(4); pose[:3, 3] = torch.tensor([0.0, 0.0, -2.0])
    ys, xs = torch.meshgrid(torch.arange(H), torch.arange(W), indexing="ij")
    inp = {"uv": torch.stack([xs, ys], -1).float().reshape(1, -1, 2), "intrinsics": K.unsqueeze(0), "pose": pose.unsqueeze(0)}
    return net, ocfg, sd, inp


def test_full_640x480_view_in_one_call():
    net, ocfg, sd, inp = _setup()
    cinp = {k: v.cuda() for k, v in inp.items()}
    P = W * H
    n_chunks = (P + CHUNK - 1) // CHUNK
    assert n_chunks == 26 and P - (n_chunks - 1) * CHUNK == 7200
    full = net.render_image(cinp, CHUNK, return_depths=True)
    iters = net.last_sampler_iters.cpu().tolist()
    sc = ocfg.sampler
    # ---- every ray
    assert len(iters) == n_chunks and all(1 <= i <= sc.max_total_iters for i in iters), iters
    assert full["rgb_values"].shape == (P, 3) and full["depth_values"].shape == (P,) and full["weight_sum"].shape == (P, 1)
    assert full["normal_map"].shape == (P, 3) and full["z_vals"].shape == (P, sc.N_samples + sc.N_samples_extra + 2)
    for k, v in full.items():
        assert torch.isfinite(v).all(), k
    assert float(full["rgb_values"].min()) >= 0.0 and float(full["rgb_values"].max()) <= 1.0
    assert float(full["weight_sum"].min()) >= 0.0 and float(full["weight_sum"].max()) <= 1.0 + 1e-5
    assert float(full["depth_values"].min()) >= 0.0 and float(full["depth_values"].max()) <= 2.0 * ocfg.scene_bounding_sphere * 1.0001
    z = full["z_vals"]
    assert bool((z[:, 1:] >= z[:, :-1]).all()), "depth rows must be sorted"
    assert float(z[:, 0].min()) >= 0.0 and torch.equal(z[:, -1], torch.full_like(z[:, -1], 2.0 * ocfg.scene_bounding_sphere))
    hit = full["weight_sum"].reshape(-1) > 0.5
    assert 0.05 < float(hit.float().mean()) < 0.95, "the view should contain both geometry and background"
    nn = full["normal_map"][hit].norm(dim=1)
    assert 0.0 < float(nn.min()) and float(nn.max()) < 10.0, "normals of rays that hit the surface: weighted sums of O(1) SDF gradients"
    # ---- the Python chunk loop (what the reference's eval loop does), bit for bit
    keys = ["rgb_values", "depth_values", "weight_sum", "normal_map"]
    it_loop = []
    with torch.no_grad():
        for ci, lo in enumerate(range(0, P, CHUNK)):
            d = dict(cinp); d["uv"] = cinp["uv"][:, lo:lo + CHUNK].contiguous()
            o = net(d)
            it_loop.append(int(net.last_sampler_iters.item()))
            for k in keys:
                assert torch.equal(full[k][lo:lo + CHUNK], o[k]), (ci, k)
    assert iters == it_loop, (iters, it_loop)
    print("per-chunk sampler iterations:", iters)
    # ---- first / middle / ragged last chunk against the fp64 oracle on a ray subset (depths: the library's)
    sd64 = {k: v.to(D) for k, v in sd.items()}
    for ci in (0, n_chunks // 2, n_chunks - 1):
        lo, hi = ci * CHUNK, min((ci + 1) * CHUNK, P)
        S = torch.unique(torch.linspace(0, hi - lo - 1, 64).long())
        sub_uv = inp["uv"][:, lo:hi][:, S]
        zc = full["z_vals"][lo:hi].cpu()[S].to(D)
        ref = orc.network_forward(sd64, ocfg, {"uv": sub_uv.to(D), "intrinsics": inp["intrinsics"].to(D), "pose": inp["pose"].to(D)},
                                  training=False, z_override=(zc, zc[:, :1]))
        for k in ("rgb_values", "depth_values", "weight_sum"):
            assert_close(full[k][lo:hi].cpu()[S], ref[k], 1e-4, f"chunk {ci} {k}", floor=1e-2)
        h = ref["weight_sum"].reshape(-1) > 1e-2
        if bool(h.any()):
            assert_close(full["normal_map"][lo:hi].cpu()[S][h], ref["normal_map"][h], 1e-4, f"chunk {ci} normal_map (weight_sum > 0.01)")
    # ---- the oracle's own sampler on the ragged last chunk (7 200 rays): same chunk composition -> same iteration count
    lo = (n_chunks - 1) * CHUNK
    tr = orc.SamplerTrace()
    sdc = {k: v.cuda() for k, v in sd.items()}                 # (the restatement's torch ops on the device: 7 200 x 128 x k points)
    cam, dirs, _ = orc.prepare_rays(cinp["uv"][:, lo:], cinp["pose"], cinp["intrinsics"])
    orc.sample_z_vals(sdc, ocfg, dirs, cam, training=False, trace=tr)
    assert tr.iters == iters[-1], (tr.iters, iters[-1])
