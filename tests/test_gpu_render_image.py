"""GPU: full-image inference as one library call (BASELINE config 4 in small): i2sdf_render_image renders
every chunk the way the reference's chunk loop does (utils.split_input -> model(chunk) -> utils.merge_output,
utils/__init__.py:35-84; model/eval/recon.py:161-182) and writes straight into the (H*W, C) outputs.

64x48 view, the synthetic.yml networks, chunks of 700 rays (5 chunks, the last one ragged):
  * bitwise equal to the Python chunk loop over `net(chunk)` (same kernels, same chunk composition), per-chunk sampler
    iteration counts included;
  * two chunks (the first and the ragged last) against the ORACLE's chunked render: the oracle's own sampler stops after the same
    number of iterations for that chunk, and with the depths the library chose the fp64 oracle reproduces every output at 1e-4;
  * dealing whole chunks to 2 ranks and concatenating gives the same image bit for bit."""
import pytest
import torch

from oracle import i2sdf_oracle as orc
from helpers import assert_close

pytestmark = pytest.mark.gpu
D = torch.float64
W, H, CHUNK = 64, 48, 700


def _setup(light=False):
    from i2sdf_amd import I2SDFNetwork, synthetic_conf
    conf = dict(synthetic_conf(light))
    conf["use_normal"] = True
    ocfg = orc.synthetic_cfg(light)
    ocfg.use_normal = True
    sd = orc.perturb_params(orc.init_params(ocfg, seed=71), 0.03, seed=72)
    sd["density.beta"] = torch.tensor(0.02)
    net = I2SDFNetwork(conf)
    net.load_state_dict(sd)
    net = net.cuda().eval()
    K = torch.eye(4); K[0, 0] = K[1, 1] = 60.0; K[0, 2], K[1, 2] = W / 2, H / 2
    pose = torch.eye(4); pose[:3, 3] = torch.tensor([0.0, 0.0, -2.0])
    ys, xs = torch.meshgrid(torch.arange(H), torch.arange(W), indexing="ij")
    inp = {"uv": torch.stack([xs, ys], -1).float().reshape(1, -1, 2), "intrinsics": K.unsqueeze(0), "pose": pose.unsqueeze(0)}
    return net, ocfg, sd, inp


@pytest.mark.parametrize("light", [False, True])
def test_render_image_one_call_equals_chunk_loop(light):
    net, ocfg, sd, inp = _setup(light)
    cinp = {k: v.cuda() for k, v in inp.items()}
    P = W * H
    full = net.render_image(cinp, CHUNK, return_depths=True)
    iters = net.last_sampler_iters.cpu().tolist()
    assert len(iters) == (P + CHUNK - 1) // CHUNK
    parts, it_loop = [], []
    with torch.no_grad():
        for lo in range(0, P, CHUNK):
            d = dict(cinp); d["uv"] = cinp["uv"][:, lo:lo + CHUNK].contiguous()
            parts.append(net(d))
            it_loop.append(int(net.last_sampler_iters.item()))
    assert iters == it_loop, (iters, it_loop)
    keys = ["rgb_values", "depth_values", "weight_sum", "normal_map"] + (["light_mask"] if light else [])
    assert sorted(k for k in full if k != "z_vals") == sorted(keys)
    for k in keys:
        assert full[k].shape[0] == P and full[k].shape == torch.cat([p[k] for p in parts], 0).shape, k
        assert torch.equal(full[k], torch.cat([p[k] for p in parts], 0)), k
    # dealing whole chunks to two ranks changes nothing
    halves = [net.render_image(cinp, CHUNK, rank=r, world_size=2) for r in range(2)]
    assert halves[0]["rgb_values"].shape[0] == 3 * CHUNK and halves[1]["rgb_values"].shape[0] == P - 3 * CHUNK
    for k in keys:
        assert torch.equal(torch.cat([h[k] for h in halves], 0), full[k]), k
    print("per-chunk sampler iterations:", iters)
    assert len(set(iters)) > 1, "the view should make the chunks converge after different iteration counts"


def test_render_image_chunks_vs_oracle():
    net, ocfg, sd, inp = _setup(False)
    cinp = {k: v.cuda() for k, v in inp.items()}
    P = W * H
    full = net.render_image(cinp, CHUNK, return_depths=True)
    iters = net.last_sampler_iters.cpu().tolist()
    n_chunks = len(iters)
    sd64 = {k: v.to(D) for k, v in sd.items()}
    for ci in (0, n_chunks // 2, n_chunks - 1):
        lo, hi = ci * CHUNK, min((ci + 1) * CHUNK, P)
        sub = {"uv": inp["uv"][:, lo:hi], "intrinsics": inp["intrinsics"], "pose": inp["pose"]}
        # (i) the oracle's own sampler on this chunk: same chunk composition -> same (batch-global) iteration count
        tr = orc.SamplerTrace()
        cam, dirs, _ = orc.prepare_rays(sub["uv"], sub["pose"], sub["intrinsics"])
        orc.sample_z_vals(sd, ocfg, dirs, cam, training=False, trace=tr)
        assert tr.iters == iters[ci], (ci, tr.iters, iters[ci])
        # (ii) with the depths the library used, every output of the chunk's rays (a spread of 48 of them) at 1e-4 vs fp64
        S = torch.unique(torch.linspace(0, hi - lo - 1, 48).long())
        z = full["z_vals"][lo:hi].cpu()[S].to(D)
        ref = orc.network_forward(sd64, ocfg, {"uv": sub["uv"][:, S].to(D), "intrinsics": sub["intrinsics"].to(D), "pose": sub["pose"].to(D)},
                                  training=False, z_override=(z, z[:, :1]))
        # rgb / weight_sum live in [0, 1] and depth in [0, 2R]: a chunk whose rays all miss the sphere (the top and bottom rows of the
        # view) has references of ~1e-6, so the error is taken relative to max(|ref|, 1e-2) there
        for k in ("rgb_values", "depth_values", "weight_sum"):
            assert_close(full[k][lo:hi].cpu()[S], ref[k], 1e-4, f"chunk {ci} {k}", floor=1e-2)
        hit = ref["weight_sum"].reshape(-1) > 1e-2
        if bool(hit.any()):
            assert_close(full["normal_map"][lo:hi].cpu()[S][hit], ref["normal_map"][hit], 1e-4, f"chunk {ci} normal_map (weight_sum > 0.01)")
