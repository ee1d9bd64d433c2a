"""CPU, world_size 2, gloo: the data-parallel pieces of the path (flat-gradient all-reduce, parameter broadcast,
pixel sharding + gather).  The HIP kernels themselves need a GPU; what is checked here is the N>1 orchestration
bench.py / a trainer uses, with a stand-in per-rank "backward" that fills the flat gradient buffer."""
import os
import socket

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp


def _free_port():
    s = socket.socket(); s.bind(("127.0.0.1", 0)); p = s.getsockname()[1]; s.close(); return p


class _FakeNet:
    """Quacks like I2SDFNetwork for i2sdf_amd.dist: a flat parameter buffer + a grad_sync hook called inside backward."""

    def __init__(self, n):
        self._flat = torch.arange(n, dtype=torch.float32)
        self.grad_sync = None

    def parameters(self):
        return [self._flat]

    def backward(self, per_ray_grads):
        g = per_ray_grads.sum(0)                 # the kernels sum the gradient over this rank's rays
        if self.grad_sync is not None:
            self.grad_sync(g)
        return g


def _worker(rank, world, port, out):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from i2sdf_amd import dist as i2d
    n = 1001
    net = _FakeNet(n)
    if rank != 0:
        net._flat += 100.0                       # diverged replica
    i2d.broadcast_parameters(net, src=0)
    assert torch.equal(net._flat, torch.arange(n, dtype=torch.float32))
    i2d.attach_data_parallel(net)
    # invariant: N-rank averaged gradient == 1-rank gradient of the mean loss over the concatenated batch
    g = torch.Generator().manual_seed(0)
    all_rays = torch.randn(64, n, generator=g)   # per-ray gradient contributions of a mean-type loss, already / local batch
    lo, hi = i2d.shard_range(64, rank, world)
    local = all_rays[lo:hi] / (hi - lo)
    got = net.backward(local)
    want = all_rays.sum(0) / 64
    assert torch.allclose(got, want, rtol=1e-5, atol=1e-6)
    # inference: shard pixels, gather outputs back in image order
    P = 37
    inp = {"uv": torch.arange(P * 2, dtype=torch.float32).reshape(1, P, 2), "pose": torch.eye(4).unsqueeze(0), "intrinsics": torch.eye(4).unsqueeze(0)}
    mine = i2d.shard_pixels(inp, rank, world)
    lo, hi = i2d.shard_range(P, rank, world)
    assert mine["uv"].shape[1] == hi - lo and torch.equal(mine["uv"][0], inp["uv"][0, lo:hi])
    outs = {"rgb_values": mine["uv"][0].repeat(1, 2)[:, :3], "depth_values": mine["uv"][0, :, 0]}
    full = i2d.gather_outputs(outs, P)
    assert torch.equal(full["depth_values"], inp["uv"][0, :, 0]) and full["rgb_values"].shape == (P, 3)
    flag = torch.tensor([1 if rank == 1 else 0], dtype=torch.int32)
    assert int(i2d.global_any(flag)) == 1
    out.put((rank, True))
    dist.destroy_process_group()


def test_two_rank_gloo():
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    for p in procs:
        p.join(120)
        assert p.exitcode == 0
    got = sorted(q.get(timeout=5) for _ in range(2))
    assert got == [(0, True), (1, True)]


def test_shard_range_partitions():
    from i2sdf_amd.dist import shard_range
    for total in (0, 1, 7, 307200):
        for world in (1, 2, 3, 8):
            spans = [shard_range(total, r, world) for r in range(world)]
            assert spans[0][0] == 0 and spans[-1][1] == total
            assert all(a[1] == b[0] for a, b in zip(spans[:-1], spans[1:]))
