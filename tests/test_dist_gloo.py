"""CPU, world_size 2, gloo: the data-parallel pieces of the path (flat-gradient all-reduce, parameter broadcast,
pixel sharding + gather).  The HIP kernels themselves need a GPU; what is checked here is the N>1 orchestration bench.py / a
trainer uses, on the REAL module wiring -- `tests/host_stub.HostStubNetwork` is `I2SDFNetwork` with its render core replaced by a
closed-form CPU function: flat parameter buffer, parameter views, the `grad_sync` hook call at the end of backward are the product's."""
import os
import socket

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp


def _free_port():
    s = socket.socket(); s.bind(("127.0.0.1", 0)); p = s.getsockname()[1]; s.close(); return p


def _grads(net, uv, target):
    net.zero_grad()
    out = net({"uv": uv})
    loss = ((out["rgb_values"] - target) ** 2).mean()
    loss.backward()
    return torch.cat([p.grad.reshape(-1) for p in net._param_list()]).clone()


def _worker(rank, world, port, out):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    import sys
    sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
    from host_stub import HostStubNetwork            # the REAL module wiring (flat buffer, parameter views, grad_sync hook) over a CPU stand-in core
    from i2sdf_amd import dist as i2d, plumbing_conf
    torch.manual_seed(7 + rank)                      # different initial weights per rank ...
    net = HostStubNetwork(plumbing_conf())
    net._ensure_flat()
    i2d.broadcast_parameters(net, src=0)             # ... until rank 0's are broadcast (one broadcast of the flat buffer)
    ref = HostStubNetwork(plumbing_conf())
    torch.manual_seed(7)
    ref._init_parameters()
    ref._ensure_flat()
    assert torch.equal(net._flat, ref._flat), "every rank must hold rank 0's parameters"
    assert all(p.data_ptr() == net._flat.data_ptr() + 4 * off for (_, off, _), p in zip(net.layout.entries, net._param_list()))
    i2d.attach_data_parallel(net)
    assert net.dp_state.comm is None and net.dp_state.world == world      # gloo group: torch.distributed carries the collective
    # invariant: the N-rank averaged gradient == the 1-rank gradient of the mean loss over the concatenated batch
    g = torch.Generator().manual_seed(0)
    Btot = 64
    uv = torch.rand(Btot, 1, 2, generator=g) * 30
    tgt = torch.rand(Btot, 3, generator=g)
    lo, hi = i2d.shard_range(Btot, rank, world)
    got = _grads(net, uv[lo:hi], tgt[lo:hi])
    want = _grads(ref, uv, tgt)
    assert float(want.abs().max()) > 0
    assert torch.allclose(got, want, rtol=1e-5, atol=1e-9), float((got - want).abs().max())
    # gradient accumulation: inside no_sync() the gradient stays the rank's own
    with i2d.no_sync(net):
        local = _grads(net, uv[lo:hi], tgt[lo:hi])
    own = _grads(ref, uv[lo:hi], tgt[lo:hi])
    assert torch.allclose(local, own, rtol=1e-5, atol=1e-9) and not torch.allclose(local, want, rtol=1e-3, atol=1e-9)
    # loss hook (equivalent mode): exchanges only in training mode and outside no_sync()
    from i2sdf_amd import I2SDFLoss
    lf = i2d.attach_loss(I2SDFLoss(), net)
    assert lf.exchange is None and lf._dp_state is net.dp_state           # throughput mode: no denominators to exchange
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
