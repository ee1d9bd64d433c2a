"""Data parallelism for the render core: one process per GPU, rays sharded, ONE collective per training step.

The reference has no distributed code at all.  Rays are independent, so the only
exchange step of the path is the parameter-gradient sum: the backward writes every gradient into one flat fp32 buffer
(800 955 floats = 3.2 MB at synthetic.yml shapes) and this module all-reduces that buffer once -- RCCL over xGMI on
MI355X (`backend="nccl"` is RCCL on ROCm), gloo on CPU for the tests.  Inference shards the pixel list; results are
concatenated (no collective on the data path)."""
from __future__ import annotations

from typing import Dict, List, Tuple

import torch
import torch.distributed as dist


def attach_data_parallel(net, group=None):
    """Make `net` (i2sdf_amd.I2SDFNetwork) average its flat gradient over the process group inside backward."""
    world = dist.get_world_size(group)

    def sync(flat_grad: torch.Tensor):
        # always through the collective once attached (a 1-rank group is a no-op for RCCL): the N=1 and N>1 code paths are the same
        dist.all_reduce(flat_grad, op=dist.ReduceOp.SUM, group=group)
        if world > 1:
            flat_grad.mul_(1.0 / world)

    net.grad_sync = sync
    return net


def broadcast_parameters(net, src: int = 0, group=None):
    """Replicate rank `src`'s parameters (one broadcast of the flat buffer when it exists, else per tensor)."""
    flat = getattr(net, "_flat", None)
    if flat is not None:
        dist.broadcast(flat, src=src, group=group)
    else:
        for p in net.parameters():
            dist.broadcast(p.data, src=src, group=group)


def shard_range(total: int, rank: int, world: int) -> Tuple[int, int]:
    """Contiguous slice [lo, hi) of `total` rays for `rank`: ceil(total/world) per rank."""
    per = (total + world - 1) // world
    lo = min(rank * per, total)
    return lo, min(lo + per, total)


def shard_pixels(model_input: Dict[str, torch.Tensor], rank: int, world: int) -> Dict[str, torch.Tensor]:
    """Eval layout (uv is (1, P, 2)): keep this rank's pixel slice; pose / intrinsics are shared."""
    P = model_input["uv"].shape[1]
    lo, hi = shard_range(P, rank, world)
    out = dict(model_input)
    out["uv"] = model_input["uv"][:, lo:hi].contiguous()
    return out


def gather_outputs(outputs: Dict[str, torch.Tensor], total: int, group=None) -> Dict[str, torch.Tensor]:
    """Concatenate per-rank render outputs back into image order (all ranks get the full result)."""
    world = dist.get_world_size(group)
    if world == 1:
        return outputs
    per = (total + world - 1) // world
    merged = {}
    for k, v in outputs.items():
        pad = torch.zeros((per,) + tuple(v.shape[1:]), dtype=v.dtype, device=v.device)
        pad[: v.shape[0]] = v
        parts = [torch.empty_like(pad) for _ in range(world)]
        dist.all_gather(parts, pad, group=group)
        merged[k] = torch.cat(parts, 0)[:total]
    return merged


def global_any(flag: torch.Tensor, group=None) -> torch.Tensor:
    """MAX-reduce a small flag tensor (e.g. the sampler's `not converged`) when bit-identical 1-GPU-equivalent
    sampling across shards is wanted; throughput runs use per-rank flags instead."""
    if dist.get_world_size(group) > 1:
        dist.all_reduce(flag, op=dist.ReduceOp.MAX, group=group)
    return flag
