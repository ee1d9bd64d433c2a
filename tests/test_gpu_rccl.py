"""GPU: the data-parallel hooks on the real backend (`nccl` = RCCL on ROCm), one rank.

A GPU box of the test pool has ONE MI355X, and RCCL refuses two ranks on one device, so what can be checked here is that
RCCL initialises next to libi2sdf_hip.so in one process (shared HIP runtime, `HSA_ENABLE_IPC_MODE_LEGACY=0`), that the
flat-gradient all-reduce inside the autograd backward, the parameter broadcast and the sampler flag reduction run on device
tensors, and that a training step with the hooks attached gives the same gradients as one without.  The N>1 arithmetic
(averaging, sharding, gather) is covered on CPU with gloo (test_dist_gloo.py); 8-GPU runs are the driver's."""
import os
import socket
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

SCRIPT = r"""
import os, sys, torch, torch.distributed as dist
sys.path.insert(0, os.environ["I2SDF_ROOT"])
torch.cuda.set_device(0)
dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda", 0))
from i2sdf_amd import I2SDFNetwork, I2SDFLoss, synthetic_conf
from i2sdf_amd import dist as i2d

def grads(attach, equivalent=False, force_iters=2):
    torch.manual_seed(0)
    conf = synthetic_conf(); conf["use_normal"] = True
    net = I2SDFNetwork(conf).cuda()
    net.train(); net.force_iters = force_iters
    if attach:
        i2d.broadcast_parameters(net, src=0)          # before the flat buffer exists: per tensor
        i2d.attach_data_parallel(net, equivalent=equivalent)     # backend nccl -> the library's own RCCL communicator (i2sdf_comm_*)
        assert net.dp_state.comm is not None and net.dp_state.comm.world == 1
    g = torch.Generator().manual_seed(3)
    B = 96
    K = torch.eye(4); K[0, 0] = K[1, 1] = 600.0; K[0, 2] = 320.0; K[1, 2] = 240.0
    pose = torch.eye(4); pose[2, 3] = -2.0
    uv = torch.stack([torch.randint(0, 640, (B,), generator=g), torch.randint(0, 480, (B,), generator=g)], -1).float().reshape(B, 1, 2)
    inp = {"uv": uv.cuda(), "intrinsics": K.repeat(B, 1, 1).cuda(), "pose": pose.repeat(B, 1, 1).cuda()}
    gt = {"rgb": torch.rand(B, 3, generator=g).cuda(), "depth": (torch.rand(B, generator=g) * 3).cuda(),
          "depth_mask": torch.ones(B, dtype=torch.bool).cuda(),
          "normal": torch.nn.functional.normalize(torch.randn(B, 3, generator=g), dim=1).cuda(),
          "normal_mask": torch.ones(B, dtype=torch.bool).cuda()}
    loss_fn = I2SDFLoss(eikonal_weight=0.1, smooth_weight=0.01, smooth_iter=150000, depth_weight=0.1, normal_weight=0.05)
    if attach:
        i2d.attach_loss(loss_fn, net)                  # equivalent mode: loss denominators through the RCCL exchange hook
    torch.manual_seed(7)                               # the module's own random draws
    out = net(inp)
    loss_fn(out, gt, 0)["loss"].backward()
    if attach:
        i2d.broadcast_parameters(net, src=0)          # now one broadcast of the flat buffer
        flag = torch.ones(1, dtype=torch.int32, device="cuda")
        assert int(i2d.global_any(flag)) == 1
        assert i2d.gather_outputs({"x": out["rgb_values"].detach()}, B)["x"].shape == (B, 3)
    return torch.cat([p.grad.reshape(-1) for p in net.parameters()])

a, b = grads(False), grads(True)
dist.barrier()
torch.cuda.synchronize()
assert torch.isfinite(a).all() and float(a.abs().max()) > 0
assert torch.equal(a, b), float((a - b).abs().max())
# 1-GPU-equivalent mode, data-dependent sampler loop: the per-iteration flag MAX-reduce, the shared-column broadcast and the loss
# denominators all go through RCCL (identity on one rank) -> bitwise the same step
# (equivalent mode runs the loss through the SEPARATE entry points -- the exchange of the denominators sits between its two launches --
# where the default single-GPU step takes the fused loss + render-backward call of round 6: same numbers up to the order of a few sums.
# Bitwise against the separate path, 2e-6 against the fused one.)
c_fused = grads(False, force_iters=0)
os.environ["I2SDF_FUSED_RENDER_LOSS"] = "0"
c = grads(False, force_iters=0)
os.environ.pop("I2SDF_FUSED_RENDER_LOSS")
d = grads(True, equivalent=True, force_iters=0)
torch.cuda.synchronize()
assert torch.equal(c, d), float((c - d).abs().max())
assert float((c_fused - d).abs().max()) <= 2e-6 * float(d.abs().max()), float((c_fused - d).abs().max() / d.abs().max())
dist.destroy_process_group()
print("RCCL_OK")
"""


def test_rccl_single_rank_hooks():
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK="0", WORLD_SIZE="1", LOCAL_RANK="0", I2SDF_ROOT=ROOT)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    r = subprocess.run([sys.executable, "-c", SCRIPT], env=env, capture_output=True, text=True, timeout=420)
    assert r.returncode == 0 and "RCCL_OK" in r.stdout, (r.stdout[-2000:], r.stderr[-4000:])
