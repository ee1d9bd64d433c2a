"""GPU: FusedAdam (i2sdf_adam_step: one HIP launch over the flat parameter buffer) vs torch.optim.Adam, the reference's optimizer
(model/trainer/recon.py:201-206), including ExponentialLR and a state_dict round trip."""
import copy

import pytest
import torch

from helpers import camera_inputs, make_gt

pytestmark = pytest.mark.gpu


def _net(seed=0):
    from i2sdf_amd import I2SDFNetwork, plumbing_conf
    conf = plumbing_conf(skip=True)
    conf["use_normal"] = True
    torch.manual_seed(seed)
    return I2SDFNetwork(conf).cuda().train()


def test_fused_adam_matches_torch_adam_on_identical_gradients():
    """Same gradient sequence into both optimizers (no rendering involved): weights agree to a few ulps after 10 steps."""
    from i2sdf_amd import FusedAdam
    a, b = _net(), _net()
    a._ensure_flat(); b._ensure_flat()
    assert torch.equal(a._flat, b._flat)
    oa = FusedAdam(a.get_param_groups(5e-4), eps=1e-15)
    ob = torch.optim.Adam(b.get_param_groups(5e-4), eps=1e-15)
    sa = torch.optim.lr_scheduler.ExponentialLR(oa, 0.9)
    sb = torch.optim.lr_scheduler.ExponentialLR(ob, 0.9)
    g = torch.Generator(device="cuda").manual_seed(1)
    for it in range(10):
        gflat = torch.randn(a._flat.numel(), device="cuda", generator=g) * (10.0 ** -(it % 4))
        off = 0
        for pa, pb in zip(a._param_list(), b._param_list()):
            k = pa.numel()
            pa.grad = gflat[off:off + k].view(pa.shape)
            pb.grad = gflat[off:off + k].view(pb.shape).clone()
            off += k
        oa.step(); ob.step(); sa.step(); sb.step()
    err = float((a._flat - b._flat).abs().max() / b._flat.abs().max())
    print("FusedAdam vs torch.optim.Adam after 10 steps: max-norm relative difference", err, "bitwise equal:", torch.equal(a._flat, b._flat))
    assert err <= 1e-6
    for pa, pb in zip(a._param_list(), b._param_list()):
        sta, stb = oa.state[pa], ob.state[pb]
        assert int(sta["step"]) == int(stb["step"]) == 10
        assert float((sta["exp_avg"] - stb["exp_avg"]).abs().max()) <= 1e-6 * float(stb["exp_avg"].abs().max() + 1e-30)
        assert float((sta["exp_avg_sq"] - stb["exp_avg_sq"]).abs().max()) <= 1e-6 * float(stb["exp_avg_sq"].abs().max() + 1e-30)


def test_fused_adam_trains_and_state_dict_round_trip():
    from i2sdf_amd import FusedAdam, I2SDFLoss
    net = _net(3)
    inp = {k: v.cuda() for k, v in camera_inputs(48, (0.0, 0.2, -1.8), W=32, H=32, f=30.0, seed=1).items()}
    gt = {k: v.cuda() for k, v in make_gt(48).items()}
    loss_fn = I2SDFLoss(eikonal_weight=0.1, depth_weight=0.1, normal_weight=0.05)
    opt = FusedAdam(net.get_param_groups(1e-3), eps=1e-15)
    net.force_iters = 1
    g = torch.Generator(device="cuda").manual_seed(0)
    draws = lambda: None
    losses = []
    for it in range(6):
        torch.manual_seed(100)                      # identical draws every step: the loss must go down
        l = loss_fn(net(inp), gt, it)["loss"]
        opt.zero_grad(set_to_none=True)
        l.backward()
        opt.step()
        losses.append(float(l))
    assert losses[-1] < losses[0], losses
    # gradients of the module are views of one flat buffer in parameter order -> the single-launch path was taken
    assert FusedAdam._contiguous_run([p.grad for p in net._param_list()]) is not None
    sd = copy.deepcopy(opt.state_dict())
    opt2 = FusedAdam(net.get_param_groups(1e-3), eps=1e-15)
    opt2.load_state_dict(sd)
    w = net._flat.clone()
    torch.manual_seed(100)
    l = loss_fn(net(inp), gt, 6)["loss"]
    opt.zero_grad(set_to_none=True); l.backward()
    gsave = [p.grad.clone() for p in net._param_list()]
    opt.step()
    w1 = net._flat.clone()
    with torch.no_grad():
        net._flat.copy_(w)
    for p, gg in zip(net._param_list(), gsave):
        p.grad = gg
    opt2.step()                                      # re-adopts the loaded moments into its flat buffers; per-tensor launches (grads are clones)
    assert float((net._flat - w1).abs().max()) <= 1e-7 * float(w1.abs().max())
