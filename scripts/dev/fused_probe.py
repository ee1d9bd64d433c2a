import torch, time
from i2sdf_amd import I2SDFNetwork, synthetic_conf
net = I2SDFNetwork(synthetic_conf()).cuda().train()
net._engine_for("cuda:0")
ps = list(net.parameters())
for fused in (False, True):
    opt = torch.optim.Adam(net.get_param_groups(1e-3), eps=1e-15, fused=fused)
    for p in ps: p.grad = torch.ones_like(p)
    v0 = sum(p._version for p in ps); f0 = net._flat.clone()
    opt.step(); torch.cuda.synchronize()
    print("fused", fused, "version delta", sum(p._version for p in ps) - v0, "flat changed", float((net._flat - f0).abs().max()),
          "views intact", all(p.data_ptr() == net._flat.data_ptr() + 4 * off for (n, off, sh), p in zip(net.layout.entries, net._param_list())))
