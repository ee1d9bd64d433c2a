"""Time the marching-cubes SDF volume (row N4): i2sdf_sdf_grid (points generated on the device) vs the same grid as a materialised
(n,3) point tensor through sdf_grid, synthetic.yml nets."""
import sys, time, torch
sys.path.insert(0, "tests"); sys.path.insert(0, ".")
from i2sdf_amd import I2SDFNetwork, synthetic_conf, uniform_axes
net = I2SDFNetwork(synthetic_conf(False)).cuda().eval()
for res in (128, 256, 512):
    ax = uniform_axes(res, (-2.0, 2.0))
    net.sdf_volume(uniform_axes(32))          # warm-up
    torch.cuda.synchronize(); t0 = time.time()
    v = net.sdf_volume(ax)
    torch.cuda.synchronize(); dt = time.time() - t0
    line = f"res {res}^3 = {res**3/1e6:.1f} M points: sdf_volume {dt*1e3:.1f} ms ({res**3/dt/1e6:.1f} M points/s, {res**3*0.918e6/dt/1e12:.0f} TFLOP/s fp32-equivalent)"
    if res <= 256:
        import numpy as np
        xx, yy, zz = np.meshgrid(ax.x, ax.y, ax.z)
        torch.cuda.synchronize(); t0 = time.time()
        pts = torch.tensor(np.vstack([xx.ravel(), yy.ravel(), zz.ravel()]).T, dtype=torch.float).cuda()
        z = net.sdf_grid(pts)
        torch.cuda.synchronize(); dt2 = time.time() - t0
        line += f"; host meshgrid + upload + sdf_grid {dt2*1e3:.1f} ms"
        assert torch.equal(z.view(res, res, res).permute(1, 0, 2), v)
    print(line)
