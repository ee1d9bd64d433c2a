"""GPU: the C ABI driven by a plain C program (examples/sdf_volume.c: gcc, HIP runtime C API, include/i2sdf.h -- no Python, no torch
on that side).  Its SDF volume must be bit-identical to the Python module's (same library calls underneath)."""
import ctypes as C
import os
import subprocess

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_c_caller_produces_the_same_volume(tmp_path):
    from i2sdf_amd import I2SDFNetwork, synthetic_conf, aligned_axes
    exe = os.path.join(ROOT, "examples", "sdf_volume")
    if not os.path.exists(exe):
        subprocess.run(["bash", os.path.join(ROOT, "examples", "build.sh")], check=True)
    torch.manual_seed(3)
    net = I2SDFNetwork(synthetic_conf(False)).cuda().eval()
    ax = aligned_axes(None, 21, np.array([-0.9, -0.5, -1.2]), np.array([1.0, 0.6, 0.8]))
    want = net.sdf_volume(ax).cpu().numpy()
    eng = net._engine_for(torch.device("cuda", 0))
    desc = eng.layout.net_desc()
    (tmp_path / "desc.bin").write_bytes(C.string_at(C.addressof(desc), C.sizeof(desc)))
    net._flat.detach().cpu().numpy().astype(np.float32).tofile(tmp_path / "params.bin")
    with open(tmp_path / "axes.bin", "wb") as f:
        f.write(np.array(ax.shape_volume, dtype=np.int32).tobytes())
        for a in ax.xyz:
            f.write(np.asarray(a, dtype=np.float32).tobytes())
    r = subprocess.run([exe, str(tmp_path / "desc.bin"), str(tmp_path / "params.bin"), str(tmp_path / "axes.bin"), str(tmp_path / "out.bin")],
                       capture_output=True, text=True, timeout=120)
    assert r.returncode == 0, r.stdout + r.stderr
    got = np.fromfile(tmp_path / "out.bin", dtype=np.float32).reshape(ax.shape_volume)
    assert np.array_equal(got, want), float(np.abs(got - want).max())
    assert "volume" in r.stdout
