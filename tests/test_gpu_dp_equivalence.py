"""GPU: 2-rank data parallelism of the REAL module on one GPU (gloo group, both ranks on cuda:0) in 1-GPU-equivalent mode:
the mean of the two ranks' gradients equals the gradient of ONE process on the concatenated batch:
batch-global sampler convergence OR over the ranks (ray_sampler.py:151), shared randperm columns (ray_sampler.py:223),
count-weighted masked loss means (model/network/__init__.py:320-336), and the flat gradient mean."""
import os
import socket
import subprocess
import sys

import pytest
import torch

pytestmark = pytest.mark.gpu
HERE = os.path.dirname(os.path.abspath(__file__))


def _free_port():
    s = socket.socket(); s.bind(("127.0.0.1", 0)); p = s.getsockname()[1]; s.close()
    return p


def test_two_rank_step_equals_one_rank_step_on_the_concatenated_batch(tmp_path):
    port, Bh = _free_port(), 48
    res_path = str(tmp_path / "dp_result.pt")
    procs = [subprocess.Popen([sys.executable, os.path.join(HERE, "dp_worker.py"), str(r), "2", str(port), str(Bh), res_path],
                              stdout=subprocess.PIPE, stderr=subprocess.STDOUT) for r in range(2)]
    outs = []
    for p in procs:
        try:
            o, _ = p.communicate(timeout=600)
        except subprocess.TimeoutExpired:
            for q in procs:
                q.kill()
            raise
        outs.append(o.decode(errors="replace"))
    assert all(p.returncode == 0 for p in procs), "\n".join(o[-3000:] for o in outs)
    res = torch.load(res_path)
    print(res["iters"], res["solo_iters"], res["ref_iters"], res["xcalls"], max(res["grad_err"].values()))
    # the batch was built so that the ranks would stop at different iteration counts on their own ...
    assert res["solo_iters"][0] != res["solo_iters"][1], "test batch does not exercise the global convergence test"
    # ... and with the device-side OR over the ranks both run exactly as long as the single process on the whole batch
    assert res["iters"][0] == res["iters"][1] == res["ref_iters"] == max(res["solo_iters"])
    assert res["extra_equal"], "rank 0's randperm columns must be used by every rank"
    assert all(c > 0 for c in res["xcalls"]), "the exchange hook was not called"
    assert res["rgb_err"] <= 1e-6, res["rgb_err"]                          # same depths -> same renders
    assert abs(res["loss_dp_mean"] - res["loss_ref"]) <= 1e-6 * abs(res["loss_ref"])
    assert res["image_equal"] and res["image_rows"] == 32 * 24, "2-rank sharded image render must equal the 1-rank image"
    assert res["volume_equal"], "2-rank sharded SDF volume must equal the 1-rank volume"
    # eval on the ATTACHED net (hook installed by the training step): rank-local, no exchange, uneven chunk counts per rank
    assert res["eval_exchange_calls"] == 0, "eval renders / a rank-0-only validation loss entered the data-parallel exchange"
    assert res["attached_image_equal"], "render_image on the attached net (7 chunks over 2 ranks) must equal the 1-rank image"
    assert res["rank0_validation_loss"] == res["rank0_validation_loss"]      # finite (not NaN)
    worst = max(res["grad_err"].items(), key=lambda kv: kv[1])
    assert worst[1] <= 1e-5, f"2-rank mean gradient vs 1-rank gradient on the concatenated batch: {worst}"
