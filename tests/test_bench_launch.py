"""CPU: bench.py's N > 1 launch path without a GPU (`--selftest-launch`: the render core is the CPU stand-in of tests/host_stub.py,
everything else -- self-spawn of one rank per GPU through torch.distributed.run on 127.0.0.1, rendezvous, the data-parallel hook,
barrier-bracketed windows with MAX over ranks, rank 0 printing ONE JSON line -- is the code the driver's multi-GPU run executes).
The driver's own form (`python -m torch.distributed.run ... bench.py --gpus N`) is exercised as well."""
import json
import os
import socket
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _port():
    s = socket.socket(); s.bind(("127.0.0.1", 0)); p = s.getsockname()[1]; s.close(); return p


def _check(stdout, n, equivalent=False):
    lines = [ln for ln in stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, f"exactly one JSON line expected, got {len(lines)}:\n{stdout[-2000:]}"
    d = json.loads(lines[0])
    assert d["n_gpus"] == n and d["config"]["world_size_observed"] == n
    assert d["steps"] == 3 and d["warmup"] == 1 and len(d["windows_ms_per_step"]) == 2
    assert d["scaling"] == "weak" and d["value"] > 0 and d["ms_per_step"] > 0 and d["higher_is_better"] is True
    assert d["strong"]["scaling"] == "strong" and d["strong"]["n_gpus"] == n and d["strong"]["global_rays"] == 8192 and d["strong"]["value"] > 0
    assert d["strong"]["rays_per_gpu"] == 8192 // n
    assert d["data"].startswith("mock"), "a self-test line must say that it is one"
    assert f"dp{n}" in d["config"]["parallelism"]
    if n > 1:
        assert "torch.distributed all_reduce" in d["config"]["parallelism"] and d["config"]["backend"] == "gloo"
        # round 6: one line shows a straggler or a transport fall-back -- the windows as the fastest and the slowest rank saw them, the rank
        # count the library's RCCL communicator reports (None here: gloo carries the collective)
        rk = d["ranks"]
        assert 0 < rk["ms_per_step_fastest_rank"] <= rk["ms_per_step_slowest_rank"] == d["ms_per_step"]
        assert len(rk["windows_ms_per_step_fastest_rank"]) == len(d["windows_ms_per_step"])
        assert rk["rccl_nranks"] is None and rk["torch_world_size"] == n and "torch.distributed" in rk["transport"]
        ar = d["allreduce_us"]                      # the collective of a step timed alone: the record the first N-GPU run will carry
        assert ar["unit"] == "us" and ar["value"] > 0 and ar["bytes"] > 3_000_000 and "torch.distributed" in ar["transport"]
        # round 5: what the collective costs THE STEP -- the same step under no_sync() and the difference, next to the collective alone
        if not equivalent:
            assert d["step_ms_without_allreduce"] > 0
            assert abs(d["exposed_allreduce_ms"] - (d["ms_per_step"] - d["step_ms_without_allreduce"])) < 1e-3
        else:
            assert "step_ms_without_allreduce" not in d      # the sampler's flag exchange keeps running under no_sync(): the pair would not isolate the all-reduce
    else:
        assert "step_ms_without_allreduce" not in d and "exposed_allreduce_ms" not in d and "ranks" not in d
    assert len(d["strong"]["windows_ms_per_step"]) >= 1 and d["strong"]["us_per_ray"] > 0 and d["us_per_ray"] > 0
    return d


def test_self_spawned_two_ranks_print_one_line():
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--selftest-launch", "--gpus", "2", "--backend", "gloo", "--steps", "3",
                        "--warmup", "1", "--windows", "2"], stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=300, cwd=ROOT)
    assert r.returncode == 0, r.stderr.decode()[-3000:]
    _check(r.stdout.decode(), 2)


def test_self_spawned_two_ranks_equivalent_mode():
    """--equivalent: attach_data_parallel(equivalent=True) in a launched job; the CPU stand-in issues the exchanges the device-side hooks of
    a real step issue (5 MAX-flag reductions + 1 denominator average per step, asserted rank-consistent inside bench.py) through the same
    TorchExchange object and group -- until round 5 these had only ever run inside pytest workers"""
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--selftest-launch", "--gpus", "2", "--backend", "gloo", "--steps", "3",
                        "--warmup", "1", "--windows", "2", "--equivalent"], stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=300, cwd=ROOT)
    assert r.returncode == 0, r.stderr.decode()[-3000:]
    d = _check(r.stdout.decode(), 2, equivalent=True)
    assert d["config"]["equivalent"] is True and d["equivalent"]["on"] is True
    assert d["equivalent"]["exchange_calls"] > 0 and d["equivalent"]["exchange_calls"] % 6 == 0      # six exchanges per step


def test_self_spawned_eight_ranks_print_one_line():
    """the real world size of the driver's scaling run: rendezvous, port logic and the rank-0 line with 8 ranks on 127.0.0.1"""
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--selftest-launch", "--gpus", "8", "--backend", "gloo", "--steps", "3",
                        "--warmup", "1", "--windows", "2"], stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=600, cwd=ROOT,
                       env=dict(os.environ, OMP_NUM_THREADS="1"))
    assert r.returncode == 0, r.stderr.decode()[-3000:]
    d = _check(r.stdout.decode(), 8)
    assert d["strong"]["rays_per_gpu"] == 1024


def test_driver_form_torchrun_two_ranks():
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1", "--master-port", str(_port()),
           os.path.join(ROOT, "bench.py"), "--selftest-launch", "--gpus", "2", "--backend", "gloo", "--steps", "3", "--warmup", "1", "--windows", "2"]
    r = subprocess.run(cmd, stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=300, cwd=ROOT)
    assert r.returncode == 0, r.stderr.decode()[-3000:]
    _check(r.stdout.decode(), 2)


def test_single_rank():
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--selftest-launch", "--steps", "3", "--warmup", "1", "--windows", "2"],
                       stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=300, cwd=ROOT)
    assert r.returncode == 0, r.stderr.decode()[-3000:]
    _check(r.stdout.decode(), 1)


def test_nccl_backend_without_gpus_fails_fast_not_hangs():
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "1", "--warmup", "1"],
                       stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=300, cwd=ROOT)
    assert r.returncode != 0
    assert not [ln for ln in r.stdout.decode().splitlines() if ln.startswith("{")], "no record may be printed by a run that could not start"
