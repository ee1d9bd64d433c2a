"""GPU: i2sdf_training_draws -- all random draws of a training forward in one launch (csrc/draws.hip).  The stream is the
library's own (Philox keyed by a seed from torch's CPU generator), so there is nothing to compare bit for bit: the tests check
the distributions the reference's torch calls have (ray_sampler.py:60-66,176-177,223,234; model/network/__init__.py:177,184),
reproducibility, and that the module uses it."""
import pytest
import torch

from test_gpu_edge_cases import _net
from helpers import camera_inputs

pytestmark = pytest.mark.gpu


def _engine():
    net, _, _ = _net(True)
    return net, net._engine_for(torch.device("cuda", 0))


def _chi2(counts, expected):
    return float(((counts.double() - expected) ** 2 / expected).sum())


def test_uniform_draws_have_the_right_distribution():
    net, eng = _engine()
    sc = eng.cfg.sampler
    B = 8192
    d = eng.training_draws(B, 12345, torch.device("cuda", 0), eik_radius=3.0)
    assert d["strat_u"].shape == (B, sc.N_samples_eval) and d["cdf_u"].shape == (B, sc.N_samples)
    assert d["eik_pts"].shape == (B, 3) and d["nbr_off"].shape == (B, 3) and d["eik_idx"].shape == (B,)
    for name, lo, hi in (("strat_u", 0.0, 1.0), ("cdf_u", 0.0, 1.0), ("eik_pts", -3.0, 3.0), ("nbr_off", -0.005, 0.005)):
        x = d[name].double().reshape(-1)
        n = x.numel()
        assert float(x.min()) >= lo and float(x.max()) < hi, name
        u = (x - lo) / (hi - lo)
        se = (1 / 12) ** 0.5 / n ** 0.5
        assert abs(float(u.mean()) - 0.5) < 5 * se, (name, float(u.mean()))
        assert abs(float(u.var()) - 1 / 12) < 0.01 / 12 + 5 / n ** 0.5 / 12, (name, float(u.var()))
        bins = torch.histc(u.float(), 64, 0, 1)
        assert _chi2(bins, n / 64) < 63 + 6 * (2 * 63) ** 0.5, (name, _chi2(bins, n / 64))        # chi2(63): mean 63, sd 11.2
    # neighbouring elements and the different streams are uncorrelated
    a, b = d["strat_u"].double().reshape(-1), d["cdf_u"].double().reshape(-1)
    m = min(a.numel(), b.numel())
    for x, y in ((a[:-1], a[1:]), (a[:m], b[:m]), (a[:-4], a[4:])):
        c = float(torch.corrcoef(torch.stack([x, y]))[0, 1])
        assert abs(c) < 5 / x.numel() ** 0.5, c
    idx = d["eik_idx"]
    assert idx.dtype == torch.int32 and int(idx.min()) >= 0 and int(idx.max()) < eng.n_z
    cnt = torch.bincount(idx.long(), minlength=eng.n_z)
    assert _chi2(cnt, B / eng.n_z) < (eng.n_z - 1) + 6 * (2 * (eng.n_z - 1)) ** 0.5


def test_extra_columns_are_uniform_k_subsets():
    """Row `it` = randperm(N_eval*(it+1))[:N_extra]: distinct, in range, every column equally likely, every position too."""
    net, eng = _engine()
    sc = eng.cfg.sampler
    k, n0 = sc.N_samples_extra, sc.N_samples_eval
    assert k > 0
    reps = 1500
    rows = torch.stack([eng.training_draws(4, 1000 + r, torch.device("cuda", 0))["extra_idx"] for r in range(reps)])   # (reps, iters, k)
    assert rows.shape == (reps, sc.max_total_iters, k) and rows.dtype == torch.int32
    for it in range(sc.max_total_iters):
        n = n0 * (it + 1)
        r = rows[:, it].long()
        assert int(r.min()) >= 0 and int(r.max()) < n
        assert all(len(set(x.tolist())) == k for x in r[:50])                                   # distinct within a row
        cnt = torch.bincount(r.reshape(-1), minlength=n)
        assert _chi2(cnt, reps * k / n) < (n - 1) + 6 * (2 * (n - 1)) ** 0.5, it
    first = torch.bincount(rows[:, 0, 0].long(), minlength=n0)                                   # the first position alone is uniform too
    assert _chi2(first, reps / n0) < (n0 - 1) + 6 * (2 * (n0 - 1)) ** 0.5


def test_draws_are_a_function_of_the_seed_and_optional():
    net, eng = _engine()
    dev = torch.device("cuda", 0)
    a, b, c = eng.training_draws(100, 7, dev, 2.0), eng.training_draws(100, 7, dev, 2.0), eng.training_draws(100, 8, dev, 2.0)
    for k in a:
        assert torch.equal(a[k], b[k]), k
        assert not torch.equal(a[k], c[k]), k
    big = eng.training_draws(333, 7, dev, 2.0)                     # counter = element index: a longer batch extends the same stream
    assert torch.equal(big["strat_u"].reshape(-1)[: a["strat_u"].numel()], a["strat_u"].reshape(-1))
    d = eng.training_draws(5, 1, dev, 2.0, want_eik=False)
    assert d["eik_pts"] is None and d["nbr_off"] is None and d["strat_u"].shape[0] == 5
    assert eng.training_draws(0, 1, dev)["strat_u"].shape[0] == 0


def test_module_forward_is_reproducible_under_manual_seed_and_uses_one_launch():
    net, ocfg, sd = _net(True)
    inp = {k: v.cuda() for k, v in camera_inputs(64, (0.0, 0.2, -1.8), W=32, H=32, f=30.0, seed=1).items()}
    outs = []
    for seed in (5, 5, 6):
        torch.manual_seed(seed)
        outs.append(net(inp))
    for k in ("rgb_values", "depth_values", "grad_theta", "diff_norm"):
        assert torch.equal(outs[0][k], outs[1][k]), k
    assert not torch.equal(outs[0]["rgb_values"], outs[2]["rgb_values"])
    assert net.last_extra_idx.dtype == torch.int32 and tuple(net.last_extra_idx.shape) == (ocfg.sampler.max_total_iters, ocfg.sampler.N_samples_extra)
    # the separate-torch-ops path still works and is statistically the same thing
    net.fused_draws = False
    torch.manual_seed(5)
    alt = net(inp)
    assert alt["rgb_values"].shape == outs[0]["rgb_values"].shape and torch.isfinite(alt["rgb_values"]).all()
    with torch.profiler.profile(activities=[torch.profiler.ProfilerActivity.CUDA]) as prof:
        net.fused_draws = True
        net(inp)
        torch.cuda.synchronize()
    names = [e.key for e in prof.key_averages()]
    assert not any("distribution" in n or "topk" in n.lower() or "TopK" in n for n in names), names
