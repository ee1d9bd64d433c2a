"""GPU: bitwise run-to-run reproducibility of the MLP and weight-gradient kernels.

Everything in the library is deterministic by construction (no atomics; split-M partials are reduced in a fixed order), so a
mismatch between two runs on the same inputs means a synchronisation bug.  This test exists because one was found this way:
hipcc does not reliably drain the LDS-DMA `vmcnt` in front of `s_barrier`, and the weight-gradient kernel read DMA pieces
that had not landed about once in four launches (a repetition stress test, round 1)."""
import pytest
import torch

from oracle import i2sdf_oracle as orc
from test_gpu_train_forward import make_engine

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("bf16x3,parts,B", [(True, 0, 360), (True, 4, 360), (False, 0, 360), (True, 2, 1024)])
def test_kernels_are_bitwise_reproducible(bf16x3, parts, B):
    """parts = 0: full rounds + split-K tail workgroups (on a side stream); parts = 4: point ranges on four streams (I2SDF_OPT_PARTS).
    B = 1024, parts = 2: the headline step's shapes (102 400 points in two ranges on two streams) -- the memory system as loaded as it gets,
    which is when a stage hand-over that counted its waits wrongly (x3.h: the counted stage wait, round 5) would read a stage too early."""
    from i2sdf_amd.config import synthetic_conf
    ocfg = orc.synthetic_cfg(False)
    sd = orc.perturb_params(orc.init_params(ocfg, seed=13), 0.05, seed=14)
    conf = dict(synthetic_conf(False))
    conf["bf16x3"] = bf16x3
    eng = make_engine(conf, sd)
    eng.set_parts(parts)
    flat = eng.layout.flat_from_state_dict(sd).cuda()
    g = torch.Generator().manual_seed(6)
    n = 97                                       # B = 360: 36 000 points: 256 full workgroups + 101 split-K tail workgroups, 36 weight-gradient chunks
    M = B * n + 3 * B
    x = ((torch.rand(M, 3, generator=g) * 2 - 1) * 1.5).cuda()
    dirs = torch.nn.functional.normalize(torch.randn(B, 3, generator=g), dim=1).cuda()
    cw = torch.randn(B * n, 3, generator=g).cuda()
    nb, sb = torch.randn(M, 3, generator=g).cuda(), torch.randn(M, generator=g).cuda()
    ref = None
    for rep in range(12 if B == 360 else 8):
        fwd = eng.sdf_forward_grad(points=x)
        rgb_h, rs, pev = eng.rgb_forward(dirs, n, fwd["feat"], B * n)
        gar, ga_last, fbar = eng.rgb_backward(rgb_h, cw, rs, B * n)
        bw = eng.sdf_backward(fwd, sbar=sb, fbar=fbar, m_fbar=B * n, nbar=nb)
        gflat = torch.zeros_like(flat)
        eng.weight_grads(flat, gflat, fwd, bw, M_main=B * n, fbar=fbar, rgb_fw={"pev": pev, "rs": rs}, rgb_bw={"gar": gar, "ga_last": ga_last})
        # saved tensors in point-major form (padding points of a blocked tile hold whatever their lanes computed: compare real points only)
        pm = lambda name, t_, m_: eng.saved_pm(name, t_, m_)       # (decodes the packed 24-bit records of abars / gus / gas when I2SDF_OPT_SAVES24 is in effect)
        if parts == 0:
            assert B != 360 or eng.blocked_points(0, M, fwd["Mp"]) in (0, 256 * 128), "the batch must exercise bulk + split-K tail"
        else:
            assert eng.blocked_points(0, M, fwd["Mp"]) == fwd["Mp"], "point ranges: no tail, every saved row blocked"
        cur = {"sdf": fwd["sdf"], "feat": fwd["feat"][:M], "grad": fwd["grad"], "hs": pm("hs", fwd["hs"], M), "abars": pm("abars", fwd["abars"], M),
               "rgb": rgb_h, "rs": pm("rs", rs, B * n), "gar": pm("gar", gar, B * n), "fbar": fbar[:B * n], "gus": pm("gus", bw["gus"], M)[1:],
               "gas": pm("gas", bw["gas"], M), "sdf_only": eng.sdf_forward(x), "param_grads": gflat}
        if ref is None:
            ref = {k: v.clone() for k, v in cur.items()}
            continue
        for k in cur:
            assert torch.equal(cur[k], ref[k]), f"{k} differs between run 0 and run {rep} ({int((cur[k] != ref[k]).sum())} entries)"


@pytest.mark.parametrize("bf16x3", [True, False])
def test_tail_overlap_changes_nothing(bf16x3):
    """I2SDF_OPT_TAIL_OVERLAP: the split-K tail workgroups of the SDF training kernels on the plan's side stream, concurrent
    with the full workgroups (fork/join with events).  Same kernels on the same points, so every output must be bitwise equal
    to the single-stream run -- a difference would mean a missing dependency between the two streams."""
    from i2sdf_amd.config import synthetic_conf
    ocfg = orc.synthetic_cfg(False)
    sd = orc.perturb_params(orc.init_params(ocfg, seed=21), 0.05, seed=22)
    conf = dict(synthetic_conf(False))
    conf["bf16x3"] = bf16x3
    eng = make_engine(conf, sd)
    eng.set_parts(0)                                 # the split-K tail only exists without point ranges
    g = torch.Generator().manual_seed(8)
    B, n = 360, 97
    M = B * n + 3 * B                              # 36 000 points: 256 full workgroups + a 3 232-point split-K tail (101 workgroups)
    x = ((torch.rand(M, 3, generator=g) * 2 - 1) * 1.5).cuda()
    nb, sb = torch.randn(M, 3, generator=g).cuda(), torch.randn(M, generator=g).cuda()
    fb = torch.randn(B * n, 256, generator=g).cuda()

    def run():
        fwd = eng.sdf_forward_grad(points=x)
        bw = eng.sdf_backward(fwd, sbar=sb, fbar=fb, m_fbar=B * n, nbar=nb)
        # consumers on the caller's stream right behind the entry points: they must see the tail's results
        pm = lambda name, t_: eng.saved_pm(name, t_, M)       # real points only (padding points of a blocked tile are never written)
        return {"sdf": fwd["sdf"].clone(), "feat": fwd["feat"][:M].clone(), "grad": fwd["grad"].clone(), "hs": pm("hs", fwd["hs"]),
                "abars": pm("abars", fwd["abars"]), "gus": pm("gus", bw["gus"])[1:], "gas": pm("gas", bw["gas"])}

    eng.set_tail_overlap(False)
    ref = run()
    eng.set_tail_overlap(True)
    for rep in range(8):
        cur = run()
        for k in cur:
            assert torch.equal(cur[k], ref[k]), f"{k}: overlap run {rep} differs from the single-stream run ({int((cur[k] != ref[k]).sum())} entries)"


def test_point_ranges_change_nothing():
    """I2SDF_OPT_PARTS: the per-point entry points cut into n point ranges, each on its own stream; inside a chain
    (i2sdf_chain_begin / _end) the ranges stay un-joined across entry points.  The kernels and the points they own are the same
    for every n >= 2, joined or not, so every output -- the parameter gradients included (fixed-order reduction of the per-chunk
    partials) -- must be bitwise equal; a difference would mean a missing dependency between the streams."""
    from i2sdf_amd.config import synthetic_conf
    ocfg = orc.synthetic_cfg(False)
    sd = orc.perturb_params(orc.init_params(ocfg, seed=31), 0.05, seed=32)
    eng = make_engine(dict(synthetic_conf(False)), sd)
    flat = eng.layout.flat_from_state_dict(sd).cuda()
    g = torch.Generator().manual_seed(9)
    B, n = 300, 97                                   # 30 000 points = 29.3 chunks of 1024: ragged last chunk, ragged last workgroup
    M = B * n + 3 * B
    x = ((torch.rand(M, 3, generator=g) * 2 - 1) * 1.5).cuda()
    dirs = torch.nn.functional.normalize(torch.randn(B, 3, generator=g), dim=1).cuda()
    cw = torch.randn(B * n, 3, generator=g).cuda()
    nb, sb = torch.randn(M, 3, generator=g).cuda(), torch.randn(M, generator=g).cuda()

    def run(chain):
        import contextlib
        with (eng.chain(M) if chain else contextlib.nullcontext()):
            fwd = eng.sdf_forward_grad(points=x)
            rgb_h, rs, pev = eng.rgb_forward(dirs, n, fwd["feat"], B * n)
        # consumers on the caller's stream right behind the chain: they must see every range's results
        out = {"sdf": fwd["sdf"].clone(), "feat": fwd["feat"][:M].clone(), "grad": fwd["grad"].clone(), "rgb": rgb_h.clone()}
        gflat = torch.zeros_like(flat)
        with (eng.chain(M) if chain else contextlib.nullcontext()):
            gar, ga_last, fbar = eng.rgb_backward(rgb_h, cw, rs, B * n)
            bw = eng.sdf_backward(fwd, sbar=sb, fbar=fbar, m_fbar=B * n, nbar=nb)
            eng.weight_grads(flat, gflat, fwd, bw, M_main=B * n, fbar=fbar, rgb_fw={"pev": pev, "rs": rs}, rgb_bw={"gar": gar, "ga_last": ga_last})
        pm = lambda name, t_, m_: eng.saved_pm(name, t_, m_)
        out.update({"hs": pm("hs", fwd["hs"], M), "abars": pm("abars", fwd["abars"], M), "rs": pm("rs", rs, B * n), "gar": pm("gar", gar, B * n),
                    "fbar": fbar[:B * n].clone(), "gus": pm("gus", bw["gus"], M)[1:], "gas": pm("gas", bw["gas"], M), "param_grads": gflat})
        return out

    eng.set_parts(2)
    ref = run(False)
    for parts, chain in ((2, True), (3, False), (3, True), (4, False), (4, True), (4, True)):
        eng.set_parts(parts)
        cur = run(chain)
        for k in cur:
            assert torch.equal(cur[k], ref[k]), f"{k}: parts={parts} chain={chain} differs from parts=2 ({int((cur[k] != ref[k]).sum())} entries)"


def test_chain_protocol_errors():
    """i2sdf_chain_begin / _end misuse is refused, not silently mis-ordered: chains do not nest, the range count cannot change inside a
    chain, end without begin is a no-op, and a batch with fewer chunks than ranges simply runs as one launch."""
    from i2sdf_amd.config import synthetic_conf
    from i2sdf_amd import lib as L
    ocfg = orc.synthetic_cfg(False)
    sd = orc.perturb_params(orc.init_params(ocfg, seed=41), 0.05, seed=42)
    eng = make_engine(dict(synthetic_conf(False)), sd, parts=2)
    h, plan, st = L.load(), eng._plan, L.stream_ptr()
    assert h.i2sdf_chain_end(plan, st) == 0                                   # nothing open: no-op
    assert h.i2sdf_chain_begin(plan, 50000, st) == 0
    assert h.i2sdf_chain_begin(plan, 50000, st) == -1                         # no nesting
    assert h.i2sdf_plan_set_option(plan, L.OPT_PARTS, 3) == -1                # not inside a chain
    assert h.i2sdf_plan_set_option(plan, L.OPT_SAVES24, 0) == -1              # nor the storage format of the tensors in flight
    assert h.i2sdf_chain_fence(plan, st) == 0
    assert h.i2sdf_chain_end(plan, st) == 0
    assert h.i2sdf_plan_set_option(plan, L.OPT_PARTS, 3) == 0 and h.i2sdf_plan_set_option(plan, L.OPT_PARTS, 2) == 0
    # a batch smaller than one chunk per range: one launch, same values as without ranges on the points of the full workgroups
    x = ((torch.rand(1500, 3, generator=torch.Generator().manual_seed(3)) * 2 - 1) * 1.5).cuda()
    with eng.chain(1500):
        a = eng.sdf_forward_grad(points=x)
    eng.set_parts(0)
    b = eng.sdf_forward_grad(points=x)
    assert torch.equal(a["sdf"], b["sdf"])
    if eng.saves24:
        # with packed 24-bit records (I2SDF_OPT_SAVES24) the ranged run takes the d sdf/dx kernel's packing instantiation, the un-ranged one (fp32 storage:
        # the option needs the ranges) the plain one: the same arithmetic, compiled separately -- equal to rounding, not bit for bit
        assert float((a["grad"] - b["grad"]).abs().max()) <= 2e-6 * float(b["grad"].abs().max())
    else:
        assert torch.equal(a["grad"], b["grad"])
    torch.cuda.synchronize()


def test_chain_with_an_entry_point_off_the_ranged_path():
    """Point ranges with the kernel families mixed (round-3 review): the radiance net on the fp32 path (rgb_bf16x3 off) does not cut its
    batch into ranges, so inside a chain it runs whole-batch on the caller's stream between two ranged entry points.  The library joins
    every range in front of it and fences the side streams behind it (plan.h: ChainGuard); the engine itself no longer turns point ranges
    on in that state.  Chained and un-chained runs must agree bit for bit -- a missing dependency shows up as a difference (or as garbage)."""
    import contextlib
    from i2sdf_amd.config import synthetic_conf
    from i2sdf_amd import lib as L
    ocfg = orc.synthetic_cfg(False)
    sd = orc.perturb_params(orc.init_params(ocfg, seed=61), 0.05, seed=62)
    eng = make_engine(dict(synthetic_conf(False)), sd)
    flat = eng.layout.flat_from_state_dict(sd).cuda()
    eng.set_rgb_bf16x3(False)
    assert eng.parts == 0                                 # the engine drops the ranges when a family leaves the ranged path ...
    L.check(eng._lib.i2sdf_plan_set_option(eng._plan, L.OPT_PARTS, 2), "i2sdf_plan_set_option")      # ... a C caller may still ask for them
    eng.parts = 2
    g = torch.Generator().manual_seed(29)
    B, n = 300, 97
    M = B * n + 3 * B
    x = ((torch.rand(M, 3, generator=g) * 2 - 1) * 1.5).cuda()
    dirs = torch.nn.functional.normalize(torch.randn(B, 3, generator=g), dim=1).cuda()
    cw = torch.randn(B * n, 3, generator=g).cuda()
    nb, sb = torch.randn(M, 3, generator=g).cuda(), torch.randn(M, generator=g).cuda()

    def run(chain):
        with (eng.chain(M) if chain else contextlib.nullcontext()):
            fwd = eng.sdf_forward_grad(points=x)
            rgb_h, rs, pev = eng.rgb_forward(dirs, n, fwd["feat"], B * n)
        out = {"sdf": fwd["sdf"].clone(), "feat": fwd["feat"][:M].clone(), "grad": fwd["grad"].clone(), "rgb": rgb_h.clone()}
        gflat = torch.zeros_like(flat)
        with (eng.chain(M) if chain else contextlib.nullcontext()):
            gar, ga_last, fbar = eng.rgb_backward(rgb_h, cw, rs, B * n)
            bw = eng.sdf_backward(fwd, sbar=sb, fbar=fbar, m_fbar=B * n, nbar=nb)
            eng.weight_grads(flat, gflat, fwd, bw, M_main=B * n, fbar=fbar, rgb_fw={"pev": pev, "rs": rs}, rgb_bw={"gar": gar, "ga_last": ga_last})
        pm = lambda name, t_, m_: eng.saved_pm(name, t_, m_)      # (padding rows are never written)
        out.update({"fbar": fbar[:B * n].clone(), "gus": pm("gus", bw["gus"], M)[1:], "gas": pm("gas", bw["gas"], M), "param_grads": gflat})
        return out

    ref = run(False)
    for rep in range(3):
        cur = run(True)
        for k in cur:
            assert torch.isfinite(cur[k]).all(), k
            assert torch.equal(cur[k], ref[k]), f"{k}: chained run {rep} differs from the un-chained one ({int((cur[k] != ref[k]).sum())} entries)"
