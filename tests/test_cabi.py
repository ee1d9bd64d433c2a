"""CPU: the C-ABI shared library loads, exports every symbol include/i2sdf.h declares, and its host-side logic
(plan construction, argument validation, error codes) behaves -- no compute call needs a GPU here."""
import ctypes as C
import os
import re
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="session")
def lib():
    from i2sdf_amd import lib as L
    if not os.path.exists(L.LIB_PATH):
        subprocess.run([sys.executable, "-c", "import __graft_entry__ as g; g.build()"], cwd=ROOT, check=True)
    return L


def header_symbols():
    text = open(os.path.join(ROOT, "include", "i2sdf.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(i2sdf_[a-z0-9_]+)\s*\(", text)))


def test_every_declared_symbol_is_exported_and_bound(lib):
    syms = header_symbols()
    assert len(syms) >= 20
    raw = C.CDLL(lib.LIB_PATH)
    for s in syms:
        assert hasattr(raw, s), f"{s} declared in include/i2sdf.h but not exported"
        assert s in lib.SIGNATURES, f"{s} has no ctypes signature in i2sdf_amd/lib.py"
    for s in lib.SIGNATURES:
        assert s in syms, f"{s} bound in lib.py but not declared in the header"


def test_version_and_error_strings(lib):
    h = lib.load()
    assert h.i2sdf_version() == 100
    assert h.i2sdf_strerror(0) == b"ok"
    assert b"invalid" in h.i2sdf_strerror(-1)
    assert h.i2sdf_wgrad_chunk_points() == 2048
    assert h.i2sdf_sampler_workspace_floats(10) > 10 * 640 * 6


def _plan(lib, conf):
    from i2sdf_amd.config import NetConfig
    from i2sdf_amd.params import ParamLayout
    lay = ParamLayout(NetConfig.from_conf(conf))
    desc = lay.net_desc()
    plan = C.c_void_p()
    rc = lib.load().i2sdf_plan_create(C.byref(desc), C.byref(plan))
    return rc, plan, lay, desc


def test_plan_layout_sizes(lib):
    from i2sdf_amd.config import synthetic_conf, plumbing_conf
    h = lib.load()
    for conf, n_params in ((synthetic_conf(), 800955), (synthetic_conf(True), 635965), (plumbing_conf(), None), (plumbing_conf(True, True), None)):
        rc, plan, lay, _ = _plan(lib, conf)
        assert rc == 0
        if n_params:
            assert lay.n_params == n_params          # SURVEY.md appendix B [probed on the reference]
        pack = h.i2sdf_plan_pack_floats(plan)
        # fp32 forward + transposed streams hold every weight about twice; the bf16x3 forward / reverse streams of the SDF net
        # add 1.5x each, and so do their 16-point-wave twins (256-wide nets); plus stage padding
        assert 2 * lay.n_params * 0.9 < pack < 8 * lay.n_params * 1.2 + 128 * 8192
        assert h.i2sdf_plan_wgrad_floats(plan) >= lay.n_params - 1
        h.i2sdf_plan_destroy(plan)


def test_unsupported_shapes_are_rejected(lib):
    from i2sdf_amd.config import synthetic_conf
    conf = synthetic_conf()
    rc, plan, lay, desc = _plan(lib, conf)
    assert rc == 0
    lib.load().i2sdf_plan_destroy(plan)
    bad = type(desc)()
    C.memmove(C.byref(bad), C.byref(desc), C.sizeof(desc))
    bad.sdf.hidden = 250                                  # not a multiple of 32
    p2 = C.c_void_p()
    assert lib.load().i2sdf_plan_create(C.byref(bad), C.byref(p2)) == -1
    C.memmove(C.byref(bad), C.byref(desc), C.sizeof(desc))
    bad.sdf.skip_layer = 0
    assert lib.load().i2sdf_plan_create(C.byref(bad), C.byref(p2)) == -1
    assert lib.load().i2sdf_plan_create(None, C.byref(p2)) == -1


def test_entry_points_validate_arguments_without_a_gpu(lib):
    from i2sdf_amd.config import plumbing_conf
    h = lib.load()
    rc, plan, lay, _ = _plan(lib, plumbing_conf())
    assert rc == 0
    P = C.c_void_p(4096)
    assert h.i2sdf_sdf_forward(plan, P, None, 10, P, None, 0, None) == -1                    # no points
    assert h.i2sdf_sdf_forward(plan, P, P, 0, P, None, 0, None) == 0                         # empty batch is a no-op
    assert h.i2sdf_sdf_forward_grad(plan, P, P, None, None, None, 0, 1, 0, 300, 300, P, P, P, P, P, P, None) == -1   # Mp not x128
    assert h.i2sdf_sdf_forward_grad(plan, P, None, None, None, None, 0, 1, 0, 300, 384, P, P, P, P, P, P, None) == -1  # no source
    assert h.i2sdf_rgb_forward(plan, P, P, 0, P, 10, 128, P, P, P, None) == -1               # n_per_ray <= 0
    assert h.i2sdf_composite_forward(P, 1e-4, P, 10, P, P, None, None, P, 4, 300, P, P, P, None, None, None, None, None) == -1  # n > 256
    assert h.i2sdf_composite_forward(P, 1e-4, P, 10, P, P, None, None, P, 4, 9, P, P, P, P, None, None, None, None) == -1     # normal without grad
    assert h.i2sdf_light_forward(plan, P, P, 10, 128, P, P, None) == -1                       # this net has no light head
    assert h.i2sdf_ray_setup(P, P, P, 4, -1, P, P, P, None) == -1
    assert h.i2sdf_ray_setup(P, None, P, 4, 2, P, P, P, None) == -1                           # no pose
    assert h.i2sdf_ray_setup_ex(P, P, 1, P, 0, 2, P, P, P, None) == 0                         # empty batch is a no-op
    tab, out = lib.RayTables(), lib.RayBatch()
    assert h.i2sdf_ray_batch(C.byref(tab), P, 0, C.byref(out), None) == 0                     # empty batch
    assert h.i2sdf_ray_batch(C.byref(tab), P, 8, C.byref(out), None) == -1                    # no camera tables
    assert h.i2sdf_ray_batch(None, P, 8, C.byref(out), None) == -1
    assert h.i2sdf_sphere_intersections(P, P, 8, 3.0, P, None, None) == -1                    # no miss counter
    h.i2sdf_plan_destroy(plan)


def test_per_point_workspaces_need_whole_workgroups_of_rows(lib):
    """The per-point workspaces (hs, abars, gus, gas, rs, gar, feat ...) have Mp rows, Mp = M rounded up to 128: the kernels run whole
    128-point workgroups and since round 5 WRITE the padding rows M..Mp of the saved tensors unconditionally (include/i2sdf.h).  A caller
    that sized a workspace to exactly M rows would be written past its end -- every per-point entry point refuses such an Mp before it
    launches anything (validation only: no call below reaches a launch)."""
    from i2sdf_amd.config import synthetic_conf
    h = lib.load()
    rc, plan, _, _ = _plan(lib, synthetic_conf(True))
    assert rc == 0
    P = C.c_void_p(4096)
    for M in (1, 127, 129, 51201):
        good = (M + 127) // 128 * 128
        for Mp in sorted({M, good - 1, good - 128, good + 64} - {good}):
            if Mp == M and M % 128 == 0:
                continue
            assert h.i2sdf_sdf_forward_grad(plan, P, P, None, None, None, 0, 1, 0, M, Mp, P, P, P, P, P, P, None) == -1, (M, Mp)
            assert h.i2sdf_sdf_backward(plan, P, P, None, None, None, 0, 1, 0, M, Mp, P, P, P, P, M, P, P, P, P, P, P, None) == -1, (M, Mp)
            assert h.i2sdf_rgb_forward(plan, P, P, 1, P, M, Mp, P, P, P, None) == -1, (M, Mp)
            assert h.i2sdf_rgb_backward(plan, P, P, P, P, M, Mp, P, P, P, None) == -1, (M, Mp)
            assert h.i2sdf_light_forward(plan, P, P, M, Mp, P, P, None) == -1, (M, Mp)
            assert h.i2sdf_light_backward(plan, P, P, P, P, M, Mp, P, P, None) == -1, (M, Mp)
            tb = lib.TrainBuffers()
            tb.M_sdf, tb.M_main, tb.Mp = M, 0, Mp
            assert h.i2sdf_weight_grads(plan, C.byref(tb), P, P, 1, P, None) == -1, (M, Mp)
    h.i2sdf_plan_destroy(plan)


def test_plan_options(lib):
    """i2sdf_plan_set_option: the bf16x3 twins exist for 256-wide nets only; unknown options are rejected."""
    from i2sdf_amd.config import synthetic_conf, plumbing_conf
    h = lib.load()
    rc, plan, _, _ = _plan(lib, synthetic_conf())
    assert rc == 0
    for opt in (lib.OPT_SDF_FWD_BF16X3, lib.OPT_WGRAD_BF16X3, lib.OPT_TRAIN_FWD_BF16X3, lib.OPT_SDF_BWD_BF16X3, lib.OPT_RGB_BF16X3,
                lib.OPT_TAIL_OVERLAP):      # (the side stream of the tail overlap is only created by the first GPU launch)
        assert h.i2sdf_plan_set_option(plan, opt, 1) == 0 and h.i2sdf_plan_set_option(plan, opt, 0) == 0
    assert h.i2sdf_plan_set_option(plan, 12345, 1) == -1
    assert h.i2sdf_plan_set_option(None, lib.OPT_WGRAD_BF16X3, 1) == -1
    h.i2sdf_plan_destroy(plan)
    rc, plan, _, _ = _plan(lib, plumbing_conf())                 # 64-wide nets: sampler forward only
    assert rc == 0
    assert h.i2sdf_plan_set_option(plan, lib.OPT_SDF_FWD_BF16X3, 1) == 0
    assert h.i2sdf_plan_set_option(plan, lib.OPT_TRAIN_FWD_BF16X3, 1) == -1
    assert h.i2sdf_plan_set_option(plan, lib.OPT_SDF_BWD_BF16X3, 1) == -1
    assert h.i2sdf_plan_set_option(plan, lib.OPT_RGB_BF16X3, 1) == -1
    h.i2sdf_plan_destroy(plan)


def test_fresh_plan_defaults(lib):
    """Round 6: a fresh plan of a 256-wide configuration runs the tested bf16x3 twins with blocked saved tensors (what a C caller gets without
    setting anything), the narrower-than-fp32 options and the point ranges stay opt-in; 64-wide plans have nothing on."""
    from i2sdf_amd.config import synthetic_conf, plumbing_conf
    h = lib.load()
    M = 1024 * 100
    Mp = (M + 127) // 128 * 128
    bulk = (Mp // 128 // 256) * 256 * 128
    rc, plan, _, _ = _plan(lib, synthetic_conf())
    assert rc == 0
    assert h.i2sdf_blocked_points(plan, 0, M, Mp, 1) == bulk and h.i2sdf_blocked_points(plan, 1, M, Mp, 1) == bulk      # blocked + bf16x3 sweeps on, parts off
    assert h.i2sdf_plan_set_option(plan, lib.OPT_SDF_BWD_BF16X3, 0) == 0
    assert h.i2sdf_blocked_points(plan, 0, M, Mp, 1) == 0 and h.i2sdf_blocked_points(plan, 1, M, Mp, 1) == bulk         # (blocked needs the bf16x3 family)
    h.i2sdf_plan_destroy(plan)
    rc, plan, _, _ = _plan(lib, plumbing_conf())
    assert rc == 0 and h.i2sdf_blocked_points(plan, 0, M, Mp, 1) == 0 and h.i2sdf_blocked_points(plan, 1, M, Mp, 1) == 0
    h.i2sdf_plan_destroy(plan)


def test_new_plan_options_and_blocked_prefix(lib):
    """Round-2 options toggle on 256-wide nets; i2sdf_blocked_points follows the split of a launch into full rounds of 128-point
    workgroups (one per CU, 256 without a device) and the split-K tail (csrc/mlp_common.h: split_bulk_points)."""
    from i2sdf_amd.config import synthetic_conf, plumbing_conf
    h = lib.load()
    rc, plan, _, _ = _plan(lib, synthetic_conf())
    assert rc == 0
    for opt in (lib.OPT_BLOCKED_SAVES, lib.OPT_WGRAD_BF16X2, lib.OPT_SAMPLER_BF16X2):
        assert h.i2sdf_plan_set_option(plan, opt, 1) == 0 and h.i2sdf_plan_set_option(plan, opt, 0) == 0
    assert h.i2sdf_plan_set_option(plan, 64, 1) != 0           # (the LDS source ring of rounds 2-3 is gone)
    for opt in (lib.OPT_TRAIN_FWD_BF16X3, lib.OPT_SDF_BWD_BF16X3, lib.OPT_RGB_BF16X3, lib.OPT_TAIL_OVERLAP):
        assert h.i2sdf_plan_set_option(plan, opt, 1) == 0
    M = 1024 * 98 + 1024 * 2                                   # the training batch: 98 shaded + 2 eikonal points per ray
    Mp = (M + 127) // 128 * 128
    n_wg = Mp // 128
    bulk = (n_wg // 256) * 256 * 128                           # 3 full rounds of 256 workgroups, 32 workgroups of tail
    assert 0 < bulk < M and (n_wg - bulk // 128) * 4 <= 256
    assert h.i2sdf_plan_set_option(plan, lib.OPT_BLOCKED_SAVES, 0) == 0
    assert h.i2sdf_blocked_points(plan, 0, M, Mp, 1) == 0      # option off: point-major rows everywhere
    assert h.i2sdf_plan_set_option(plan, lib.OPT_BLOCKED_SAVES, 1) == 0
    assert h.i2sdf_blocked_points(plan, 0, M, Mp, 1) == bulk
    assert h.i2sdf_blocked_points(plan, 1, M, Mp, 1) == bulk
    assert h.i2sdf_blocked_points(plan, 0, M, Mp, 0) == Mp     # eikonal-only launches (no feature output) are never split
    M2 = 360 * 98
    Mp2 = (M2 + 127) // 128 * 128
    assert h.i2sdf_blocked_points(plan, 0, M2, Mp2, 1) == 256 * 128   # one full round + 20 tail workgroups
    M3 = 200 * 98                                                      # less than one round: no split, all blocked
    Mp3 = (M3 + 127) // 128 * 128
    assert h.i2sdf_blocked_points(plan, 0, M3, Mp3, 1) == Mp3
    assert h.i2sdf_blocked_points(None, 0, M, Mp, 1) == 0
    # round 3, I2SDF_OPT_PARTS: point ranges instead of a split-K tail -> every saved row is blocked; chains are no-ops while it is off
    assert h.i2sdf_chain_begin(plan, M, None) == 0 and h.i2sdf_chain_fence(plan, None) == 0 and h.i2sdf_chain_end(plan, None) == 0
    assert h.i2sdf_chain_begin(None, M, None) == -1 and h.i2sdf_chain_begin(plan, -1, None) == -1
    assert h.i2sdf_plan_set_option(plan, lib.OPT_PARTS, lib.MAX_PARTS + 1) == -1 and h.i2sdf_plan_set_option(plan, lib.OPT_PARTS, -1) == -1
    for n in (2, 3, lib.MAX_PARTS):
        assert h.i2sdf_plan_set_option(plan, lib.OPT_PARTS, n) == 0
        assert h.i2sdf_blocked_points(plan, 0, M, Mp, 1) == Mp and h.i2sdf_blocked_points(plan, 1, M, Mp, 1) == Mp
        assert h.i2sdf_blocked_points(plan, 0, M2, Mp2, 1) == Mp2
    for n in (0, 1):                                            # 0 / 1 = off: back to full rounds + split-K tail
        assert h.i2sdf_plan_set_option(plan, lib.OPT_PARTS, n) == 0
        assert h.i2sdf_blocked_points(plan, 0, M, Mp, 1) == bulk
    # round 6, I2SDF_OPT_SAVES24: packed 24-bit records of abars / gus / gas take effect only with point ranges + blocked saves + the bf16x3 families
    # + the two-plane weight gradients all on (which = 2: Mp or 0)
    assert h.i2sdf_blocked_points(plan, 2, M, Mp, 1) == 0                       # option off
    assert h.i2sdf_plan_set_option(plan, lib.OPT_SAVES24, 1) == 0
    assert h.i2sdf_blocked_points(plan, 2, M, Mp, 1) == 0                       # no point ranges, no two-plane weight gradients
    assert h.i2sdf_plan_set_option(plan, lib.OPT_PARTS, 2) == 0
    assert h.i2sdf_blocked_points(plan, 2, M, Mp, 1) == 0                       # still the fp32-equivalent weight gradients: fp32 storage
    assert h.i2sdf_plan_set_option(plan, lib.OPT_WGRAD_BF16X2, 1) == 0
    assert h.i2sdf_blocked_points(plan, 2, M, Mp, 1) == Mp and h.i2sdf_blocked_points(plan, 2, M2, Mp2, 1) == Mp2
    assert h.i2sdf_plan_set_option(plan, lib.OPT_BLOCKED_SAVES, 0) == 0 and h.i2sdf_blocked_points(plan, 2, M, Mp, 1) == 0
    assert h.i2sdf_plan_set_option(plan, lib.OPT_BLOCKED_SAVES, 1) == 0
    assert h.i2sdf_plan_set_option(plan, lib.OPT_SDF_BWD_BF16X3, 0) == 0 and h.i2sdf_blocked_points(plan, 2, M, Mp, 1) == 0
    assert h.i2sdf_plan_set_option(plan, lib.OPT_SDF_BWD_BF16X3, 1) == 0
    assert h.i2sdf_plan_set_option(plan, lib.OPT_SAVES24, 0) == 0 and h.i2sdf_blocked_points(plan, 2, M, Mp, 1) == 0
    h.i2sdf_plan_destroy(plan)
    rc, plan, _, _ = _plan(lib, plumbing_conf())                # 64-wide nets have no bf16x3 train path: nothing is blocked
    assert rc == 0
    h.i2sdf_plan_set_option(plan, lib.OPT_BLOCKED_SAVES, 1)
    assert h.i2sdf_blocked_points(plan, 0, M, Mp, 1) == 0
    assert h.i2sdf_plan_set_option(plan, lib.OPT_SAMPLER_BF16X2, 1) == -1      # the two-plane sampler stream exists for 256-wide nets only
    assert h.i2sdf_plan_set_option(plan, lib.OPT_SAVES24, 1) == -1             # ... and so do the packing kernels
    h.i2sdf_plan_destroy(plan)


def test_saved_to_point_major_matches_the_documented_layout():
    """Engine.saved_to_point_major against the address formula of include/i2sdf.h (row of point m at (m/32)*8192 + (m%32)*16,
    k-chunk kc at +512*kc) written as explicit loops."""
    import torch
    from i2sdf_amd.engine import RenderEngine as Engine
    Lr, Mp, nb = 2, 256, 128
    flat = torch.arange(Lr * Mp * 256, dtype=torch.float32).reshape(Lr, Mp, 256)
    got = Engine.saved_to_point_major(flat, nb)
    for l in range(Lr):
        base = flat[l].reshape(-1)
        for m in (0, 1, 31, 32, 77, 127):
            for kc in (0, 5, 15):
                off = (m // 32) * 8192 + (m % 32) * 16 + 512 * kc
                assert torch.equal(got[l, m, 16 * kc:16 * kc + 16], base[off:off + 16])
        assert torch.equal(got[l, nb:], flat[l, nb:])            # the tail behind the blocked prefix keeps ordinary rows
    assert torch.equal(Engine.saved_to_point_major(flat, 0), flat)


def test_missing_library_fails_loudly(monkeypatch, lib):
    from i2sdf_amd import lib as L
    monkeypatch.setattr(L, "_lib", None)
    monkeypatch.setattr(L, "LIB_PATH", "/nonexistent/libi2sdf_hip.so")
    with pytest.raises(L.I2SDFError):
        L.load()


def test_module_refuses_cpu_tensors():
    import torch
    from i2sdf_amd import I2SDFNetwork, plumbing_conf
    net = I2SDFNetwork(plumbing_conf())
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        net.implicit_network(torch.zeros(4, 3))


def test_saved_to_point_major_decodes_packed_24_bit_records():
    """I2SDF_OPT_SAVES24: the layout csrc/x3.h (p24_store8) documents, written as explicit loops by an independent encoder, decoded by
    Engine.saved_to_point_major: every value comes back rounded to 16 significant bits (relative error <= 2^-16), layers not packed stay exact."""
    import numpy as np
    import torch
    from i2sdf_amd.engine import RenderEngine as Engine
    g = torch.Generator().manual_seed(5)
    Lr, Mp = 2, 64
    x = (torch.randn(Lr, Mp, 256, generator=g) * torch.exp(4 * torch.randn(Lr, Mp, 256, generator=g))).float()
    x[0, 3, 7] = 0.0; x[0, 4, 9] = -0.0
    buf = np.zeros((Lr, Mp * 256), dtype=np.uint32)
    bits = x.numpy().view(np.uint32)
    for m in range(Mp):
        blk, p = m // 32, m % 32
        for kc in range(16):
            for hi in range(2):
                cols = [16 * kc + 4 * hi + u for u in range(4)] + [16 * kc + 8 + 4 * hi + u for u in range(4)]
                b = (bits[0, m, cols].astype(np.uint64) + 0x80).astype(np.uint32)          # layer 0 packed
                for j in range(4):
                    buf[0, blk * 6144 + kc * 384 + p * 8 + hi * 4 + j] = (b[2 * j] >> 16) | ((b[2 * j + 1] >> 16) << 16)
                for j in range(2):
                    w = 0
                    for k in range(4):
                        w |= ((int(b[4 * j + k]) >> 8) & 0xFF) << (8 * k)
                    buf[0, blk * 6144 + kc * 384 + 256 + p * 4 + hi * 2 + j] = w
                # layer 1: fp32, blocked
                for u, c in enumerate(cols):
                    buf[1, blk * 8192 + kc * 512 + p * 16 + (4 * hi + u if u < 4 else 8 + 4 * hi + u - 4)] = bits[1, m, c]
    t = torch.from_numpy(buf.view(np.float32).reshape(Lr, Mp, 256).copy())
    got = Engine.saved_to_point_major(t, Mp, [True, False])
    assert torch.equal(got[1], x[1])
    want = torch.from_numpy(((bits[0].astype(np.uint64) + 0x80).astype(np.uint32) & np.uint32(0xFFFFFF00)).view(np.float32))
    assert torch.equal(got[0], want)
    rel = ((got[0].double() - x[0].double()).abs() / x[0].double().abs().clamp_min(1e-300))[x[0] != 0]
    assert float(rel.max()) <= 2.0 ** -16 * 1.0001
