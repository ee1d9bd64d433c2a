"""GPU: grad_theta / diff_norm of the extra points (model/network/__init__.py:188-193) -- the fused HIP forward and backward
(i2sdf_eikonal_outputs_*) against the oracle's formula (oracle/i2sdf_oracle.py render(): F.normalize(eps=1e-6) + torch.norm)
evaluated by torch autograd in fp64, including the degenerate rows (zero gradient, gradient below eps, identical normals)."""
import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu


def _reference(g_all, B, w_theta, w_diff):
    g = g_all.double().clone().requires_grad_(True)
    theta = g[: 2 * B]
    nrm = F.normalize(g[B:], dim=1, eps=1e-6)
    diff = torch.norm(nrm[:B] - nrm[B:], dim=1)
    ((theta * w_theta.double()).sum() + (diff * w_diff.double()).sum()).backward()
    return theta.detach(), diff.detach(), g.grad


@pytest.mark.parametrize("B", [1, 300, 1024])
def test_eikonal_outputs_match_formula(B):
    from i2sdf_amd.network import _EikonalOutputsFn
    gen = torch.Generator().manual_seed(B)
    g_all = torch.randn(3 * B, 3, generator=gen)
    if B >= 300:
        g_all[B + 1] = 0.0                                   # zero gradient at a near-surface point
        g_all[2 * B + 2] = 0.0                               # ... and at a neighbour
        g_all[B + 3] = torch.tensor([3e-7, -2e-7, 1e-7])     # norm below the normalisation eps
        g_all[2 * B + 4] = g_all[B + 4] * 2.0                # identical normals: diff_norm = 0, d||x|| taken as 0
        g_all[B + 5] = 0.0; g_all[2 * B + 5] = 0.0
    w_theta, w_diff = torch.randn(2 * B, 3, generator=gen), torch.randn(B, generator=gen)
    t_ref, d_ref, gb_ref = _reference(g_all, B, w_theta, w_diff)
    x = g_all.cuda().requires_grad_(True)
    theta, diff = _EikonalOutputsFn.apply(x, B)
    assert torch.equal(theta.detach().cpu(), g_all[: 2 * B])
    assert torch.allclose(diff.detach().cpu().double(), d_ref, rtol=2e-6, atol=2e-6)
    ((theta * w_theta.cuda()).sum() + (diff * w_diff.cuda()).sum()).backward()
    got = x.grad.cpu().double()
    scale = gb_ref.abs().max()
    big = gb_ref.abs() > 1e3                                   # rows below eps have 1/eps-sized gradients: compare those relatively
    assert torch.allclose(got[~big], gb_ref[~big], rtol=1e-5, atol=1e-5), float((got - gb_ref)[~big].abs().max())
    if big.any():
        assert torch.allclose(got[big], gb_ref[big], rtol=1e-4), float(((got - gb_ref)[big] / scale).abs().max())


def test_eikonal_outputs_partial_gradients():
    """Only one of the two outputs used by the loss (smoothness term inactive, or eikonal weight 0): the other gradient is None."""
    from i2sdf_amd.network import _EikonalOutputsFn
    B = 257
    gen = torch.Generator().manual_seed(9)
    g_all = torch.randn(3 * B, 3, generator=gen)
    w_theta, w_diff = torch.randn(2 * B, 3, generator=gen), torch.randn(B, generator=gen)
    zero_t, zero_d = torch.zeros_like(w_theta), torch.zeros_like(w_diff)
    for wt, wd, use_t, use_d in ((w_theta, zero_d, True, False), (zero_t, w_diff, False, True)):
        _, _, gb_ref = _reference(g_all, B, wt, wd)
        x = g_all.cuda().requires_grad_(True)
        theta, diff = _EikonalOutputsFn.apply(x, B)
        loss = (theta * wt.cuda()).sum() if use_t else (diff * wd.cuda()).sum()
        loss.backward()
        assert torch.allclose(x.grad.cpu().double(), gb_ref, rtol=1e-5, atol=1e-5)
