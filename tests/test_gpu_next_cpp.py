"""GPU tests of the C++/LibTorch hosts of the "next" rows (csrc/torch_next.cpp, declared in gsr_torch_next.hpp:
what GS-LIVM's own translation units would call) against the Python hosts of the same C ABI entry points
(gs-livm_amd/loss.py, model.py, ply.py), which the other test files hold against plain Torch f32 restatements of the
reference.  Both routes launch the same kernels with the same arguments: results must be bit-identical."""
import os

import numpy as np
import pytest
import torch

import gs_livm_amd as G
from gs_livm_amd import _capi, ply
from gs_livm_amd import synthetic as S

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def nx(gpu_device):
    return G.torch_ops().next


def _leaves(P, M, dev, seed=0):
    gen = torch.Generator().manual_seed(seed)
    r = lambda *s: torch.randn(s, generator=gen).to(dev)  # noqa: E731
    return dict(xyz=r(P, 3), fdc=r(P, 1, 3), frest=r(P, M - 1, 3) * 0.1, scaling=r(P, 3) * 0.3 - 3.0,
                rotation=r(P, 4), opacity=r(P, 1))


def test_window_and_loss_match_the_python_route(nx, gpu_device):
    assert torch.equal(nx.reference_window_1d(), G.reference_window_1d())
    gen = torch.Generator().manual_seed(3)
    for shape, lam in (((3, 512, 640), 0.2), ((3, 67, 45), 0.8), ((1, 16, 16), 0.0)):
        img = torch.rand(shape, generator=gen).to(gpu_device)
        gt = (0.6 * torch.rand(shape, generator=gen).to(gpu_device) + 0.4 * img.roll(1, 2)).clamp(0, 1)
        a = img.clone().requires_grad_(True)
        b = img.clone().requires_grad_(True)
        la = nx.photometric_loss(a, gt, lam)
        lb = G.photometric_loss(b, gt, lam)
        assert la.shape == () and float(la) == float(lb)
        (ga,) = torch.autograd.grad(2.5 * la, a)
        (gb,) = torch.autograd.grad(2.5 * lb, b)
        assert torch.equal(ga, gb)
        parts = nx.photometric_loss_parts(img, gt, lam)
        assert float(parts[0]) == float(la) and parts.shape == (3,)
        # an explicit window (the centred Gaussian of the original SSIM code) goes through as given
        x = torch.arange(11, dtype=torch.float32) - 5
        w = torch.exp(-x * x / 4.5)
        w = w / w.sum()
        assert float(nx.photometric_loss(img, gt, lam, w)) == float(G.photometric_loss(img, gt, lam, window11=w))
    with pytest.raises(ValueError):
        nx.photometric_loss(img, gt[:, :-1], 0.2)


@pytest.mark.parametrize("M", [1, 4])
def test_activations_and_adam_match_the_python_route(M, nx, gpu_device):
    P = 10_007
    L = _leaves(P, M, gpu_device, seed=M)
    names = ("scaling", "rotation", "opacity", "fdc", "frest")
    a = [L[k].clone().requires_grad_(True) for k in names]
    b = [L[k].clone().requires_grad_(True) for k in names]
    out_a = nx.activate(*a)
    out_b = G.FusedActivations.apply(*b)
    for x, y in zip(out_a, out_b):
        assert torch.equal(x, y)
    ws = [torch.randn_like(x) for x in out_b]
    sum((x * w).sum() for x, w in zip(out_a, ws)).backward()
    sum((x * w).sum() for x, w in zip(out_b, ws)).backward()
    for x, y in zip(a, b):
        assert torch.equal(x.grad, y.grad)
    # Adam: the reference's six groups and learning rates (GaussianModel::Training_setup), three steps
    order = ("xyz", "fdc", "frest", "scaling", "rotation", "opacity")
    lrs = [0.0005, 0.001, 0.001 / 20.0, 0.0025, 0.0025, 0.025]
    pa = [L[k].clone().requires_grad_(True) for k in order]
    pb = [L[k].clone().requires_grad_(True) for k in order]
    opt_a = nx.FusedAdam(pa, lrs, 0.9, 0.999, 1e-15)
    opt_b = G.FusedAdam([{"params": [p], "lr": lr} for p, lr in zip(pb, lrs) if p.numel()], eps=1e-15)
    gen = torch.Generator().manual_seed(9)
    for it in range(3):
        for x, y in zip(pa, pb):
            g = torch.randn(x.shape, generator=gen).to(gpu_device)
            x.grad, y.grad = g.clone(), g.clone()
        opt_a.step(True)
        opt_b.step(zero_grads=True)
        for x, y in zip(pa, pb):
            assert torch.equal(x, y)
            if x.numel():
                assert not x.grad.any() and not y.grad.any()   # cleared by the kernel that consumed them
    assert opt_a.step_count() == 3
    # the one-kernel tail: chain rule + Adam + next activations
    ma = [t.clone() for t in opt_a.exp_avg()]
    va = [t.clone() for t in opt_a.exp_avg_sq()]
    g_act = [torch.randn(s, generator=gen).to(gpu_device) for s in ((P, 3), (P, 3), (P, 4), (P, 1), (P, M, 3))]
    ref_p = [p.detach().clone() for p in pa]
    want = _capi.model_step(ref_p, ma, va, *g_act, lrs, 0.9, 0.999, 1e-15, 4)
    got = opt_a.step_model(*g_act)
    for x, y in zip(got, want):
        assert torch.equal(x, y)
    for x, y in zip(opt_a.params(), ref_p):
        assert torch.equal(x, y)
    # growing a leaf: moments get zero rows (cat_tensors_to_optimizer, gaussian.cu:451-472)
    bigger = torch.cat([opt_a.params()[0].detach(), torch.zeros((5, 3), device=gpu_device)], 0)
    old_m = opt_a.exp_avg()[0].clone()
    opt_a.replace_param(0, bigger)
    assert opt_a.exp_avg()[0].shape == (P + 5, 3) and torch.equal(opt_a.exp_avg()[0][:P], old_m)
    assert not opt_a.exp_avg()[0][P:].any() and not opt_a.exp_avg_sq()[0][P:].any()


@pytest.mark.parametrize("M", [1, 4])
def test_growth_and_ply_export_match_the_python_route(M, nx, gpu_device, tmp_path):
    n, dev = 4099, gpu_device
    gen = torch.Generator().manual_seed(2)
    xyz = torch.randn((n, 3), generator=gen).to(dev)
    A = torch.randn((n, 3, 3), generator=gen).to(dev) * 0.05
    covs = A @ A.transpose(1, 2) + 1e-4 * torch.eye(3, device=dev)
    rgbs = torch.rand((n, 3), generator=gen).to(dev) * 255

    def outs():
        f = dict(dtype=torch.float32, device=dev)
        return [torch.full((n, 3), 7.0, **f), torch.full((n, 1, 3), 7.0, **f), torch.full((n, M - 1, 3), 7.0, **f),
                torch.full((n, 3), 7.0, **f), torch.full((n, 4), 7.0, **f), torch.full((n, 1), 7.0, **f)]
    oa, ob = outs(), outs()
    nx.init_gaussians(xyz, covs, rgbs, 1.5, *oa)
    _capi.init_gaussians(xyz, covs, rgbs, 1.5, *ob)
    for x, y in zip(oa, ob):
        assert torch.equal(x, y) and (x.numel() == 0 or not (x == 7.0).all())
    o_xyz, o_dc, o_rest, o_scal, o_rot, o_op = oa
    rows_a = nx.pack_ply_rows(o_xyz, o_dc, o_rest, o_op, o_scal, o_rot)
    rows_b = _capi.pack_ply_rows(o_xyz, o_dc, o_rest, o_op, o_scal, o_rot)
    assert torch.equal(rows_a, rows_b) and rows_a.shape == (n, 14 + 3 * M)
    assert nx.ply_attribute_names(M) == ply.attribute_names(M)
    path = os.path.join(str(tmp_path), "cpp.ply")
    nbytes = nx.write_ply(path, o_xyz, o_dc, o_rest, o_op, o_scal, o_rot)
    blob = open(path, "rb").read()
    assert nbytes == len(blob) and blob == ply.ply_bytes(rows_b.cpu().numpy(), M)   # = tinyply's bytes (test_ply.py)
    back = ply.load_ply(path)
    assert np.array_equal(back["xyz"], o_xyz.cpu().numpy()) and np.array_equal(back["rotation"], o_rot.cpu().numpy())
