"""GPU tests of the fused caller-side kernels (csrc/optimizer.hip; SURVEY.md 8(f) "next" row 1) against the
Torch ops the reference runs (include/gs/gs/gaussian.cuh:40-54, torch::optim::Adam as set up in
src/gs/gaussian.cu:396-428).  This is floating-point work: the reference here IS plain PyTorch f32."""
import numpy as np
import pytest
import torch

import gs_livm_amd as G
from gs_livm_amd import synthetic as S

pytestmark = pytest.mark.gpu


def _raw(P, D, dev, seed=0):
    g = S.make_gaussians(P, seed, sh_degree=D)
    t = lambda a: torch.from_numpy(np.ascontiguousarray(a, dtype=np.float32)).to(dev)  # noqa: E731
    return dict(xyz=t(g["means3D"]), scaling=t(np.log(g["scales"])),
                rotation=t(g["rotations"] * np.random.default_rng(seed).uniform(0.5, 2.0, (P, 1))),
                opacity=t(np.log(g["opacities"] / (1 - g["opacities"]))), f_dc=t(g["shs"][:, :1]),
                f_rest=t(g["shs"][:, 1:]))


@pytest.mark.parametrize("P,D", [(1001, 1), (4096, 0), (777, 3)])
def test_fused_activations_match_torch_ops(P, D, gpu_device):
    r = _raw(P, D, gpu_device, seed=P)
    leaves = {k: v.clone().requires_grad_(True) for k, v in r.items() if k != "xyz"}
    ref = (torch.exp(leaves["scaling"]), torch.nn.functional.normalize(leaves["rotation"]),
           torch.sigmoid(leaves["opacity"]), torch.cat([leaves["f_dc"], leaves["f_rest"]], 1))
    mine_leaves = {k: v.clone().requires_grad_(True) for k, v in r.items() if k != "xyz"}
    got = G.FusedActivations.apply(mine_leaves["scaling"], mine_leaves["rotation"], mine_leaves["opacity"],
                                   mine_leaves["f_dc"], mine_leaves["f_rest"])
    for a, b in zip(got, ref):
        assert a.shape == b.shape
        torch.testing.assert_close(a, b, rtol=2e-6, atol=1e-7)
    gen = torch.Generator(device="cpu").manual_seed(1)
    ups = [torch.randn(x.shape, generator=gen).to(gpu_device) for x in ref]
    torch.autograd.backward(list(ref), ups)
    torch.autograd.backward(list(got), ups)
    for k in leaves:
        if leaves[k].numel():
            torch.testing.assert_close(mine_leaves[k].grad, leaves[k].grad, rtol=1e-5, atol=1e-6)


def test_fused_adam_matches_torch_adam(gpu_device):
    dev = gpu_device
    gen = torch.Generator(device="cpu").manual_seed(3)
    shapes = [(1001, 3), (1001, 1, 3), (1001, 3, 3), (1001, 3), (1001, 4), (1001, 1), (5,), (8, 4)]  # odd tails
    lrs = [5e-4, 1e-3, 5e-5, 2.5e-3, 2.5e-3, 2.5e-2, 1e-2, 3e-3]
    base = [torch.randn(s, generator=gen).to(dev) for s in shapes]
    pa = [torch.nn.Parameter(b.clone()) for b in base]
    pb = [torch.nn.Parameter(b.clone()) for b in base]
    # one parameter is a misaligned view (8-byte offset): exercises the scalar path
    flat = torch.zeros(2 + 1001 * 4, device=dev)
    pb[4] = torch.nn.Parameter(flat[2:].view(1001, 4))
    with torch.no_grad():
        pb[4].copy_(base[4])
    ref = torch.optim.Adam([{"params": [p], "lr": lr} for p, lr in zip(pa, lrs)], eps=1e-15)
    mine = G.FusedAdam([{"params": [p], "lr": lr} for p, lr in zip(pb, lrs)], eps=1e-15)
    for it in range(6):
        for a, b in zip(pa, pb):
            g = torch.randn(a.shape, generator=gen).to(dev) * (10.0 ** (it - 3))
            a.grad = g.clone()
            if b.grad is None:
                b.grad = torch.zeros_like(b)
            b.grad.add_(g)  # FusedAdam zeroes the gradients it consumed: accumulate like autograd does
        ref.step()
        mine.step()
        for a, b in zip(pa, pb):
            torch.testing.assert_close(b.data, a.data, rtol=2e-6, atol=1e-7)
            assert not b.grad.any()
    # moments: the kernel uses the C++ torch::optim::Adam update m = m*b1 + g*(1-b1) (what the reference links);
    # the Python optimiser uses lerp_, which rounds differently when g jumps by orders of magnitude
    for a, b in zip(pa, pb):
        for key in ("exp_avg", "exp_avg_sq"):
            r = ref.state[a][key]
            torch.testing.assert_close(mine.state[b][key], r, rtol=5e-5, atol=1e-5 * float(r.abs().max()))


# M = 4 / 9 / 16 (odd last workgroup at M = 9) and M = 1: SH degree 0 is the product's setting (parameters.cuh:39) and
# the one bench.py times -- k_model_step's sh_direct branch, an exact multiple of the workgroup size and an odd count
@pytest.mark.parametrize("P,D", [(3000, 1), (3001, 2), (1500, 3), (4096, 0), (4097, 0)])
def test_one_optimiser_iteration_fused_vs_torch_ops(P, D, gpu_device):
    """activations -> rasterizer -> backward -> Adam: the fused path and the reference's separate Torch ops
    (gaussian.cuh:40-54 getters + torch::optim::Adam as configured in gaussian.cu:396-428) end in the same parameters."""
    dev = gpu_device
    W, H = 200, 120
    r = _raw(P, D, dev, seed=9)
    cam = S.make_camera(W, H)
    st = G.GaussianRasterizationSettings(H, W, cam["tanfovx"], cam["tanfovy"], torch.ones(3, device=dev), 1.0,
                                         torch.from_numpy(cam["viewmatrix"]).to(dev),
                                         torch.from_numpy(cam["projmatrix"]).to(dev), D,
                                         torch.from_numpy(cam["campos"]).to(dev), False)
    dcol, dacc = S.make_upstream_grads(W, H, 9)
    wc, wa = torch.from_numpy(dcol).to(dev) * 1e3, torch.from_numpy(dacc).to(dev) * 1e3

    def run(fused, tail=False):
        m = G.GaussianParameters(r["xyz"].clone(), r["f_dc"].clone(), r["f_rest"].clone(), r["scaling"].clone(),
                                 r["rotation"].clone(), r["opacity"].clone())
        m.fused_tail = tail
        opt = (G.FusedAdam if fused else torch.optim.Adam)(m.param_groups(), eps=1e-15)
        for _ in range(3):
            if fused:
                xyz, op, sc, rot, shs = m.activated()
            else:
                xyz, op, sc, rot, shs = (m._xyz, torch.sigmoid(m._opacity), torch.exp(m._scaling),
                                         torch.nn.functional.normalize(m._rotation),
                                         torch.cat([m._features_dc, m._features_rest], 1))
            means2D = torch.zeros_like(xyz, requires_grad=True)
            color, radii, depth, acc = G.GaussianRasterizer(st)(xyz, means2D, op, shs=shs, scales=sc, rotations=rot)
            torch.autograd.backward([color, acc], [wc, wa])
            if tail:
                assert m._scaling.grad is None and m._act_grads is not None  # raw-space gradients never materialise
                opt.step_model(m)
                assert m._next_act is not None and m._act_grads is None
            else:
                opt.step()
            if not fused:
                opt.zero_grad(set_to_none=True)
        out = [p.detach().clone() for p in m.parameters()]
        if tail:  # the activations the step left behind are those of the updated parameters
            sc, rot, op, shs = m._next_act
            torch.testing.assert_close(sc, torch.exp(m._scaling.detach()), rtol=2e-6, atol=1e-7)
            torch.testing.assert_close(rot, torch.nn.functional.normalize(m._rotation.detach()), rtol=2e-6, atol=1e-7)
            torch.testing.assert_close(op, torch.sigmoid(m._opacity.detach()), rtol=2e-6, atol=1e-7)
            torch.testing.assert_close(shs, torch.cat([m._features_dc, m._features_rest], 1).detach(), rtol=0, atol=0)
        return out

    a, b, c = run(True), run(False), run(True, tail=True)
    for x, y in zip(a, b):
        torch.testing.assert_close(x, y, rtol=1e-4, atol=2e-6)
    for x, z in zip(a, c):  # one-kernel tail vs three kernels: the same arithmetic
        torch.testing.assert_close(z, x, rtol=5e-6, atol=1e-7)
