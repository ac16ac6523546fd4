"""GPU tests of "next" row 4 (csrc/growth.hip, gs-livm_amd/model.py GrowableGaussians / GrowableAdam, ply.py):
map growth against the reference's procedure restated with plain Torch f32 ops (GaussianModel::addNewPointcloud
+ densification_postfix + cat_tensors_to_optimizer, src/gs/gaussian.cu:241-313, 451-472, 524-540), and the PLY
export through the device row packer against files written by the reference's vendored tinyply."""
import os

import numpy as np
import pytest
import torch

import gs_livm_amd as G
from gs_livm_amd import ply

pytestmark = pytest.mark.gpu
GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
C0 = 0.28209479177387814


def _cloud(n, seed, dev):
    r = np.random.default_rng(seed)
    A = r.standard_normal((n, 3, 3)).astype(np.float32) * 0.05
    covs = A @ A.transpose(0, 2, 1) + 1e-4 * np.eye(3, dtype=np.float32)
    t = lambda a: torch.from_numpy(np.ascontiguousarray(a, np.float32)).to(dev)  # noqa: E731
    return t(r.uniform(-3, 3, (n, 3))), t(covs), t(r.integers(0, 256, (n, 3)).astype(np.float32))


def _reference_new_rows(xyz, covs, rgbs, scale_factor, M):
    """addNewPointcloud's tensor algebra (src/gs/gaussian.cu:276-310), plain Torch f32."""
    n = xyz.shape[0]
    scale_p = covs.diagonal(0, -2, -1)
    scaling = torch.log(torch.sqrt(scale_p * scale_factor))
    rotation = torch.zeros((n, 4), device=xyz.device)
    rotation[:, 0] = 1
    opacity = torch.log(0.5 * torch.ones((n, 1), device=xyz.device) / (1 - 0.5 * torch.ones((n, 1), device=xyz.device)))
    fused = (rgbs / 255.0 - 0.5) / np.float32(C0)
    feats = torch.zeros((n, 3, M), device=xyz.device)
    feats[:, :3, 0] = fused
    return dict(_xyz=xyz, _features_dc=feats[:, :, 0:1].transpose(1, 2).contiguous(),
                _features_rest=feats[:, :, 1:].transpose(1, 2).contiguous(), _scaling=scaling, _rotation=rotation,
                _opacity=opacity)


@pytest.mark.parametrize("n,M", [(1, 1), (1000, 4), (4097, 16)])
def test_init_kernel_matches_the_references_tensor_algebra(n, M, gpu_device):
    xyz, covs, rgbs = _cloud(n, n, gpu_device)
    m = G.GrowableGaussians(8, M, gpu_device)
    lo, hi = m.add_new_pointcloud(xyz, covs, rgbs, scale_factor=1.7)
    assert (lo, hi) == (0, n) and m.P == n and m.capacity >= n
    ref = _reference_new_rows(xyz, covs, rgbs, 1.7, M)
    for k, v in ref.items():
        got = getattr(m, k).detach()
        assert got.shape == v.shape, k
        torch.testing.assert_close(got, v, rtol=2e-6, atol=1e-7)
    assert torch.equal(m._rotation.detach()[:, 0], torch.ones(n, device=gpu_device)) and not m._opacity.detach().any()


def test_growth_keeps_parameters_and_adam_state_like_the_references_cat(gpu_device):
    """Three growth events (one crosses the capacity) interleaved with optimiser steps: the capacity-buffer
    model + GrowableAdam end where the reference's procedure does -- torch.cat of every parameter, zeros appended
    to both Adam moments, the step count kept (cat_tensors_to_optimizer)."""
    dev, M = gpu_device, 4
    mine = G.GrowableGaussians(1500, M, dev)
    opt = G.GrowableAdam(mine)
    names = G.GrowableGaussians._NAMES
    lrs = [g["lr"] for g in mine.param_groups()]
    ref_p = {k: torch.zeros((0,) + getattr(mine, k).shape[1:], device=dev) for k in names}
    ref_m = {k: torch.zeros_like(v) for k, v in ref_p.items()}
    ref_v = {k: torch.zeros_like(v) for k, v in ref_p.items()}
    gen = torch.Generator(device="cpu").manual_seed(5)
    step = 0
    for event, n in enumerate((1000, 400, 2100)):  # 1000 -> 1400 -> 3500 (> 1500: the buffers double)
        xyz, covs, rgbs = _cloud(n, 100 + event, dev)
        old = {k: getattr(mine, k).detach().clone() for k in names}
        mine.add_new_pointcloud(xyz, covs, rgbs, scale_factor=1.0)
        new = _reference_new_rows(xyz, covs, rgbs, 1.0, M)
        for k in names:
            assert torch.equal(getattr(mine, k).detach()[:old[k].shape[0]], old[k]), k  # old rows untouched, bit for bit
            ref_p[k] = torch.cat([ref_p[k], new[k]], 0)
            ref_m[k] = torch.cat([ref_m[k], torch.zeros_like(new[k])], 0)
            ref_v[k] = torch.cat([ref_v[k], torch.zeros_like(new[k])], 0)
            with torch.no_grad():
                getattr(mine, k).copy_(ref_p[k])  # same starting point (init kernel differs by <= 2 ulp)
        for _ in range(3):
            step += 1
            for k, lr in zip(names, lrs):
                if not ref_p[k].numel():
                    continue
                g = torch.randn(ref_p[k].shape, generator=gen).to(dev)
                getattr(mine, k).grad = g.clone()
                # torch::optim::Adam (no weight decay / amsgrad), shared step count for old and new rows
                ref_m[k] = ref_m[k] * 0.9 + g * 0.1
                ref_v[k] = ref_v[k] * 0.999 + g * g * 0.001
                bc1, bc2 = 1 - 0.9 ** step, 1 - 0.999 ** step
                denom = (ref_v[k].sqrt() / np.sqrt(bc2)) + 1e-15
                ref_p[k] = ref_p[k] - (lr / bc1) * ref_m[k] / denom
            opt.step()
        for k in names:
            if ref_p[k].numel():
                torch.testing.assert_close(getattr(mine, k).detach(), ref_p[k], rtol=2e-5, atol=1e-7)
                m, v = mine.moments(k)
                # f32 rounding of m*b1 + g*(1-b1) near cancellation: absolute slack of a few 1e-8 on values of O(0.1)
                torch.testing.assert_close(m, ref_m[k], rtol=1e-5, atol=2e-7)
                torch.testing.assert_close(v, ref_v[k], rtol=1e-5, atol=1e-8)
    assert mine.P == 3500 and mine.capacity >= 3500


@pytest.mark.parametrize("name,P,M", [("ply_P7_M4", 7, 4), ("ply_P300_M1", 300, 1), ("ply_P33_M16", 33, 16)])
def test_save_ply_writes_the_references_bytes(name, P, M, gpu_device, tmp_path):
    """Model on the device -> k_pack_ply_rows -> one D2H copy -> file: identical to what the reference's vendored
    tinyply wrote for the same tensors (tests/golden, make_golden_ply.py)."""
    z = np.load(os.path.join(GOLDEN, name + ".npz"))
    t = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(gpu_device)  # noqa: E731
    m = G.GaussianParameters(t(z["xyz"]), t(z["features_dc"]), t(z["features_rest"]), t(z["scaling"]),
                             t(z["rotation"]), t(z["opacity"]))
    path = ply.save_ply(tmp_path, m, iteration=7)
    assert path.endswith(os.path.join("point_cloud", "iteration_7", "point_cloud.ply"))  # Save_ply's folder rule
    assert open(path, "rb").read() == open(os.path.join(GOLDEN, name + ".ply"), "rb").read()
    rows = G._capi.pack_ply_rows(m._xyz, m._features_dc, m._features_rest, m._opacity, m._scaling, m._rotation)
    assert np.array_equal(rows.cpu().numpy(), ply.rows_numpy(**{k: z[k] for k in z.files}))


def test_pack_rows_large_model_round_trip(gpu_device, tmp_path):
    P, M = 200_003, 4
    gen = torch.Generator(device="cpu").manual_seed(2)
    r = lambda *s: torch.randn(s, generator=gen).to(gpu_device)  # noqa: E731
    m = G.GaussianParameters(r(P, 3), r(P, 1, 3), r(P, M - 1, 3), r(P, 3), r(P, 4), r(P, 1))
    back = ply.load_ply(ply.save_ply(tmp_path, m))
    for k, name in (("xyz", "_xyz"), ("features_dc", "_features_dc"), ("features_rest", "_features_rest"),
                    ("opacity", "_opacity"), ("scaling", "_scaling"), ("rotation", "_rotation")):
        assert np.array_equal(back[k], getattr(m, name).detach().cpu().numpy()), k
