"""Measurement of "next" row 4 on one MI355X: growing a 2 M-Gaussian model by n points and exporting it to PLY,
product path vs the reference's procedure restated with Torch ops (torch.cat of six parameter tensors + twelve
Adam moments, src/gs/gaussian.cu:451-472, 524-540; seven .cpu() copies + host interleave, :494-573).

usage: python tools/bench_growth.py [P=2000000] [n=20000] [M=1]   -> one JSON line
Algorithmic bytes: growth = n * (14 + 3M) * 4 B written (+ 60 B/point read); the cat procedure moves
3 * P * (14 + 3M) * 4 B read + as much written (parameters + two moments).  Export = P * (14 + 3M) * 4 B read and
written on the device, then the same over PCIe."""
import json, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
import numpy as np
import torch
import gs_livm_amd as G
from gs_livm_amd import ply

P = int(sys.argv[1]) if len(sys.argv) > 1 else 2_000_000
n = int(sys.argv[2]) if len(sys.argv) > 2 else 20_000
M = int(sys.argv[3]) if len(sys.argv) > 3 else 1
dev = torch.device("cuda:0")
gen = torch.Generator(device="cpu").manual_seed(0)
r = lambda *s: torch.randn(s, generator=gen).to(dev)  # noqa: E731
xyz, rgbs = r(n, 3), torch.rand((n, 3), generator=gen).to(dev) * 255
A = r(n, 3, 3) * 0.05
covs = A @ A.transpose(1, 2) + 1e-4 * torch.eye(3, device=dev)


def timed(fn, reps=10):
    fn(); torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / reps * 1e3


# ---- product: capacity buffers, in-place initialisation ----
m = G.GrowableGaussians(P + 64 * n, M, dev)
m.add_new_pointcloud(r(P, 3), (r(P, 3, 3) * 0.05) @ (r(P, 3, 3) * 0.05).transpose(1, 2).abs() + 1e-3 * torch.eye(3, device=dev), torch.rand((P, 3), generator=gen).to(dev) * 255)
opt = G.GrowableAdam(m)
ms_grow = timed(lambda: m.add_new_pointcloud(xyz, covs, rgbs, 1.0))

# ---- the reference's procedure with Torch ops ----
names = G.GrowableGaussians._NAMES
ref_p = {k: getattr(m, k).detach()[:P].clone() for k in names}
ref_m = {k: torch.zeros_like(v) for k, v in ref_p.items()}
ref_v = {k: torch.zeros_like(v) for k, v in ref_p.items()}


def grow_ref():
    scale_p = covs.diagonal(0, -2, -1)
    new = dict(_xyz=xyz, _scaling=torch.log(torch.sqrt(scale_p * 1.0)),
               _rotation=torch.nn.functional.pad(torch.ones((n, 1), device=dev), (0, 3)),
               _opacity=torch.zeros((n, 1), device=dev),
               _features_dc=((rgbs / 255.0 - 0.5) / 0.28209479177387814).unsqueeze(1),
               _features_rest=torch.zeros((n, M - 1, 3), device=dev))
    out = {}
    for k in names:
        out[k] = (torch.cat([ref_p[k], new[k]], 0), torch.cat([ref_m[k], torch.zeros_like(new[k])], 0),
                  torch.cat([ref_v[k], torch.zeros_like(new[k])], 0))
    return out


ms_grow_ref = timed(grow_ref, reps=5)

# ---- export ----
Pn = m.P
rf = 14 + 3 * M
ms_pack = timed(lambda: G._capi.pack_ply_rows(m._xyz, m._features_dc, m._features_rest, m._opacity, m._scaling, m._rotation))
t0 = time.perf_counter(); path = ply.save_ply("/tmp/gsr_ply_bench", m); ms_save = (time.perf_counter() - t0) * 1e3


def export_ref():
    cols = [m._xyz.detach().cpu(), torch.zeros_like(m._xyz).cpu(), m._features_dc.detach().transpose(1, 2).flatten(1).cpu(),
            m._features_rest.detach().transpose(1, 2).flatten(1).cpu(), m._opacity.detach().cpu(), m._scaling.detach().cpu(),
            m._rotation.detach().cpu()]
    return np.concatenate([c.numpy() for c in cols], 1)  # the interleave tinyply performs on the host


t0 = time.perf_counter(); export_ref(); ms_export_ref = (time.perf_counter() - t0) * 1e3
row_bytes = rf * 4
print(json.dumps({
    "P": P, "n_new": n, "M": M,
    "grow_ms": round(ms_grow, 4), "grow_reference_procedure_ms": round(ms_grow_ref, 4),
    "grow_speedup": round(ms_grow_ref / ms_grow, 1),
    "grow_reference_GBps": round(3 * 2 * P * row_bytes * (14 + 3 * M - 3) / rf / (ms_grow_ref * 1e-3) / 1e9, 1),
    "pack_rows_ms": round(ms_pack, 4), "pack_rows_GBps": round(2 * Pn * row_bytes / (ms_pack * 1e-3) / 1e9, 1),
    "save_ply_ms_incl_D2H_and_file": round(ms_save, 2), "reference_host_interleave_ms_excl_file": round(ms_export_ref, 2),
    "file_bytes": os.path.getsize(path)}))
