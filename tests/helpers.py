"""Shared helpers for the parity tests: numpy scene -> device tensors -> C ABI -> numpy."""
import numpy as np
import torch

import gs_livm_amd as G


def to_dev(scene, dev):
    t = {}
    for k in ("bg", "means3D", "shs", "opacities", "scales", "rotations", "viewmatrix", "projmatrix", "campos",
              "colors_precomp", "cov3D_precomp"):
        v = scene.get(k)
        t[k] = torch.empty(0, device=dev) if v is None else torch.from_numpy(np.ascontiguousarray(v)).to(dev)
    return t


def hip_forward(scene, dev, debug=True):
    t = to_dev(scene, dev)
    out = G.rasterize_forward(t["bg"], t["means3D"], t["colors_precomp"], t["opacities"], t["scales"],
                              t["rotations"], scene.get("scale_modifier", 1.0), t["cov3D_precomp"], t["viewmatrix"],
                              t["projmatrix"], scene["tanfovx"], scene["tanfovy"], scene["H"], scene["W"], t["shs"],
                              scene["sh_degree"], t["campos"], False, debug)
    return t, out


def hip_backward(scene, t, fwd, dL_dcolor, dL_dacc, dev, debug=True):
    R, color, depth, acc, radii, geom, binning, img = fwd
    dc = torch.from_numpy(dL_dcolor).to(dev)
    da = torch.from_numpy(dL_dacc).to(dev)
    g = G.rasterize_backward(t["bg"], t["means3D"], radii, t["colors_precomp"], t["scales"], t["rotations"],
                             scene.get("scale_modifier", 1.0), t["cov3D_precomp"], t["viewmatrix"], t["projmatrix"],
                             scene["tanfovx"], scene["tanfovy"], dc, da, t["shs"], scene["sh_degree"], t["campos"],
                             geom, R, binning, img, debug, return_conic=True)
    names = ("dL_dmeans2D", "dL_dcolors", "dL_dopacity", "dL_dmeans3D", "dL_dcov3D", "dL_dsh", "dL_dscales",
             "dL_drotations", "dL_dconic")
    return {n: x.cpu().numpy() for n, x in zip(names, g)}


def grad_close(got, ref, name):
    """SURVEY.md Appendix B tolerance, |d| <= 1e-5 * max|g| + 1e-4 * |g|, with |g| taken as the largest
    component of the SAME Gaussian's gradient group (row): the f32 summation order differs from the
    oracle's, and the conic -> cov2D -> cov3D chain cancels large terms, so one component of a group can
    carry the rounding of its siblings (observed: inputs equal to 7 digits, one output off by 1.2e-4 rel)."""
    ref = ref.reshape(got.shape)
    if ref.size == 0:
        return
    P = ref.shape[0]
    scale = float(np.abs(ref).max())
    rowmax = np.abs(ref.reshape(P, -1)).max(1).reshape((P,) + (1,) * (ref.ndim - 1))
    tol = 1e-5 * scale + 1e-4 * rowmax
    bad = np.abs(got - ref) > tol
    assert not bad.any(), "%s: %d / %d outside tolerance, worst |d|=%.3e (max|g|=%.3e)" % (
        name, int(bad.sum()), bad.size, float(np.abs(got - ref).max()), scale)
