"""Synthetic scenes for tests and bench.

Camera matrices follow the reference's conventions (src/gs/camera.cu:36-82): the
tensors handed to the rasterizer are the TRANSPOSED 4x4 matrices, i.e. their row-major
memory is the column-major layout the kernels index (auxiliary.h:48-64).
Values are post-activation, as the rasterizer sees them (include/gs/gs/gaussian.cuh:40-54):
scales = exp(_scaling), rotations = normalize(_rotation), opacities = sigmoid(_opacity).
"""
import math

import numpy as np

ZNEAR, ZFAR = 0.01, 100.0  # src/gs/camera.cu:33-34


def projection_matrix(znear, zfar, fovx, fovy):
    """src/gs/camera.cu:57-82 (getProjectionMatrix); returns the math matrix P (not transposed)."""
    ty, tx = math.tan(fovy / 2.0), math.tan(fovx / 2.0)
    top, right = ty * znear, tx * znear
    bottom, left = -top, -right
    P = np.zeros((4, 4), np.float32)
    P[0, 0] = 2.0 * znear / (right - left)
    P[1, 1] = 2.0 * znear / (top - bottom)
    P[0, 2] = (right + left) / (right - left)
    P[1, 2] = (top + bottom) / (top - bottom)
    P[3, 2] = 1.0
    P[2, 2] = zfar / (zfar - znear)
    P[2, 3] = -(zfar * znear) / (zfar - znear)
    return P


def make_camera(W, H, fovx_deg=60.0, yaw_deg=0.0, position=(0.0, 0.0, 0.0)):
    """Camera at `position` looking down +z, rotated by yaw about +y (src/gs/camera.cu:36-48)."""
    fovx = math.radians(fovx_deg)
    fovy = 2.0 * math.atan(math.tan(fovx / 2.0) * H / W)
    a = math.radians(yaw_deg)
    R = np.array([[math.cos(a), 0, math.sin(a)], [0, 1, 0], [-math.sin(a), 0, math.cos(a)]], np.float32)  # cam->world
    T = np.asarray(position, np.float32)
    Tcw = np.eye(4, dtype=np.float32)
    Tcw[:3, :3] = R.T
    Tcw[:3, 3] = -R.T @ T
    view = np.ascontiguousarray(Tcw.T)  # world_view_transform tensor
    proj = np.ascontiguousarray(projection_matrix(ZNEAR, ZFAR, fovx, fovy).T)
    full = (view @ proj).astype(np.float32)
    campos = np.linalg.inv(view.astype(np.float64))[3, :3].astype(np.float32)
    return {
        "W": int(W), "H": int(H), "tanfovx": float(np.float32(math.tan(fovx * 0.5))),
        "tanfovy": float(np.float32(math.tan(fovy * 0.5))), "viewmatrix": view, "projmatrix": full, "campos": campos,
    }


def make_gaussians(P, seed, sh_degree=0, fovx_deg=60.0, aspect=16.0 / 9.0, zmin=1.0, zmax=40.0):
    """frustum-slab means (+2 % near-culled), anisotropic scales U(0.004,0.06) (+1 % with one
    axis in U(0.3,0.5) -> scale-culled), random unit quaternions, opacity U(0.05,0.95), f_dc U(-1.5,1.5),
    f_rest N(0,0.1^2)."""
    rng = np.random.default_rng(seed)
    M = (sh_degree + 1) ** 2
    t = math.tan(math.radians(fovx_deg) / 2.0)
    z = rng.uniform(zmin, zmax, P)
    near = rng.random(P) < 0.02
    z = np.where(near, rng.uniform(-5.0, 0.2, P), z)
    zabs = np.maximum(np.abs(z), 0.2)
    x = rng.uniform(-1, 1, P) * 1.1 * zabs * t
    y = rng.uniform(-1, 1, P) * 1.1 * zabs * t / aspect
    means = np.stack([x, y, z], 1).astype(np.float32)
    scales = rng.uniform(0.004, 0.06, (P, 3))
    big = rng.random(P) < 0.01
    ax = rng.integers(0, 3, P)
    scales[np.arange(P)[big], ax[big]] = rng.uniform(0.3, 0.5, int(big.sum()))
    q = rng.standard_normal((P, 4))
    q /= np.linalg.norm(q, axis=1, keepdims=True)
    opac = rng.uniform(0.05, 0.95, (P, 1))
    shs = np.zeros((P, M, 3))
    shs[:, 0, :] = rng.uniform(-1.5, 1.5, (P, 3))
    if M > 1:
        shs[:, 1:, :] = rng.standard_normal((P, M - 1, 3)) * 0.1
    return {
        "means3D": means, "scales": scales.astype(np.float32), "rotations": q.astype(np.float32),
        "opacities": opac.astype(np.float32), "shs": shs.astype(np.float32), "sh_degree": int(sh_degree),
    }


def make_scene(P, W, H, seed, sh_degree=0, yaw_deg=0.0, bg=(1.0, 1.0, 1.0)):
    s = make_gaussians(P, seed, sh_degree, aspect=W / H)
    s.update(make_camera(W, H, yaw_deg=yaw_deg))
    s["bg"] = np.asarray(bg, np.float32)
    s["scale_modifier"] = 1.0
    s["colors_precomp"] = None
    s["cov3D_precomp"] = None
    return s


def make_upstream_grads(W, H, seed):
    """Backward seeds of SURVEY.md 8(d): dL_dcolor ~ N(0,1)/(3HW), dL_dacc ~ N(0,1)/(HW)."""
    rng = np.random.default_rng(seed + 1000003)
    dcol = (rng.standard_normal((3, H, W)) / (3.0 * H * W)).astype(np.float32)
    dacc = (rng.standard_normal((1, H, W)) / (1.0 * H * W)).astype(np.float32)
    return dcol, dacc


# BASELINE.json configs (P, W, H, seed)
CONFIGS = {
    "C1": (10_000, 640, 480, 1),
    "C2": (500_000, 1280, 720, 2),
    "C3": (2_000_000, 1920, 1080, 3),
    "C4": (2_000_000, 1920, 1080, 3),  # the C3 scene seen from the eight C4_YAWS_DEG cameras (one view per GPU)
    # not a BASELINE config: the SHAPE of C5 (HKU-Campus replay renders 640x512 at SH degree 0 over a map that grows
    # to 1e5..1e6 Gaussians) on the synthetic scene -- the small-frame, host-bound regime
    "C5shape": (300_000, 640, 512, 5),
}
C4_YAWS_DEG = (-21.0, -15.0, -9.0, -3.0, 3.0, 9.0, 15.0, 21.0)
