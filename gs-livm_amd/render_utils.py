"""Host-side mirror of the two callers' views of the rasterizer that sit directly on top of the operator:

* `Camera`  -- the part of the reference's Camera the rasterizer consumes (include/gs/gs/camera.cuh:47-110,
  src/gs/camera.cu:14-57): world_view_transform (the transposed world-to-camera matrix), the projection of
  getProjectionMatrix (camera.cu:59-82, znear 0.01, zfar 100), full_proj_transform = view @ projection and the
  camera centre, with the reference's getter names.
* `render`  -- include/gs/gs/render_utils.cuh:13-56: settings from the camera, activated parameters from the model,
  one rasterizer call, (color, depth, depth_acc) back.  Differences, both in the caller's favour: no
  `torch.cuda.synchronize()` before the rasterizer (render_utils.cuh:49) and scalars stay on
  the host.  `override_color` is accepted as precomputed colours (the reference declares it and ignores it).
"""
import math

import torch

from .rasterizer import GaussianRasterizationSettings, GaussianRasterizer

from .synthetic import ZFAR, ZNEAR, projection_matrix


def get_projection_matrix(znear, zfar, fov_x, fov_y):
    """getProjectionMatrix (src/gs/camera.cu:59-82) as the TENSOR the reference builds: Eigen stores P column-major and
    the reference wraps that memory as a row-major tensor, i.e. P transposed."""
    return torch.from_numpy(projection_matrix(znear, zfar, fov_x, fov_y).T.copy())


class Camera:
    """R: camera-to-world rotation [3,3], T: camera position [3] (the reference's _R, _T), FoVs in radians."""

    def __init__(self, R, T, FoVx, FoVy, image_width, image_height, device="cuda", uid=0, image_name=""):
        R = torch.as_tensor(R, dtype=torch.float32).cpu()
        T = torch.as_tensor(T, dtype=torch.float32).cpu()
        self._R, self._T, self._FoVx, self._FoVy = R, T, float(FoVx), float(FoVy)
        self._image_width, self._image_height = int(image_width), int(image_height)
        self._uid, self._image_name = uid, image_name
        Tcw = torch.eye(4, dtype=torch.float32)
        Tcw[:3, :3] = R.t()
        Tcw[:3, 3] = -(R.t() @ T)
        view = Tcw.t().contiguous()  # column-major Eigen memory read as a row-major tensor (camera.cu:39-40)
        proj = get_projection_matrix(ZNEAR, ZFAR, self._FoVx, self._FoVy)
        self._world_view_transform = view.to(device)
        self._projection_matrix = proj.to(device)
        self._full_proj_transform = (view @ proj).to(device)
        self._camera_center = torch.linalg.inv(view)[3, :3].contiguous().to(device)

    def Get_image_height(self): return self._image_height
    def Get_image_width(self): return self._image_width
    def Get_FoVx(self): return self._FoVx
    def Get_FoVy(self): return self._FoVy
    def Get_uid(self): return self._uid
    def Get_image_name(self): return self._image_name
    def Get_R(self): return self._R
    def Get_T(self): return self._T
    def Get_world_view_transform(self): return self._world_view_transform
    def Get_projection_matrix(self): return self._projection_matrix
    def Get_full_proj_transform(self): return self._full_proj_transform
    def Get_camera_center(self): return self._camera_center


def render(viewpoint_camera, gaussian_model, bg_color, scaling_modifier=1.0, override_color=None):
    """-> (color [3,H,W], depth [1,H,W], depth_acc [1,H,W]); render_utils.cuh:13-56."""
    cam = viewpoint_camera
    dev = gaussian_model.Get_xyz().device
    settings = GaussianRasterizationSettings(
        image_height=int(cam.Get_image_height()), image_width=int(cam.Get_image_width()),
        tanfovx=math.tan(cam.Get_FoVx() * 0.5), tanfovy=math.tan(cam.Get_FoVy() * 0.5),
        bg=bg_color.to(dev), scale_modifier=float(scaling_modifier),
        viewmatrix=cam.Get_world_view_transform(), projmatrix=cam.Get_full_proj_transform(),
        sh_degree=int(gaussian_model.Get_max_sh_degree()), camera_center=cam.Get_camera_center(), prefiltered=False)
    rasterizer = GaussianRasterizer(settings)
    if hasattr(gaussian_model, "activated"):  # one fused launch for all five getters
        means3D, opacity, scales, rotations, shs = gaussian_model.activated()
    else:
        means3D, opacity = gaussian_model.Get_xyz(), gaussian_model.Get_opacity()
        scales, rotations, shs = gaussian_model.Get_scaling(), gaussian_model.Get_rotation(), gaussian_model.Get_features()
    means2D = torch.zeros_like(means3D, requires_grad=True)
    if override_color is not None and override_color.numel():
        color, radii, depth, depth_acc = rasterizer(means3D, means2D, opacity, colors_precomp=override_color,
                                                    scales=scales, rotations=rotations)
    else:
        color, radii, depth, depth_acc = rasterizer(means3D, means2D, opacity, shs=shs, scales=scales,
                                                    rotations=rotations)
    return color, depth, depth_acc
