"""PLY export / import in the reference's format (SURVEY.md section 8(f) "next" row 4).

GaussianModel::Save_ply (src/gs/gaussian.cu:494-522) copies seven tensors to the host one by one and lets tinyply
interleave them on a CPU thread (Write_output_ply, :542-573).  Here the rows are interleaved on the device
(csrc/growth.hip, k_pack_ply_rows), cross PCIe as ONE contiguous copy and are written behind a header that is
byte-identical to tinyply's (pinned by tests/golden/ply_*.ply, which the reference's vendored tinyply produced).

File layout: "ply / format binary_little_endian 1.0 / element vertex P / property float <name> ... / end_header",
then P rows of 14 + 3M little-endian f32 in the order of construct_list_of_attributes (:474-492):
  x y z  nx ny nz  f_dc_0..2  f_rest_0..3(M-1)-1  opacity  scale_0..2  rot_0..3
"""
import os

import numpy as np
import torch

from . import _capi


def attribute_names(M):
    """construct_list_of_attributes (src/gs/gaussian.cu:474-492) for M SH coefficients per channel."""
    names = ["x", "y", "z", "nx", "ny", "nz"]
    names += ["f_dc_%d" % i for i in range(3)]
    names += ["f_rest_%d" % i for i in range(3 * (M - 1))]
    names += ["opacity"] + ["scale_%d" % i for i in range(3)] + ["rot_%d" % i for i in range(4)]
    return names


def header_bytes(P, M):
    lines = ["ply", "format binary_little_endian 1.0", "element vertex %d" % P]
    lines += ["property float " + n for n in attribute_names(M)]
    lines += ["end_header"]
    return ("\n".join(lines) + "\n").encode("ascii")


def rows_numpy(xyz, features_dc, features_rest, opacity, scaling, rotation):
    """Host restatement of the row layout (numpy): what k_pack_ply_rows is checked against."""
    P = xyz.shape[0]
    f_dc = np.ascontiguousarray(np.transpose(features_dc, (0, 2, 1))).reshape(P, -1)       # transpose(1,2).flatten(1)
    f_rest = np.ascontiguousarray(np.transpose(features_rest, (0, 2, 1))).reshape(P, -1)
    cols = [xyz, np.zeros_like(xyz), f_dc, f_rest, opacity.reshape(P, 1), scaling, rotation]
    return np.ascontiguousarray(np.concatenate([np.asarray(c, np.float32).reshape(P, -1) for c in cols], 1))


def ply_bytes(rows, M):
    """Header + rows -> the file content (rows: [P, 14 + 3M] float32 numpy array)."""
    rows = np.ascontiguousarray(rows, dtype="<f4")
    assert rows.ndim == 2 and rows.shape[1] == 14 + 3 * M
    return header_bytes(rows.shape[0], M) + rows.tobytes()


def save_ply(folder, model, iteration=0):
    """GaussianModel::Save_ply: writes <folder>/point_cloud/iteration_<n>/point_cloud.ply; returns the path.
    `model` exposes the six leaves as _xyz, _features_dc, _features_rest, _opacity, _scaling, _rotation."""
    d = os.path.join(str(folder), "point_cloud", "iteration_%d" % iteration)
    os.makedirs(d, exist_ok=True)
    M = 1 + int(model._features_rest.size(1))
    with torch.no_grad():
        rows = _capi.pack_ply_rows(model._xyz, model._features_dc, model._features_rest, model._opacity,
                                   model._scaling, model._rotation)
    host = rows.cpu().numpy()  # one D2H copy
    path = os.path.join(d, "point_cloud.ply")
    with open(path, "wb") as fh:
        fh.write(ply_bytes(host, M))
    return path


def load_ply(path):
    """Inverse of save_ply: dict of numpy arrays in the leaf shapes (xyz, features_dc [P,1,3],
    features_rest [P,M-1,3], opacity [P,1], scaling, rotation).  Accepts exactly the reference's layout."""
    with open(path, "rb") as fh:
        blob = fh.read()
    end = blob.index(b"end_header\n") + len(b"end_header\n")
    head = blob[:end].decode("ascii").split("\n")
    if head[0] != "ply" or head[1] != "format binary_little_endian 1.0":
        raise ValueError("not a binary little-endian PLY")
    P = int(head[2].split()[2])
    props = [ln.split()[2] for ln in head[3:] if ln.startswith("property float ")]
    n_rest = sum(1 for p in props if p.startswith("f_rest_"))
    M = 1 + n_rest // 3
    if props != attribute_names(M):
        raise ValueError("unexpected property list")
    rows = np.frombuffer(blob, dtype="<f4", count=P * len(props), offset=end).reshape(P, len(props))
    o = 9 + n_rest
    return dict(xyz=rows[:, 0:3].copy(),
                features_dc=rows[:, 6:9].reshape(P, 3, 1).transpose(0, 2, 1).copy(),
                features_rest=rows[:, 9:o].reshape(P, 3, M - 1).transpose(0, 2, 1).copy(),
                opacity=rows[:, o:o + 1].copy(), scaling=rows[:, o + 1:o + 4].copy(),
                rotation=rows[:, o + 4:o + 8].copy())
