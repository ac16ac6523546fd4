"""Generates tests/golden/ply_*.{npz,ply}: the reference's PLY export of small seeded models, produced by the
reference's OWN vendored tinyply (oracle/_ref/write_ply, built from /root/reference/external/tinyply by
oracle/Makefile -- build container only).  The .ply files are reference-library output: they pin the product's
exporter (gs-livm_amd/ply.py) byte for byte.  The .npz holds the model tensors the file was written from.

Run from the repo root:  python tests/golden/make_golden_ply.py
"""
import os
import subprocess
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
CASES = {"ply_P7_M4": (7, 4, 101), "ply_P300_M1": (300, 1, 102), "ply_P33_M16": (33, 16, 103)}


def model(P, M, seed):
    r = np.random.default_rng(seed)
    f = lambda *s: r.standard_normal(s).astype(np.float32)  # noqa: E731
    return dict(xyz=f(P, 3), features_dc=f(P, 1, 3), features_rest=f(P, M - 1, 3), opacity=f(P, 1),
                scaling=f(P, 3), rotation=f(P, 4))


def reference_columns(m):
    """The seven tensors Save_ply hands to Write_output_ply (src/gs/gaussian.cu:498-509)."""
    P = m["xyz"].shape[0]
    f_dc = np.ascontiguousarray(m["features_dc"].transpose(0, 2, 1)).reshape(P, -1)
    f_rest = np.ascontiguousarray(m["features_rest"].transpose(0, 2, 1)).reshape(P, -1)
    return [m["xyz"], np.zeros_like(m["xyz"]), f_dc, f_rest, m["opacity"], m["scaling"], m["rotation"]]


def main():
    exe = os.path.join(ROOT, "oracle", "_ref", "write_ply")
    subprocess.check_call(["make", "-C", os.path.join(ROOT, "oracle"), "ref"], stdout=subprocess.DEVNULL)
    if not os.path.exists(exe):
        sys.exit("oracle/_ref/write_ply not built (needs /root/reference)")
    for name, (P, M, seed) in CASES.items():
        m = model(P, M, seed)
        raw = os.path.join(HERE, name + ".f32.tmp")
        with open(raw, "wb") as fh:
            for a in reference_columns(m):
                fh.write(np.ascontiguousarray(a, np.float32).tobytes())
        out = os.path.join(HERE, name + ".ply")
        subprocess.check_call([exe, str(P), str(M), raw, out])
        os.remove(raw)
        np.savez_compressed(os.path.join(HERE, name + ".npz"), **m)
        print(name, os.path.getsize(out), "bytes")


if __name__ == "__main__":
    main()
