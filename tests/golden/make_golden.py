"""Generates the committed fixtures under tests/golden/.

WHAT THESE ARE.  The reference (CUDA) cannot be built or run in the build image and ships no
tests or golden vectors for the rasterizer, so:
  * survey_single_gaussian.json -- the ONE reference output on record: the single-Gaussian known
    answer the survey obtained from the reference's own kernel bodies.
    Hand-transcribed, not produced by this script.
  * oracle_*.npz -- inputs and every intermediate / output / gradient of oracle/gsr_oracle.c on small
    seeded scenes.  They pin the oracle against accidental change and let the GPU parity tests run
    against fixed vectors.  They are oracle output, NOT reference output: parity is "unpinned"
    beyond the survey's known answer.

Run from the repo root:  python tests/golden/make_golden.py
"""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))

from gs_livm_amd import synthetic as S  # noqa: E402
from oracle import oracle as O  # noqa: E402

CASES = {  # name -> (P, W, H, seed, sh_degree)
    "oracle_P300_70x50_D3": (300, 70, 50, 11, 3),
    "oracle_P2000_160x96_D0": (2000, 160, 96, 12, 0),
}


def make(name):
    P, W, H, seed, D = CASES[name]
    sc = S.make_scene(P, W, H, seed, sh_degree=D)
    O.set_threads(1)  # deterministic accumulation order for the backward
    fr = O.forward(sc)
    dcol, dacc = S.make_upstream_grads(W, H, seed)
    g = O.backward(fr, sc, dcol, dacc)
    out = {"meta": np.array([P, W, H, seed, D, fr.R], np.int64), "dL_dcolor_in": dcol, "dL_dacc_in": dacc}
    for k in ("means3D", "scales", "rotations", "opacities", "shs", "viewmatrix", "projmatrix", "campos", "bg"):
        out["in_" + k] = sc[k]
    out["in_tanfov"] = np.array([sc["tanfovx"], sc["tanfovy"]], np.float32)
    for k in ("radii", "means2D", "depths", "cov3D", "rgb", "conic_opacity", "tiles_touched", "point_offsets",
              "clamped", "keys", "point_list", "ranges", "final_T", "n_contrib", "out_color", "out_depth", "out_acc",
              "fragile"):
        out["fw_" + k] = getattr(fr, k)
    for k, v in g.items():
        out["bw_" + k] = v
    # integer stages under the product's tile culling (oracle tight mode): same images/gradients, shorter lists
    ft = O.forward(sc, tight=True)
    for k in ("out_color", "out_depth", "out_acc", "final_T", "radii"):
        assert np.array_equal(getattr(ft, k), getattr(fr, k)), k
    gt = O.backward(ft, sc, dcol, dacc)
    assert all(np.array_equal(gt[k], g[k]) for k in g)
    out["meta_tight"] = np.array([ft.R], np.int64)
    for k in ("tiles_touched", "point_offsets", "keys", "point_list", "ranges", "n_contrib"):
        out["tw_" + k] = getattr(ft, k)
    path = os.path.join(HERE, name + ".npz")
    if os.path.exists(path):  # the reference-mode vectors are pinned: regenerating must not change them
        old = np.load(path)
        for k in old.files:
            if k in out and not k.startswith("tw_") and k != "meta_tight":
                assert np.array_equal(old[k], out[k]), "fixture field changed: " + k
    np.savez_compressed(os.path.join(HERE, name + ".npz"), **out)
    print(name, "R =", fr.R, "visible =", int((fr.radii > 0).sum()))


if __name__ == "__main__":
    for n in CASES:
        make(n)
