import sys
import numpy as np, torch
import os; ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import gs_livm_amd as G
from gs_livm_amd import synthetic as S
from oracle import oracle as O
from helpers import hip_forward, hip_backward
from test_gpu_parity import masked_grads
dev = torch.device("cuda:0")
P, W, H, seed, D = 40000, 500, 300, 6, 1
sc = S.make_scene(P, W, H, seed, sh_degree=D)
fr = O.forward(sc); t, fwd = hip_forward(sc, dev)
dcol, dacc = masked_grads(W, H, seed, fr.fragile)
O.set_threads(1); ref = O.backward(fr, sc, dcol, dacc); got = hip_backward(sc, t, fwd, dcol, dacc, dev)
for k in ("dL_dcov3D", "dL_dscales", "dL_drotations", "dL_dmeans3D"):
    a, b = got[k].reshape(P, -1), ref[k].reshape(P, -1)
    scale = np.abs(b).max(); tol = 1e-5 * scale + 1e-4 * np.abs(b)
    bad = np.argwhere(np.abs(a - b) > tol)
    print(k, "bad", len(bad), "scale", scale)
    for (i, j) in bad[:5]:
        print("  row", i, "col", j, "got", a[i], "ref", b[i])
        print("   conic got", got["dL_dconic"][i].ravel(), "ref", ref["dL_dconic"][i].ravel(), "radius", fr.radii[i], "tiles", fr.tiles_touched[i])
        print("   conic_opacity", fr.conic_opacity[i], "cov3D", fr.cov3D[i])
