import sys, os
ROOT='/root/repo'; sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT,'tests'))
import numpy as np, torch
import gs_livm_amd as G
from gs_livm_amd import synthetic as S
from oracle import oracle as O
from helpers import hip_forward, hip_backward
dev=torch.device('cuda:0')
rng = np.random.default_rng(20240611)
for case in range(12):
    W, H = int(rng.integers(17, 700)), int(rng.integers(9, 420))
    P = int(rng.integers(50, 30_000)); D = int(rng.integers(0, 4)); seed = 1000 + case
    g = S.make_gaussians(P, seed, sh_degree=D, fovx_deg=float(rng.uniform(35, 100)), aspect=W / H, zmin=float(rng.uniform(0.3, 2.0)), zmax=float(rng.uniform(3.0, 60.0)))
    cam = S.make_camera(W, H, fovx_deg=float(rng.uniform(35, 100)), yaw_deg=float(rng.uniform(-25, 25)), position=tuple(rng.uniform(-0.5, 0.5, 3)))
    sc = dict(g, **cam, bg=rng.uniform(0, 1, 3).astype(np.float32), colors_precomp=None, cov3D_precomp=None, scale_modifier=float(rng.uniform(0.5, 1.5)))
    sc["scales"] = (sc["scales"] * rng.uniform(0.5, 2.0, (P, 3))).astype(np.float32)
    sc["rotations"] = (sc["rotations"] * rng.uniform(0.5, 2.0, (P, 1))).astype(np.float32)
    sc["opacities"] = rng.uniform(0.0, 1.0, (P, 1)).astype(np.float32) ** float(rng.uniform(0.5, 3.0))
    fr = O.forward(sc, tight=True)
    t, fwd = hip_forward(sc, dev)
    dcol, dacc = S.make_upstream_grads(W, H, seed)
    keep=(fr.fragile==0).astype(np.float32); dcol*=keep[None]; dacc*=keep[None]
    O.set_threads(1)
    ref = O.backward(fr, sc, dcol, dacc); got = hip_backward(sc, t, fwd, dcol, dacc, dev)
    for k in ref:
        r=ref[k].reshape(got[k].shape) if got[k].size else ref[k]; gg=got[k]
        if not r.size: continue
        Pn=r.shape[0]; scale=np.abs(r).max(); rowmax=np.abs(r.reshape(Pn,-1)).max(1).reshape((Pn,)+(1,)*(r.ndim-1))
        tol=1e-5*scale+1e-4*rowmax
        bad=np.abs(gg-r) > tol
        if bad.any():
            print('   frac bad %.2e, worst ratio %.1f' % (bad.mean(), float((np.abs(gg-r)/tol).max())))
            idx=np.argwhere(bad)
            print("case",case,"P",P,W,H,"D",D,k,"bad",bad.sum(),"of",bad.size,"scale",scale)
            for ii in idx[:3]:
                i=ii[0]; print("  row",i,"got",gg[i].ravel(),"ref",r[i].ravel(),"radius",fr.radii[i],"op",sc["opacities"][i],"conic",fr.conic_opacity[i],"tiles",fr.tiles_touched[i])
    print("case",case,"done R",fr.R, "frag", fr.fragile.mean())
