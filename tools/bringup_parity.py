"""First bring-up script (not a test): stage-by-stage parity of the HIP path vs the oracle."""
import sys, time
import numpy as np, torch
import os; ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import gs_livm_amd as G
from gs_livm_amd import synthetic as S
from oracle import oracle as O
from helpers import hip_forward, hip_backward

dev = torch.device("cuda:0")
for (P, W, H, seed, D) in [(300, 70, 50, 11, 3), (10000, 640, 480, 1, 0), (20000, 333, 257, 5, 3)]:
    sc = S.make_scene(P, W, H, seed, sh_degree=D)
    fr = O.forward(sc)
    t, fwd = hip_forward(sc, dev)
    R = fwd[0]
    print("== P=%d %dx%d D=%d  R oracle=%d hip=%d" % (P, W, H, D, fr.R, R), flush=True)
    v = G.state_views(fwd[5], fwd[6], fwd[7], P, R, W, H)
    radii = fwd[4].cpu().numpy()
    print(" radii equal:", np.array_equal(radii, fr.radii), " tiles:", np.array_equal(v["tiles_touched"].cpu().numpy().view(np.uint32), fr.tiles_touched),
          " offsets:", np.array_equal(v["point_offsets"].cpu().numpy().view(np.uint32), fr.point_offsets))
    vis = fr.radii > 0
    sp = v["splats"].cpu().numpy()
    print(" means2D:", np.array_equal(sp[vis, 0:2], fr.means2D[vis]), " depth:", np.array_equal(sp[vis, 9], fr.depths[vis]),
          " conic:", np.array_equal(sp[vis][:, [2, 3, 4]], fr.conic_opacity[vis][:, :3]), " rgb maxerr:", np.abs(sp[vis, 6:9] - fr.rgb[vis]).max() if vis.any() else 0)
    if R == fr.R and R > 0:
        keys = v["keys"].cpu().numpy().view(np.uint64); pl = v["point_list"].cpu().numpy().view(np.uint32)
        print(" keys:", np.array_equal(keys, fr.keys), " point_list:", np.array_equal(pl, fr.point_list),
              " ranges:", np.array_equal(v["ranges"].cpu().numpy().view(np.uint32), fr.ranges))
    col = fwd[1].cpu().numpy(); dep = fwd[2].cpu().numpy(); acc = fwd[3].cpu().numpy()
    print(" color maxerr %.3e depth %.3e acc %.3e" % (np.abs(col - fr.out_color).max(), np.abs(dep - fr.out_depth).max(), np.abs(acc - fr.out_acc).max()))
    nc = v["n_contrib"].cpu().numpy().view(np.uint32)
    print(" n_contrib mismatches: %d / %d ; final_T maxerr %.3e" % ((nc != fr.n_contrib).sum(), nc.size, np.abs(v["final_T"].cpu().numpy() - fr.final_T).max()))
    dcol, dacc = S.make_upstream_grads(W, H, seed)
    go = O.backward(fr, sc, dcol, dacc)
    gh = hip_backward(sc, t, fwd, dcol, dacc, dev)
    for k in go:
        a, b = gh[k].reshape(-1), go[k].reshape(-1)
        sc_ = np.abs(b).max() + 1e-30
        print("  %-14s max|ref| %.3e  maxerr/max %.3e" % (k, sc_, np.abs(a - b).max() / sc_))
print("DONE")
