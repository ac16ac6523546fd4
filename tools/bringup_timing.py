import sys, time
import numpy as np, torch
import os; ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import gs_livm_amd as G
from gs_livm_amd import synthetic as S
from oracle import oracle as O
from helpers import hip_forward, hip_backward
dev = torch.device("cuda:0")
for (P, W, H, seed, D) in [(20000, 333, 257, 5, 3), (60000, 640, 480, 7, 1)]:
    sc = S.make_scene(P, W, H, seed, sh_degree=D)
    fr = O.forward(sc)
    t, fwd = hip_forward(sc, dev)
    col = fwd[1].cpu().numpy()
    err = np.abs(col - fr.out_color).max(0)
    bad = err > 1e-4
    print("P=%d bad pixels %d, of which fragile %d; fragile total %d; max err non-fragile %.3e" % (P, bad.sum(), (bad & (fr.fragile > 0)).sum(), fr.fragile.sum(), err[fr.fragile == 0].max()), flush=True)
# timing at C2 / C3
for name in ("C2", "C3"):
    P, W, H, seed = S.CONFIGS[name]
    sc = S.make_scene(P, W, H, seed)
    t, fwd = hip_forward(sc, dev, debug=False)
    dcol, dacc = S.make_upstream_grads(W, H, seed)
    torch.cuda.synchronize()
    print(name, "R =", fwd[0], "P_vis =", int((fwd[4] > 0).sum()), flush=True)
    v = G.state_views(fwd[5], fwd[6], fwd[7], P, fwd[0], W, H)
    rng = v["ranges"].cpu().numpy().astype(np.int64); ln = rng[:, 1] - rng[:, 0]
    print("  tile list len mean %.1f max %d ; n_contrib mean %.1f max %d" % (ln.mean(), ln.max(), v["n_contrib"].float().mean().item(), v["n_contrib"].max().item()))
    dc = torch.from_numpy(dcol).to(dev); da = torch.from_numpy(dacc).to(dev)
    def fw():
        return G.rasterize_forward(t["bg"], t["means3D"], t["colors_precomp"], t["opacities"], t["scales"], t["rotations"], 1.0, t["cov3D_precomp"], t["viewmatrix"], t["projmatrix"], sc["tanfovx"], sc["tanfovy"], H, W, t["shs"], 0, t["campos"], False, False)
    def bw(f):
        return G.rasterize_backward(t["bg"], t["means3D"], f[4], t["colors_precomp"], t["scales"], t["rotations"], 1.0, t["cov3D_precomp"], t["viewmatrix"], t["projmatrix"], sc["tanfovx"], sc["tanfovy"], dc, da, t["shs"], 0, t["campos"], f[5], f[0], f[6], f[7], False)
    for it in range(3):
        torch.cuda.synchronize(); t0 = time.time(); f = fw(); torch.cuda.synchronize(); t1 = time.time(); g = bw(f); torch.cuda.synchronize(); t2 = time.time()
        print("  %s iter %d: fwd %.2f ms  bwd %.2f ms" % (name, it, (t1 - t0) * 1e3, (t2 - t1) * 1e3), flush=True)
    # determinism of backward
    g2 = bw(f); torch.cuda.synchronize()
    print("  backward bitwise reproducible:", all(torch.equal(a, b) for a, b in zip(g, g2)))
print("DONE")
