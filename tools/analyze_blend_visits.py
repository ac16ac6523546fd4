"""Where do the blend backward's (quad, splat) visits go?  Statistics of the walked tile lists of a workload, computed
with plain torch on the GPU from the forward's own state (ranges, point_list, splat records, n_contrib, quad_last):

  * (tile, entry) instances the backward walks, (quad, entry) pairs with at least one contributing pixel,
  * lane utilisation of those visits, distribution of quads hit per instance,
  * what coarser work units (two quads per wave: 8x16 / 16x8 halves; one wave per tile) would visit.

    python tools/analyze_blend_visits.py [C3|C2|C1|C5shape]      (run on the GPU box)
Design input for render.hip; not part of the product or the tests.
"""
import json
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import gs_livm_amd as G  # noqa: E402
from gs_livm_amd import synthetic as S  # noqa: E402
from helpers import hip_forward  # noqa: E402

dev = torch.device("cuda:0")
name = sys.argv[1] if len(sys.argv) > 1 else "C3"
P, W, H, seed = S.CONFIGS[name]
sc = S.make_scene(P, W, H, seed)
t, fwd = hip_forward(sc, dev, debug=False)
R = fwd[0]
v = G.state_views(fwd[5], fwd[6], fwd[7], P, R, W, H)
gx, gy = (W + 15) // 16, (H + 15) // 16
ranges = v["ranges"].long()
ql = v["quad_last"].long()
ncontrib = torch.zeros((gy * 16, gx * 16), dtype=torch.long, device=dev)
ncontrib[:H, :W] = v["n_contrib"].long()
splats = v["splats"]
pl = v["point_list"].long()

tot = dict(instances=0, quad_visits_any=0, quad_lanes_ok=0, half_h_visits=0, half_v_visits=0, tile_visits=0,
           quad_candidates=0)
hist_quads = np.zeros(5, np.int64)
lane_hist = np.zeros(9, np.int64)  # ok lanes per visit in eighths
jj, ii = torch.meshgrid(torch.arange(16, device=dev), torch.arange(16, device=dev), indexing="ij")
quad_of = ((jj // 8) * 2 + (ii // 8)).reshape(-1)  # [256] quad id of each pixel of a tile
for ty in range(gy):
    for tx in range(gx):
        tile = ty * gx + tx
        n = int(ql[tile].max())
        if n == 0:
            continue
        ids = pl[ranges[tile, 0]:ranges[tile, 0] + n]
        rec = splats[ids]  # [n, 12]
        px = (tx * 16 + ii).reshape(1, -1).float()
        py = (ty * 16 + jj).reshape(1, -1).float()
        dx = rec[:, 0:1] - px
        dy = rec[:, 1:2] - py
        power = -0.5 * (rec[:, 2:3] * dx * dx + rec[:, 4:5] * dy * dy) - rec[:, 3:4] * dx * dy
        alpha = torch.clamp(rec[:, 5:6] * torch.exp(power), max=0.99)
        lastc = ncontrib[ty * 16:ty * 16 + 16, tx * 16:tx * 16 + 16].reshape(1, -1)
        pos = torch.arange(n, device=dev).reshape(-1, 1)
        ok = (pos < lastc) & (power <= 0) & (alpha >= 1.0 / 255.0)  # [n, 256]
        okq = torch.stack([ok[:, quad_of == q].sum(1) for q in range(4)], 1)  # [n, 4] ok lanes per quad
        anyq = okq > 0
        tot["instances"] += n
        tot["quad_candidates"] += int((pos < ql[tile].reshape(1, 4)).sum())
        tot["quad_visits_any"] += int(anyq.sum())
        tot["quad_lanes_ok"] += int(okq.sum())
        tot["half_h_visits"] += int((anyq[:, 0] | anyq[:, 1]).sum() + (anyq[:, 2] | anyq[:, 3]).sum())  # 16x8 halves
        tot["half_v_visits"] += int((anyq[:, 0] | anyq[:, 2]).sum() + (anyq[:, 1] | anyq[:, 3]).sum())  # 8x16 halves
        tot["tile_visits"] += int(anyq.any(1).sum())
        hist_quads += np.bincount(anyq.sum(1).cpu().numpy(), minlength=5)
        lane_hist += np.bincount(((okq[anyq] + 7) // 8).cpu().numpy(), minlength=9)
out = dict(workload=name, R=R, **tot, quads_hit_per_instance_hist=hist_quads.tolist(),
           ok_lanes_per_visit_hist_in_eighths=lane_hist.tolist(),
           lane_utilisation=tot["quad_lanes_ok"] / max(1, 64 * tot["quad_visits_any"]))
print(json.dumps(out))
