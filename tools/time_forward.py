"""Forward-only kernel timing at a BASELINE config (library event profiler).  usage: time_forward.py [C3] [iters]"""
import sys, os
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch
import gs_livm_amd as G
from gs_livm_amd import synthetic as S
from helpers import hip_forward
name = sys.argv[1] if len(sys.argv) > 1 else "C3"
P, W, H, seed = S.CONFIGS[name]
sc = S.make_scene(P, W, H, seed)
dev = torch.device("cuda:0")
for _ in range(3):
    t, fwd = hip_forward(sc, dev, debug=False)
torch.cuda.synchronize()
G.profile_enable(True)
n = int(sys.argv[2]) if len(sys.argv) > 2 else 5
for _ in range(n):
    t, fwd = hip_forward(sc, dev, debug=False)
torch.cuda.synchronize()
G.profile_enable(False)
for k, (ms, c) in sorted(G.profile_read().items(), key=lambda kv: -kv[1][0]):
    if c: print("  %-26s %8.4f ms  x%d" % (k, ms / n, c // n))
