"""Soak run of the randomised parity check of tests/test_gpu_parity.py with fresh generator seeds (GPU):
    python tools/soak_random_scenes.py [first_seed] [n_seeds] [cases_per_seed]
Every case goes through the full forward + backward comparison against the CPU oracle; the first failure aborts.
SOAK_NEAR_FAR=1: the randomised near/far-vs-one-chain equivalence check instead, e.g. with GSR_PRE_HIST_MIN_P=0
(partial depth sort on these small scenes) and / or GSR_ASYNC_FAR=0 (host-decided far-chain speculation)."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import torch

import test_gpu_parity as T

first = int(sys.argv[1]) if len(sys.argv) > 1 else 1
n = int(sys.argv[2]) if len(sys.argv) > 2 else 5
cases = int(sys.argv[3]) if len(sys.argv) > 3 else 8
dev = torch.device("cuda:0")
near_far = os.environ.get("SOAK_NEAR_FAR") is not None  # the near/far equivalence check instead (any GSR_* knob applies)
for seed in range(first, first + n):
    (T._random_near_far if near_far else T._random_scenes)(np.random.default_rng(seed), cases, dev)
    print("generator seed %d: %d cases ok" % (seed, cases), flush=True)
