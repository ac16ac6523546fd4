#!/bin/bash
# Long runs (the synthetic optimiser makes the scene more transparent step by step: tiles stop saturating inside a small
# budget) with the adaptive near budget and with fixed ones:  bash tools/drift_sweep.sh [steps=100]
steps=${1:-100}
for ne in 0 320 480 640 800 1000; do
  python3 bench.py --no-cpu-baseline --steps $steps --near-entries $ne 2>/dev/null | python3 -c "
import sys, json
d = json.loads(sys.stdin.read().strip().splitlines()[-1]); sp = d['speculation']
print('near entries %4s: %.4f ms/step  misses %d  skips %d  scale_q8 %d' % ('$ne' if $ne else 'auto', d['ms_per_step'], sp['far_skip_misses'], sp['far_skips'], sp['near_budget_scale_q8']))"
done
