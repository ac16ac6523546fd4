#!/bin/bash
# run-to-run spread of the default bench at two step counts: bash tools/bench_spread.sh [runs=4]
runs=${1:-4}
for k in 20 100; do
  for i in $(seq 1 $runs); do
    v=$(python3 bench.py --no-cpu-baseline --steps $k 2>/dev/null | python3 -c "import sys,json; print(json.loads(sys.stdin.read().strip().splitlines()[-1])['ms_per_step'])")
    echo "steps $k run $i: $v"
  done
done
