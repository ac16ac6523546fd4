#!/bin/bash
# A/B of an environment knob on the default bench, alternating runs on one box:  bash tools/ab_env.sh KNOB=VALUE [runs=4] [bench flags...]
kv=${1:?KNOB=VALUE}; runs=${2:-4}; shift; shift
for i in $(seq 1 $runs); do
  a=$(python3 bench.py --no-cpu-baseline "$@" 2>/dev/null | python3 -c "import sys,json; print(json.loads(sys.stdin.read().strip().splitlines()[-1])['ms_per_step'])")
  b=$(env $kv python3 bench.py --no-cpu-baseline "$@" 2>/dev/null | python3 -c "import sys,json; print(json.loads(sys.stdin.read().strip().splitlines()[-1])['ms_per_step'])")
  echo "run $i: default $a   $kv $b"
done
