#!/bin/bash
# the configurations of tools/loop_sweep.sh from the sixth on, appended to an existing sweep file (a run that was cut short)
set -e -o pipefail
tag=${1:?tag}
out=gpurun_out/loop_$tag
mkdir -p "$out"
f="$out/${tag}_loop_sweep_rest.jsonl"
: > "$f"
run() { python3 bench.py --no-cpu-baseline "$@" >> "$f" 2>> "$out/err_rest.log"; echo "[loop_sweep] $* done"; }
run --views-per-step 3 --view-threads 3
run --rotate-views --per-step
run --views-per-step 4 --grow-every 3 --per-step
run --views-per-step 8 --loss photometric
run --views-per-step 8 --view-threads 4 --loss photometric
run --opacity-scale 0.1
run --workload C5shape --views-per-step 8 --loss photometric
python3 tools/print_bench.py "$f" || true
