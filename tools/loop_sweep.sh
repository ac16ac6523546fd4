#!/bin/bash
# The reference's loop shapes on one MI355X (run through gpurun from the repo root): bash tools/loop_sweep.sh r03a
#   -> gpurun_out/loop_<tag>/<tag>_loop_sweep.jsonl, one bench.py line per configuration:
#   single fixed view (BASELINE C3), K = 3 and K = 8 different views per optimiser iteration (one backward, one Adam
#   step; lioOptimization.cpp:1691-1737, 1822-1832) from one thread and from 2 / 4 rendering threads on streams of their
#   own (gs_livm_amd.multiview.ViewThreads: the reference renders from four threads), a camera that changes every step,
#   K = 4 with the map growing every
#   third step (gaussian.cu:241-313), K = 8 under the photometric loss, and a scene that never saturates (opacities x 0.1:
#   the one-chain floor of the forward).
set -e -o pipefail
tag=${1:?tag}
out=gpurun_out/loop_$tag
mkdir -p "$out"
f="$out/${tag}_loop_sweep.jsonl"
: > "$f"
run() { python3 bench.py --no-cpu-baseline "$@" >> "$f" 2>> "$out/err.log"; echo "[loop_sweep] $* done"; }
run
run --views-per-step 3
run --views-per-step 8 --per-step
run --views-per-step 8 --view-threads 2
run --views-per-step 8 --view-threads 4
run --views-per-step 3 --view-threads 3
run --rotate-views --per-step
run --views-per-step 4 --grow-every 3 --per-step
run --views-per-step 8 --loss photometric
run --views-per-step 8 --view-threads 4 --loss photometric
run --opacity-scale 0.1
run --workload C5shape --views-per-step 8 --loss photometric
python3 tools/print_bench.py "$f" || true
