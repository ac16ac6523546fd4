#!/bin/bash
# Regenerates the per-round evidence under profiles/ on an MI355X box (run through gpurun from the repo root):
#   bash tools/profile_round.sh r01i
# 1. rocprofv3 --kernel-trace --stats of the bench command  -> <tag>_kernel_stats.csv, <tag>_bench_under_rocprof.json
# 2. FETCH_SIZE / WRITE_SIZE in two separate --pmc passes    -> <tag>_pmc_traffic.json   (tools/pmc_summary.py)
# 3. SQ counters in their own --pmc pass                      -> <tag>_sq_counters.json   (tools/sq_summary.py)
# 4. the clean bench line with the CPU-oracle leg            -> <tag>_bench.json (it cites 2. and 3. of THIS run)
# Counter passes carry --kernel-trace only (never a sys/hip/hsa trace); python3 comes directly after `--`.
# They run with GSR_ASYNC_FAR=0: counter collection serialises the dispatches of all queues, and a far chain parked behind
# a stream-side wait (asynchronous near/far frames) then never sees the near blend it waits for -- the run hangs.  The
# counters of the kernels themselves do not depend on which stream the far chain was enqueued on.
set -e -o pipefail
tag=${1:?tag}
out=gpurun_out/prof_$tag
rm -rf "$out"; mkdir -p "$out"
root=$(pwd)
cd /tmp && export TMPDIR=/tmp && cd "$root"
rocprofv3 --kernel-trace --stats --output-format csv -d "$out/stats" -- python3 bench.py --steps 20 --warmup 5 --no-cpu-baseline \
    > "$out/${tag}_bench_under_rocprof.json" 2> "$out/stats.err"
cp "$(find "$out/stats" -name '*kernel_stats.csv' | head -1)" "$out/${tag}_kernel_stats.csv"
echo "[profile_round] stats done"
# the counter passes come BEFORE the clean bench line: bench.py prints the committed summaries' figures
# (profiles/*_latest.json, with their source) and must find THIS build's, not the previous round's
(
export GSR_ASYNC_FAR=0
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d "$out/pmc_fetch" -- python3 bench.py --steps 3 --warmup 1 --no-cpu-baseline \
    > /dev/null 2> "$out/pmc_fetch.err"
rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d "$out/pmc_write" -- python3 bench.py --steps 3 --warmup 1 --no-cpu-baseline \
    > /dev/null 2> "$out/pmc_write.err"
python3 tools/pmc_summary.py "$out/pmc_fetch" "$out/pmc_write" "$out/${tag}_pmc_traffic.json" workload=C3 sh_degree=0 tag=$tag
echo "[profile_round] traffic done"
rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_ACTIVE_INST_VALU SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_BUSY_CYCLES \
    --output-format csv -d "$out/pmc_sq" -- python3 bench.py --steps 3 --warmup 1 --no-cpu-baseline > /dev/null 2> "$out/pmc_sq.err"
python3 tools/sq_summary.py "$out/pmc_sq" "$out/${tag}_sq_counters.json" workload=C3 sh_degree=0 tag=$tag
echo "[profile_round] sq done"
)
cp "$out/${tag}_pmc_traffic.json" profiles/pmc_traffic_latest.json
cp "$out/${tag}_sq_counters.json" profiles/sq_counters_latest.json
python3 bench.py --steps 20 --warmup 5 > "$out/${tag}_bench.json" 2> "$out/bench.err"
echo "[profile_round] clean bench done"
rm -rf "$out/stats" "$out/pmc_fetch" "$out/pmc_write" "$out/pmc_sq"
ls -la "$out"
