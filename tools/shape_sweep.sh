#!/bin/bash
# The other BASELINE shapes on the current binary, one bench line each -> gpurun_out/<tag>_shape_sweep.jsonl
#   bash tools/shape_sweep.sh r02c
set -e -o pipefail
tag=${1:?tag}
out=gpurun_out/${tag}_shape_sweep.jsonl
: > "$out"
for flags in "--workload C1" "--workload C2" "--workload C2 --forward-only" "--workload C5shape" \
             "--workload C5shape --loss photometric" "--workload C3 --sh-degree 3" "--workload C3 --loss photometric" \
             "--workload C3"; do
  python3 bench.py $flags --no-cpu-baseline >> "$out" 2>> gpurun_out/${tag}_shape_sweep.err
  echo "[shape_sweep] $flags done"
done
python3 - "$out" <<'PY'
import json, sys
for line in open(sys.argv[1]):
    d = json.loads(line)
    ws = d["workload_stats"]
    print("%-60s %.4f ms/step  R=%d near_far=%s" % (d["config"]["workload"][:60] + (" +photo" if d["config"]["loss"] != "seeded" else ""),
                                                 d["ms_per_step"], ws["R"], ws.get("near_far")))
PY
