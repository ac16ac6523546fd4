#!/bin/bash
# Extra SQ counter passes on the two blend kernels (who waits for what): bash tools/sq_probe.sh <tag>
# GSR_ASYNC_FAR=0: counter collection serialises all queues.  Output: gpurun_out/<tag>_sqprobe/*.csv + a summary.
set -e -o pipefail
tag=${1:?tag}
out=gpurun_out/${tag}_sqprobe
mkdir -p "$out"
cd /tmp 2>/dev/null && export TMPDIR=/tmp && cd - > /dev/null
export GSR_ASYNC_FAR=0
rocprofv3 --list-avail > "$out/avail.txt" 2>&1 || true
i=0
for set in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_ANY" \
           "SQ_WAIT_INST_LDS SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_INST_CYCLES_SALU SQ_INST_CYCLES_VMEM SQ_WAVE_CYCLES" \
           "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_LDS_ADDR_CONFLICT SQ_INSTS_SMEM SQ_IFETCH SQ_WAVES" \
           "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_BRANCH" \
           "SQ_THREAD_CYCLES_VALU SQ_VALU_MFMA_BUSY_CYCLES SQ_INST_LEVEL_LDS SQ_INST_LEVEL_VMEM SQ_ACCUM_PREV_HIRES SQ_WAVE_CYCLES"; do
  i=$((i + 1))
  rocprofv3 --kernel-trace --pmc $set --output-format csv -d "$out/p$i" -- python3 bench.py --steps 3 --warmup 1 --no-cpu-baseline \
      > /dev/null 2> "$out/p$i.err" || echo "[sq_probe] pass $i failed (see $out/p$i.err)"
  echo "[sq_probe] pass $i done"
done
python3 - "$out" <<'PY'
import csv, glob, sys, collections, os
out = sys.argv[1]
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob(os.path.join(out, "p*", "**", "*counter_collection.csv"), recursive=True):
    for row in csv.DictReader(open(f)):
        k = row.get("Kernel_Name", "")
        if "k_blend" not in k: continue
        name = k.split("(")[0].replace("gsr::", "")[:40]
        acc[name][row["Counter_Name"]].append(float(row["Counter_Value"]))
with open(os.path.join(out, "summary.txt"), "w") as fh:
    for k, cs in sorted(acc.items()):
        print(k, file=fh); print(k)
        for c, v in sorted(cs.items()):
            line = "   %-28s mean %.4g  (n=%d)" % (c, sum(v) / len(v), len(v))
            print(line, file=fh); print(line)
PY
