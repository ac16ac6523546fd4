"""Timeline of one bench step from a rocprofv3 --kernel-trace CSV: per kernel (in launch order) its duration and
the idle gap on the GPU before it.  usage: python tools/gap_analysis.py <*_kernel_trace.csv> [steps_from_end]"""
import csv, re, sys
rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
def short(n):
    n = re.sub(r"^void ", "", n); n = re.sub(r"\(.*$", "", n); n = re.sub(r"<.*$", "", n)
    return n.split("::")[-1]
names = [short(r["Kernel_Name"]) for r in rows]
# a step starts at k_activate (or k_preprocess when the optimiser is off); take the last complete steps
starts = [i for i, n in enumerate(names) if n == "k_activate"]
if len(starts) < 3:  # one-kernel optimiser tail: k_activate only runs once, before the first step
    starts = [i for i, n in enumerate(names) if n == "k_preprocess"]
nsteps = int(sys.argv[2]) if len(sys.argv) > 2 else 10
starts = starts[-(nsteps + 1):]
agg = {}
order = []
tot_busy = tot_gap = 0.0
for a, b in zip(starts[:-1], starts[1:]):
    pos = {}
    for i in range(a, b):
        n = names[i]; pos[n] = pos.get(n, 0) + 1; key = "%s#%d" % (n, pos[n])
        dur = (int(rows[i]["End_Timestamp"]) - int(rows[i]["Start_Timestamp"])) / 1e3
        gap = (int(rows[i]["Start_Timestamp"]) - int(rows[i - 1]["End_Timestamp"])) / 1e3 if i else 0.0
        if key not in agg: agg[key] = [0.0, 0.0, 0]; order.append(key)
        agg[key][0] += dur; agg[key][1] += gap; agg[key][2] += 1
        tot_busy += dur; tot_gap += gap
n = len(starts) - 1
print("%d steps; per step: busy %.1f us, idle %.1f us, %d launches" % (n, tot_busy / n, tot_gap / n, len(order)))
for k in order:
    d, g, c = agg[k]
    print("  %-34s dur %8.1f us   gap before %7.1f us" % (k, d / c, g / c))
