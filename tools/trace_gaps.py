"""Per-step timeline from a rocprofv3 --kernel-trace csv: kernels of the last full step in start order with the idle gap
before each (GPU time with nothing running on any queue), to see where a step's time goes between kernels.
    python tools/trace_gaps.py <dir with *_kernel_trace.csv>"""
import csv
import glob
import os
import sys

path = glob.glob(os.path.join(sys.argv[1], "**", "*kernel_trace.csv"), recursive=True)[0]
rows = list(csv.DictReader(open(path)))
ev = sorted(((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"]) for r in rows), key=lambda e: e[0])
# steps are delimited by k_model_step
ends = [i for i, e in enumerate(ev) if "k_model_step" in e[2]]
a, b = ends[-3] + 1, ends[-2] + 1  # one steady-state step
step = ev[a:b]
t_prev_end = ev[a - 1][1]
busy_until = t_prev_end
tot_gap = 0
for s, e, n in step:
    gap = max(0, s - busy_until)
    tot_gap += gap
    short = n.split("(")[0].replace("void gsr::", "").replace("gsr::", "")[:60]
    print("%8.1f us  gap %6.1f  dur %7.1f  %s" % ((s - t_prev_end) / 1e3, gap / 1e3, (e - s) / 1e3, short))
    busy_until = max(busy_until, e)
print("step %.1f us, idle gaps %.1f us" % ((step[-1][1] - t_prev_end) / 1e3, tot_gap / 1e3))
