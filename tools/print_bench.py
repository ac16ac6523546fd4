"""Prints the headline and the per-kernel table of bench.py JSON lines side by side:  python tools/print_bench.py a.json b.json"""
import json
import sys

runs = [json.loads(open(f).read().strip().splitlines()[-1]) for f in sys.argv[1:]]
names = []
for d in runs:
    for k in d["kernels"]:
        if k not in names:
            names.append(k)
print("%-26s" % "ms_per_step", "  ".join("%9.4f" % d["ms_per_step"] for d in runs))
for k in names:
    print("%-26s" % k, "  ".join("%9.4f" % d["kernels"].get(k, {}).get("ms_per_step", float("nan")) for d in runs))
print("%-26s" % "kernel sum", "  ".join("%9.4f" % d["whole_path"]["kernel_ms_per_step"] for d in runs))
