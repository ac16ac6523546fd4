#!/usr/bin/env python3
"""Turns two rocprofv3 --pmc runs (FETCH_SIZE and WRITE_SIZE, collected in SEPARATE passes as
MI355X_MICROARCH.md "rocprofv3 PMC slots" requires: FETCH_SIZE costs 3 of the 4 TCC slots, WRITE_SIZE 2)
into per-kernel HBM traffic per launch.

    rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d gpurun_out/pmc_fetch -- python3 bench.py --steps 3 --warmup 1 --no-cpu-baseline
    rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d gpurun_out/pmc_write -- python3 bench.py --steps 3 --warmup 1 --no-cpu-baseline
    python tools/pmc_summary.py gpurun_out/pmc_fetch gpurun_out/pmc_write profiles/<name>.json

Corrections (MI355X_MICROARCH.md, section HBM): the counters are in KiB; on gfx950 FETCH_SIZE reports
exactly half of the bytes of a coalesced streaming read, so it is doubled.  Calibration in THIS access
pattern: k_sort_scatter must read 8 B per pair and FETCH_SIZE x 2 lands within 2 % of that; WRITE_SIZE is
taken as is (it matches k_emit's 8 B per instance within 5 %).
"""
import collections
import csv
import glob
import json
import sys


def load(dirname, counter):
    path = (glob.glob(dirname + "/*/*counter_collection.csv") + glob.glob(dirname + "/*counter_collection.csv"))[0]
    d = collections.defaultdict(list)
    for r in csv.DictReader(open(path)):
        kn = r["Kernel_Name"]
        if kn.startswith("void "):  # template instantiations are printed with their return type
            kn = kn[5:]
        if r["Counter_Name"] == counter and kn.startswith("gsr::"):
            name = kn.split("(")[0].replace("gsr::", "").split("<")[0]
            if name == "k_emit_scatter":  # the tile sort's first pass (pairs generated in place): same profiler id
                name = "k_sort_scatter"
            if name == "k_sort_hist_all":  # the depth sort's all-digit histogram: bench.py calls it by its profiler id
                name = "k_sort_hist[depth]"
            d[name].append((float(r["Counter_Value"]), int(r["Grid_Size"])))
    return d


def split_sorts(name, lst):
    """The radix-sort kernels run on P pairs (depth sort) and on R pairs (tile sort): separate them by grid size."""
    if not name.startswith("k_sort") or not lst:
        return {name: lst}
    big = max(g for _, g in lst)
    return {name: [x for x in lst if x[1] * 4 > big], name + "[depth]": [x for x in lst if x[1] * 4 <= big]}


META_DEFAULT = {"workload": "C3", "sh_degree": 0, "loss": "seeded", "forward_only": False, "reference_rects": False,
                "n_gpus": 1}


def parse_meta(pairs):
    """key=value arguments describing the profiled bench command; bench.py prints a committed summary's figures only
    when its _meta equals the run's own workload / flags."""
    meta = dict(META_DEFAULT)
    for kv in pairs:
        k, v = kv.split("=", 1)
        meta[k] = json.loads(v) if v in ("true", "false") or v.lstrip("-").isdigit() else v
    return meta


def main(fetch_dir, write_dir, out_path, *meta):
    f, w = load(fetch_dir, "FETCH_SIZE"), load(write_dir, "WRITE_SIZE")
    out = {"_doc": "HBM bytes per launch = 2 * FETCH_SIZE KiB * 1024 + WRITE_SIZE KiB * 1024 (see tools/pmc_summary.py)",
           "_meta": parse_meta(meta)}
    for k in sorted(set(f) | set(w)):
        fs, ws = split_sorts(k, f.get(k, [])), split_sorts(k, w.get(k, []))
        for kk in sorted(set(fs) | set(ws)):
            fa = [v for v, _ in fs.get(kk, [])]
            wa = [v for v, _ in ws.get(kk, [])]
            if not fa or not wa:
                continue
            fetch_b = 2.0 * 1024.0 * sum(fa) / len(fa)
            write_b = 1024.0 * sum(wa) / len(wa)
            out[kk] = {"launches_sampled": len(fa), "fetch_bytes": round(fetch_b), "write_bytes": round(write_b),
                       "hbm_bytes_per_launch": round(fetch_b + write_b)}
    json.dump(out, open(out_path, "w"), indent=1, sort_keys=True)
    for k, v in out.items():
        if not k.startswith("_"):
            print("%-28s fetch %8.1f MB  write %8.1f MB" % (k, v["fetch_bytes"] / 1e6, v["write_bytes"] / 1e6))


if __name__ == "__main__":
    main(*sys.argv[1:])
