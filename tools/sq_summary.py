#!/usr/bin/env python3
"""Per-kernel SQ counter summary from one rocprofv3 --pmc run (own run, --kernel-trace only):

    rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_ACTIVE_INST_VALU SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_BUSY_CYCLES \
        --output-format csv -d gpurun_out/pmc_sq -o run -- python3 bench.py --steps 3 --warmup 1 --no-cpu-baseline
    python tools/sq_summary.py gpurun_out/pmc_sq profiles/<name>.json

valu_issue_ms = wave-instructions x 2 cycles (wave64 on the SIMD-32 VALU, MI355X_MICROARCH.md) / (1024 SIMDs x 2.4 GHz):
the time the kernel would need if VALU issue were the only limit."""
import collections, csv, glob, json, sys

def main(d, out, *meta):
    path = (glob.glob(d + "/*/*counter_collection.csv") + glob.glob(d + "/*counter_collection.csv"))[0]
    agg = collections.defaultdict(lambda: collections.defaultdict(float))
    disp = collections.defaultdict(set)
    dur = collections.defaultdict(float)
    for r in csv.DictReader(open(path)):
        kn = r["Kernel_Name"]
        if kn.startswith("void "):
            kn = kn[5:]
        if not kn.startswith("gsr::"):
            continue
        name = kn.split("(")[0].replace("gsr::", "")
        agg[name][r["Counter_Name"]] += float(r["Counter_Value"])
        if r["Dispatch_Id"] not in disp[name]:
            disp[name].add(r["Dispatch_Id"])
            dur[name] += (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e6
    from pmc_summary import parse_meta
    res = {"_doc": __doc__.strip().split("\n\n")[-1], "_meta": parse_meta(meta)}
    for k in sorted(agg, key=lambda k: -dur[k]):
        n = len(disp[k])
        c = {cn: v / n for cn, v in agg[k].items()}
        e = {"launches_sampled": n, "ms_under_counters": round(dur[k] / n, 4)}
        e.update({cn: round(v) for cn, v in c.items()})
        if "SQ_INSTS_VALU" in c:
            e["valu_issue_ms"] = round(c["SQ_INSTS_VALU"] * 2 / (1024 * 2.4e9) * 1e3, 4)
        if c.get("SQ_WAVE_CYCLES"):
            for cn in ("SQ_ACTIVE_INST_VALU", "SQ_WAIT_INST_ANY"):
                if cn in c:
                    e[cn + "/SQ_WAVE_CYCLES"] = round(c[cn] / c["SQ_WAVE_CYCLES"], 3)
        res[k] = e
        print("%-44s x%-3d %8.3f ms  VALU %10.0f  issue-bound %.3f ms" % (k[:44], n, e["ms_under_counters"], c.get("SQ_INSTS_VALU", 0), e.get("valu_issue_ms", 0)))
    json.dump(res, open(out, "w"), indent=1)

if __name__ == "__main__":
    main(*sys.argv[1:])
