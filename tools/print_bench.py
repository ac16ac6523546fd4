"""Condensed view of one or more bench.py JSON lines: python tools/print_bench.py file.json [...]"""
import json
import sys

items = []
for path in sys.argv[1:]:
    try:
        lines = [l for l in open(path) if l.startswith("{")]
        items += [("%s#%d" % (path, k) if len(lines) > 1 else path, json.loads(l)) for k, l in enumerate(lines)]
    except (OSError, ValueError) as e:
        print(path, "unreadable:", e)
for path, d in items:
    sp = d.get("speculation") or {}
    print("%s [%s]: ms/step %.4f  ms/view %.4f  value %.1f  | ovf %s skips %s misses %s async %s lost %s scale %s" % (
        path, (d.get("config") or {}).get("workload", "")[:60], d["ms_per_step"], d.get("ms_per_view", d["ms_per_step"]), d["value"], sp.get("overflows"),
        sp.get("far_skips"), sp.get("far_skip_misses"), sp.get("async_far_frames"), sp.get("async_outcomes_lost"),
        sp.get("near_budget_scale_q8")))
    K = (d.get("config") or {}).get("views_per_gpu_per_step", 1) or 1
    ks = d.get("kernels") or {}
    print("    per view (us):", "  ".join("%s %.0f" % (k.replace("k_", ""), 1e3 * v["ms_per_step"] / K)
                                         for k, v in list(ks.items())[:16]))
    ws = d.get("workload_stats") or {}
    if ws:
        print("    R %s  near/far %s  walk %s" % (ws.get("R"), ws.get("near_far"),
                                                    {k: int(v) for k, v in ws.get("backward_walk_per_tile", {}).items()}))
    if d.get("per_step"):
        ps = d["per_step"]
        print("    per-step ms:", " ".join("%.2f" % p["ms"] for p in ps))
        print("    per-step (skips, misses, ovf, scale):", " ".join("%d/%d/%d/%d" % (
            p["far_skips"], p["far_skip_misses"], p["overflows"], p["near_budget_scale_q8"]) for p in ps))
