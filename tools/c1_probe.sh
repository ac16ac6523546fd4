for i in 1 2 3 4; do
python3 bench.py --workload C1 --no-cpu-baseline --steps 200 --warmup 20 --per-step > gpurun_out/c1p_$i.json 2>/dev/null
python3 - gpurun_out/c1p_$i.json <<'PY'
import json,sys
d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
ms=sorted(s["ms"] for s in d["per_step"])
n=len(ms)
print("ms/step %.4f  p10 %.3f p50 %.3f p90 %.3f max %.3f  slow_path %d" % (d["ms_per_step"], ms[n//10], ms[n//2], ms[9*n//10], ms[-1], d.get("mailbox_slow_path_hits",-1)))
PY
done
