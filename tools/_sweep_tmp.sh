timeout -k 10 600 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "near_far or speculati or c3_near" > gpurun_out/t12.log 2>&1; tail -2 gpurun_out/t12.log
grep -q " passed" gpurun_out/t12.log || exit 1
for rep in 1 2; do
python bench.py --no-cpu-baseline 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read()); k=d['kernels']
g=lambda n: k.get(n,{}).get('ms_per_step',0)
print(' ms/step %.4f  sum %.4f  dsort %.4f dhist/compact %.4f scan %.4f pre %.4f' % (d['ms_per_step'], d['whole_path']['kernel_ms_per_step'], g('k_sort_scatter[depth]'), g('k_sort_hist[depth]'), g('k_scan_offsets'), g('k_preprocess')))"; done
