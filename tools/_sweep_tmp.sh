timeout -k 10 900 python -m pytest tests -m gpu -x -q > gpurun_out/t10.log 2>&1; tail -3 gpurun_out/t10.log
grep -q " passed" gpurun_out/t10.log || exit 1
for rep in 1 2; do
for mode in 1 0; do echo "GSR_ASYNC_FAR=$mode"; GSR_ASYNC_FAR=$mode python bench.py --no-cpu-baseline 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read()); k=d['kernels']
print(' ms/step %.4f  sum %.4f  model_step %.4f bwd %.4f fwd %.4f emit %.4f' % (d['ms_per_step'], d['whole_path']['kernel_ms_per_step'], k['k_model_step']['ms_per_step'], k['k_blend_backward']['ms_per_step'], k['k_blend_forward']['ms_per_step'], k['k_emit']['ms_per_step']))"; done; done
