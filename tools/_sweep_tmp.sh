for rep in 1 2 3; do python bench.py --workload C5shape --loss photometric --no-cpu-baseline 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print(' C5shape+photo ms/step %.4f  sum %.4f' % (d['ms_per_step'], d['whole_path']['kernel_ms_per_step']))"; done
python bench.py --no-cpu-baseline 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print(' C3 ms/step %.4f  sum %.4f' % (d['ms_per_step'], d['whole_path']['kernel_ms_per_step']))"
