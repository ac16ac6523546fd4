timeout -k 10 900 python -m pytest tests -m gpu -x -q > gpurun_out/t11.log 2>&1; tail -2 gpurun_out/t11.log
grep -q " passed" gpurun_out/t11.log || exit 1
python bench.py --no-cpu-baseline 2>/dev/null > gpurun_out/b6.json; python tools/print_bench.py gpurun_out/b6.json
