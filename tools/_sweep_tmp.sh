for rep in 1 2; do for v in "" _v6; do echo "variant=$v"; GSR_LIB=$PWD/gs-livm_amd/libgsraster_hip$v.so python tools/time_forward.py C3 8 2>&1 | grep -E "k_preprocess"; done; done
timeout -k 10 900 python -m pytest tests -m gpu -x -q > gpurun_out/t7.log 2>&1; tail -3 gpurun_out/t7.log
