#!/bin/bash
cd $GRAFT_REPO_ROOT
python -m pytest tests/test_binned_gpu.py tests/test_parity_gpu.py tests/test_fuzz_gpu.py tests/test_residual_gpu.py tests/test_point_order_gpu.py tests/test_trace_gpu.py -x -q -m gpu > gpurun_out/r4_t10.log 2>&1; echo "tests rc=$?"; tail -3 gpurun_out/r4_t10.log
export PIGS_AMD_HOST=ctypes
for rep in 1 2; do
python3 tools/kernel_times.py 0.5 2>&1 | grep kappa
PIGS_AMD_LIB=build/variants/libpigs_bw5.so python3 tools/kernel_times.py 0.5 2>&1 | grep kappa
PIGS_AMD_LIB=build/variants/libpigs_bw4.so python3 tools/kernel_times.py 0.5 2>&1 | grep kappa
done
python3 tools/kernel_times.py 1.3 2>&1 | grep kappa
PIGS_AMD_LIB=build/variants/libpigs_bw4.so python3 tools/kernel_times.py 1.3 2>&1 | grep kappa
unset PIGS_AMD_HOST
for c in clustered random; do python3 tools/preprocess_cases.py $c 0.5 2>&1 | grep kappa; done
