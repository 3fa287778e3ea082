#!/bin/bash
cd $GRAFT_REPO_ROOT
python -m pytest tests/test_binned_gpu.py tests/test_point_order_gpu.py tests/test_fused_first_gpu.py tests/test_fuzz_gpu.py -x -q -m gpu > gpurun_out/r4_t8.log 2>&1; echo "tests rc=$?"; tail -3 gpurun_out/r4_t8.log
for c in clustered clustered:0.3 random; do python3 tools/preprocess_cases.py $c 0.5 2>&1 | grep kappa; done
