#!/bin/bash
cd $GRAFT_REPO_ROOT
python -m pytest tests/test_lattice_gpu.py tests/test_point_order_gpu.py tests/test_fused_first_gpu.py -x -q -m gpu > gpurun_out/r4_t6.log 2>&1; echo "tests rc=$?"; tail -3 gpurun_out/r4_t6.log
tools/prof_run.sh r4e_cold cold --steps 200 || exit 1
PIGS_LATTICE=1 tools/prof_run.sh r4e_512cold_lat cold --steps 200 --lat 128 --res 512 || exit 1
python tools/size_sweep.py 2>&1 | grep -v amdgpu.ids | tail -12
