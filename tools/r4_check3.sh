#!/bin/bash
cd $GRAFT_REPO_ROOT
python -m pytest tests/test_lattice_gpu.py tests/test_point_order_gpu.py tests/test_conditioning_gpu.py tests/test_binned_gpu.py -x -q -m gpu > gpurun_out/r4_t5.log 2>&1; echo "tests rc=$?"; tail -3 gpurun_out/r4_t5.log
tools/prof_run.sh r4d_cold cold --steps 200 || exit 1
PIGS_LATTICE=0 tools/prof_run.sh r4d_cold_sorted cold --steps 200 || exit 1
tools/prof_run.sh r4d_c2cold cold --steps 200 --lat 90 --res 256 || exit 1
tools/prof_run.sh r4d_512cold cold --steps 200 --lat 128 --res 512 || exit 1
PIGS_LATTICE=1 tools/prof_run.sh r4d_512cold_lat cold --steps 200 --lat 128 --res 512 || exit 1
