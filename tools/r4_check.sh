#!/bin/bash
# round-4 development check on the GPU box: new tests, cold / warm kernel profiles of the default build and of
# the A/B switches, the atomics micro-benchmarks, then the whole GPU suite.  usage: tools/r4_check.sh [quick]
cd $GRAFT_REPO_ROOT
python -m pytest tests/test_lattice_gpu.py tests/test_fused_first_gpu.py tests/test_host_gpu.py tests/test_graph_gpu.py tests/test_distributed_gpu.py -x -q -m gpu > gpurun_out/r4_t1.log 2>&1
echo "new tests rc=$?"; tail -5 gpurun_out/r4_t1.log
tools/prof_run.sh r4a_cold cold --steps 200 || exit 1
PIGS_NO_FUSED_FIRST=1 tools/prof_run.sh r4a_cold_nofuse cold --steps 200 || exit 1
PIGS_NO_FUSED_FIRST=1 PIGS_LATTICE=0 tools/prof_run.sh r4a_cold_base cold --steps 200 || exit 1
tools/prof_run.sh r4a_warm warm --steps 200 || exit 1
tools/prof_run.sh r4a_fwdbwd fwdbwd --steps 100 || exit 1
(cd tools/ubench && ./atomics3 > ../../gpurun_out/r4_atomics3.txt 2>&1; ./atomics4 > ../../gpurun_out/r4_atomics4.txt 2>&1; true)
cat gpurun_out/r4_atomics3.txt gpurun_out/r4_atomics4.txt
[ "$1" = quick ] && exit 0
python -m pytest tests -x -q -m gpu > gpurun_out/r4_t2.log 2>&1
echo "all gpu tests rc=$?"; tail -5 gpurun_out/r4_t2.log
