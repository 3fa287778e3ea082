#!/bin/bash
cd $GRAFT_REPO_ROOT
python -m pytest tests/test_lattice_gpu.py tests/test_point_order_gpu.py tests/test_binned_gpu.py -x -q -m gpu > gpurun_out/r4_t3.log 2>&1; echo "tests rc=$?"; tail -3 gpurun_out/r4_t3.log
tools/prof_run.sh r4b_cold cold --steps 200 || exit 1
tools/prof_run.sh r4b_c2cold cold --steps 200 --lat 90 --res 256 || exit 1
PIGS_LATTICE=0 tools/prof_run.sh r4b_c2cold_base cold --steps 200 --lat 90 --res 256 || exit 1
export TMPDIR=/tmp
for c in clustered:0.15 random; do
    ( cd /tmp && rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/prof_r4e_${c%%:*}_cold -- python3 $GRAFT_REPO_ROOT/tools/preprocess_cases.py $c 0.5 cold > $GRAFT_REPO_ROOT/gpurun_out/prof_r4e_${c%%:*}_cold.log 2>&1 )
    grep kappa gpurun_out/prof_r4e_${c%%:*}_cold.log
    python3 tools/prof_summary.py gpurun_out/prof_r4e_${c%%:*}_cold
done
python tools/fuzz_dense.py 1500 > gpurun_out/r4_fuzz_dense.txt 2>&1; tail -4 gpurun_out/r4_fuzz_dense.txt | cut -c1-600
