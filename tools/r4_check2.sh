#!/bin/bash
cd $GRAFT_REPO_ROOT
python -m pytest tests -x -q -m gpu > gpurun_out/r4_t4.log 2>&1; echo "all gpu tests rc=$?"; tail -4 gpurun_out/r4_t4.log
tools/prof_run.sh r4c_cold cold --steps 200 || exit 1
tools/prof_run.sh r4c_c2cold cold --steps 200 --lat 90 --res 256 || exit 1
tools/prof_run.sh r4c_512cold cold --steps 200 --lat 128 --res 512 || exit 1
export TMPDIR=/tmp
for c in clustered:0.15; do
    ( cd /tmp && rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/prof_r4f_${c%%:*}_cold -- python3 $GRAFT_REPO_ROOT/tools/preprocess_cases.py $c 0.5 cold > $GRAFT_REPO_ROOT/gpurun_out/prof_r4f_${c%%:*}_cold.log 2>&1 )
    grep kappa gpurun_out/prof_r4f_${c%%:*}_cold.log
    python3 tools/prof_summary.py gpurun_out/prof_r4f_${c%%:*}_cold
done
