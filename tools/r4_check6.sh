#!/bin/bash
cd $GRAFT_REPO_ROOT
python -m pytest tests/test_point_order_gpu.py -x -q -m gpu > gpurun_out/r4_t9.log 2>&1; echo "tests rc=$?"; tail -3 gpurun_out/r4_t9.log
export TMPDIR=/tmp
for c in clustered:0.15; do
    ( cd /tmp && rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/prof_r4g_${c%%:*}_cold -- python3 $GRAFT_REPO_ROOT/tools/preprocess_cases.py $c 0.5 cold > $GRAFT_REPO_ROOT/gpurun_out/prof_r4g_${c%%:*}_cold.log 2>&1 )
    grep kappa gpurun_out/prof_r4g_${c%%:*}_cold.log
    python3 tools/prof_summary.py gpurun_out/prof_r4g_${c%%:*}_cold
done
