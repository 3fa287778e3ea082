#!/bin/bash
# usage (on the GPU box, from the repo root): tools/pmc_run.sh <tag> "<counter set>" <prof_step.py args...>
# one rocprofv3 --pmc pass (kernel trace only) of tools/prof_step.py; prints per-kernel counter means
set -e
tag=$1; shift
set_=$1; shift
out=$GRAFT_REPO_ROOT/gpurun_out/pmc_$tag
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --pmc $set_ --output-format csv -d $out -- python3 $GRAFT_REPO_ROOT/tools/prof_step.py "$@" > $GRAFT_REPO_ROOT/gpurun_out/pmc_$tag.log 2>&1
python3 $GRAFT_REPO_ROOT/tools/pmc_summary.py $out
