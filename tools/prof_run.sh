#!/bin/bash
# usage (on the GPU box, from the repo root): tools/prof_run.sh <tag> <prof_step.py args...>
# runs rocprofv3 --kernel-trace --stats on tools/prof_step.py and prints the kernel summary
set -e
tag=$1; shift
out=$GRAFT_REPO_ROOT/gpurun_out/prof_$tag
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $out -- python3 $GRAFT_REPO_ROOT/tools/prof_step.py "$@" > $GRAFT_REPO_ROOT/gpurun_out/prof_$tag.log 2>&1
tail -2 $GRAFT_REPO_ROOT/gpurun_out/prof_$tag.log
python3 $GRAFT_REPO_ROOT/tools/prof_summary.py $out
