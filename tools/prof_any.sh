#!/bin/bash
# usage (on the GPU box, from the repo root): tools/prof_any.sh <tag> <script.py> [args...]
# rocprofv3 --kernel-trace --stats on any script of tools/; prints the kernel summary
set -e
tag=$1; shift
script=$1; shift
out=$GRAFT_REPO_ROOT/gpurun_out/prof_$tag
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $out -- python3 $GRAFT_REPO_ROOT/$script "$@" > $GRAFT_REPO_ROOT/gpurun_out/prof_$tag.log 2>&1
tail -2 $GRAFT_REPO_ROOT/gpurun_out/prof_$tag.log
python3 $GRAFT_REPO_ROOT/tools/prof_summary.py $out
