#!/bin/bash
cd $GRAFT_REPO_ROOT
export PIGS_AMD_HOST=ctypes
for rep in 1 2 3; do
python3 tools/kernel_times.py 0.5 2>&1 | grep kappa
PIGS_AMD_LIB=build/variants/libpigs_oldbwd.so python3 tools/kernel_times.py 0.5 2>&1 | grep kappa
done
