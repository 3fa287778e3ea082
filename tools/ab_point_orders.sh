#!/bin/bash
# usage (GPU box, repo root): tools/ab_point_orders.sh > gpurun_out/r03_ab_point_orders.txt
# Same-box A/B of the two mechanisms for points in no order, C3 size, kappa 0.5, through
# tools/preprocess_cases.py (cold / warm step, forward launches, backward launches):
#   the library's own choice   |   PIGS_SAMPLES_ORDER=ordered (one-pass samples build) PIGS_STAGE=0 (no staging)
#   |   coarse-bin build without staging   |   one-pass build with staging
for c in random shuffled clustered:0.3 grid; do
  echo "# $c"
  python3 tools/preprocess_cases.py $c 0.5 2>&1 | grep kappa | sed 's/^/library          : /'
  PIGS_SAMPLES_ORDER=ordered PIGS_STAGE=0 python3 tools/preprocess_cases.py $c 0.5 2>&1 | grep kappa | sed 's/^/one-pass, direct  : /'
  PIGS_SAMPLES_ORDER=unordered PIGS_STAGE=0 python3 tools/preprocess_cases.py $c 0.5 2>&1 | grep kappa | sed 's/^/coarse, direct    : /'
  PIGS_SAMPLES_ORDER=ordered PIGS_STAGE=1 python3 tools/preprocess_cases.py $c 0.5 2>&1 | grep kappa | sed 's/^/one-pass, staged  : /'
  PIGS_SAMPLES_ORDER=unordered PIGS_STAGE=1 python3 tools/preprocess_cases.py $c 0.5 2>&1 | grep kappa | sed 's/^/coarse, staged    : /'
done
