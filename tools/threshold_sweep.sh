#!/bin/bash
# usage (GPU box, repo root): tools/threshold_sweep.sh > gpurun_out/r03_point_order_thresholds.txt
# Where the coarse-bin samples build and the staging start to pay (COARSE_MIN_POINTS / STAGE_MIN_POINTS, plan.hip):
# uniform random points, one Gaussian per 16 points, kappa 0.5, each mechanism forced on / off by the environment.
for res in 192 256 384 512 768 1024; do
  lat=$((res/4))
  echo "# $res x $res points, $lat x $lat Gaussians"
  for mode in "ordered 0" "unordered 0" "ordered 1" "unordered 1"; do set -- $mode
    PIGS_CASE_RES=$res PIGS_CASE_LAT=$lat PIGS_SAMPLES_ORDER=$1 PIGS_STAGE=$2 python3 tools/preprocess_cases.py random 0.5 2>&1 | grep kappa | sed "s/^/samples build=$1 staging=$2: /"
  done
  PIGS_CASE_RES=$res PIGS_CASE_LAT=$lat python3 tools/preprocess_cases.py random 0.5 2>&1 | grep kappa | sed "s/^/the library's choice            : /"
done
