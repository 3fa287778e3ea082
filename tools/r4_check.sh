#!/bin/bash
# round-4 development check on the GPU box: new tests, cold / warm / fwd+bwd kernel profiles of the default build and
# of the A/B switches, then (unless "quick") the whole GPU suite.  usage: tools/r4_check.sh [quick]
cd $GRAFT_REPO_ROOT
python -m pytest tests/test_lattice_gpu.py tests/test_fused_first_gpu.py tests/test_host_gpu.py tests/test_graph_gpu.py tests/test_distributed_gpu.py -x -q -m gpu > gpurun_out/r4_t1.log 2>&1
echo "new tests rc=$?"; tail -5 gpurun_out/r4_t1.log
tools/prof_run.sh r4a_cold cold --steps 200 || exit 1
tools/prof_run.sh r4a_cold_defer cold --steps 200 --defer || exit 1
PIGS_LATTICE=0 tools/prof_run.sh r4a_cold_base cold --steps 200 || exit 1
tools/prof_run.sh r4a_warm warm --steps 200 || exit 1
tools/prof_run.sh r4a_fwdbwd fwdbwd --steps 100 || exit 1
python - <<'PY'
import csv, glob
import numpy as np
for tag, key in (("r4a_cold", "samples_bbox"), ("r4a_cold_defer", "samples_bbox"), ("r4a_cold_base", "samples_bbox"), ("r4a_warm", "plan_count"), ("r4a_fwdbwd", "plan_count")):
    f = glob.glob(f"gpurun_out/prof_{tag}/*/*kernel_trace.csv")[0]
    rows = sorted(csv.DictReader(open(f)), key=lambda r: int(r["Start_Timestamp"]))
    starts = np.array([int(r["Start_Timestamp"]) for r in rows if key in r["Kernel_Name"]])
    per = np.diff(starts)[20:] / 1e3
    print(f"{tag}: step period on the device timeline: median {np.median(per):.1f} us, mean {per.mean():.1f}, p90 {np.percentile(per, 90):.1f}")
PY
[ "$1" = quick ] && exit 0
python -m pytest tests -x -q -m gpu > gpurun_out/r4_t2.log 2>&1
echo "all gpu tests rc=$?"; tail -5 gpurun_out/r4_t2.log
