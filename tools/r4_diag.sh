#!/bin/bash
# round-4 diagnostics on the GPU box: where clustered / random point sets spend their time, the aggregate list probe,
# the conditioning test, a short bench line
cd $GRAFT_REPO_ROOT
python -m pytest tests/test_conditioning_gpu.py -x -q -m gpu -s > gpurun_out/r4_cond.log 2>&1; echo "conditioning rc=$?"; grep -E "ill-conditioned|passed|failed|Error|assert" gpurun_out/r4_cond.log | head
for c in clustered:0.15 random; do python tools/tile_stats.py $c 2>&1 | grep -v amdgpu.ids; done
export TMPDIR=/tmp
for c in clustered:0.15 random; do
  for w in cold warm; do
    ( cd /tmp && rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/prof_r4d_${c%%:*}_$w -- python3 $GRAFT_REPO_ROOT/tools/preprocess_cases.py $c 0.5 $w > $GRAFT_REPO_ROOT/gpurun_out/prof_r4d_${c%%:*}_$w.log 2>&1 )
    grep kappa gpurun_out/prof_r4d_${c%%:*}_$w.log
    python3 tools/prof_summary.py gpurun_out/prof_r4d_${c%%:*}_$w
  done
done
python tools/aggregate_lists_probe.py 2>&1 | grep -v amdgpu.ids
python bench.py --steps 50 --warmup 10 > gpurun_out/r4_bench_short.json 2> gpurun_out/r4_bench_short.err; echo "bench rc=$?"; tail -3 gpurun_out/r4_bench_short.err
python - <<'PY'
import json
d = json.load(open("gpurun_out/r4_bench_short.json"))
for k in ("value", "ms_per_step", "value_warm_plan"):
    print(k, d[k])
print("warm", d["warm_plan"]["ms_per_step"])
for k in ("roofline", "roofline_first", "roofline_bwd"):
    r = d[k]; print(k, r and {x: r[x] for x in ("kernel", "kernel_ms", "frac")})
print("fwd_bwd", {k: v for k, v in d["fwd_bwd"].items() if k != "what"})
print("c2", d["c2"]); print("kappa13", {k: v for k, v in d["kappa_1_3"].items() if k != "valu"}); print("unordered", d["unordered_points"]); print("small", d["small"]); print("two", d["two_streams"])
print("cpu", d.get("cpu_baseline"))
PY
