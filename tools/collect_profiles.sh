#!/bin/bash
# usage (on the GPU box, from the repo root): tools/collect_profiles.sh <round tag, e.g. r03> [part]
# Collects everything profiles/README.md lists into gpurun_out/<tag>_profiles/ (copy what is to be judged into
# profiles/ afterwards).  part = stats | pmc | lines | all (default all).  rocprofv3 always with the program
# itself behind `--`, --pmc passes each in their own run with --kernel-trace only.
set -e
tag=${1:-r04}; part=${2:-all}
root=$GRAFT_REPO_ROOT
out=$root/gpurun_out/${tag}_profiles
mkdir -p $out
cd /tmp && export TMPDIR=/tmp
stats() {   # name, program args...
    name=$1; shift
    rm -rf $out/raw_$name
    rocprofv3 --kernel-trace --stats --output-format csv -d $out/raw_$name -- python3 "$@" > $out/$name.log 2>&1
    cp $(find $out/raw_$name -name '*kernel_stats.csv' | head -1) $out/${tag}_${name}_kernel_stats.csv
    echo "== $name"; tail -1 $out/$name.log | cut -c1-300
}
pmc() {     # name, counters, prof_step args...
    name=$1; set_=$2; shift; shift
    rm -rf $out/pmc_$name
    rocprofv3 --kernel-trace --pmc $set_ --output-format csv -d $out/pmc_$name -- python3 $root/tools/prof_step.py "$@" > $out/pmc_$name.log 2>&1
    cp $(find $out/pmc_$name -name '*counter_collection.csv' | head -1) $out/${tag}_c3_k0.5_pmc_$name.csv
    echo "== pmc $name done"
}
if [ $part = stats ] || [ $part = all ]; then
    stats bench $root/bench.py
    stats fwd $root/tools/prof_step.py fwd --steps 200
    stats cold $root/tools/prof_step.py cold --steps 200
    stats warm $root/tools/prof_step.py warm --steps 200
    stats fwdbwd $root/tools/prof_step.py fwdbwd --steps 200
    stats cold_k13 $root/tools/prof_step.py cold --steps 100 --kappa 1.3
    stats fwdbwd_k13 $root/tools/prof_step.py fwdbwd --steps 100 --kappa 1.3
    # round 4 A/B, same box: the index-tiled order off (every grid sorted, as in round 3); the tile lists deferred into
    # the first forward's launch (plan_lists_forward_kernel); BASELINE configs[1]'s sizes
    PIGS_LATTICE=0 stats cold_sorted $root/tools/prof_step.py cold --steps 200
    stats cold_deferred $root/tools/prof_step.py cold --steps 200 --defer
    stats c2_cold $root/tools/prof_step.py cold --steps 200 --lat 90 --res 256
    stats c2_fwdbwd $root/tools/prof_step.py fwdbwd --steps 200 --lat 90 --res 256
    python3 - <<PY > $out/${tag}_step_periods.txt
import csv, glob
import numpy as np
for name, key in (("cold", "samples_bbox"), ("cold_sorted", "samples_bbox"), ("cold_deferred", "samples_bbox"), ("warm", "plan_count"), ("fwdbwd", "plan_count"), ("c2_cold", "samples_bbox")):
    f = glob.glob("$out/raw_%s/**/*kernel_trace.csv" % name, recursive=True)
    if not f:
        continue
    rows = sorted(csv.DictReader(open(f[0])), key=lambda r: int(r["Start_Timestamp"]))
    starts = np.array([int(r["Start_Timestamp"]) for r in rows if key in r["Kernel_Name"]])
    per = np.diff(starts)[20:] / 1e3
    print(f"{name}: step period on the device timeline (start of {key} to the next): median {np.median(per):.1f} us, mean {per.mean():.1f}, p90 {np.percentile(per, 90):.1f} ({len(per)} steps, under rocprofv3 --kernel-trace)")
PY
    cat $out/${tag}_step_periods.txt
fi
if [ $part = pmc ] || [ $part = all ]; then
    SQ="SQ_WAVES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_ACTIVE_INST_VALU SQ_WAIT_ANY SQ_WAIT_INST_ANY"
    pmc fetch "FETCH_SIZE" fwd --steps 20
    pmc write "WRITE_SIZE" fwd --steps 20
    pmc hit "TCC_HIT_sum TCC_MISS_sum" fwd --steps 20
    pmc sq "$SQ" fwd --steps 20
    pmc bfetch "FETCH_SIZE" bwd --steps 20
    pmc bwrite "WRITE_SIZE" bwd --steps 20
    pmc bhit "TCC_HIT_sum TCC_MISS_sum" bwd --steps 20
    pmc bsq "$SQ" bwd --steps 20
fi
if [ $part = lines ] || [ $part = all ]; then
    cd $root
    python3 bench.py > $out/${tag}_bench_c3_k0.5.json 2> $out/bench.err
    python3 bench.py --steps 20 --warmup 5 --no-cpu-baseline > $out/${tag}_bench_c3_k0.5_20steps.json 2>> $out/bench.err
    python3 tools/size_sweep.py > $out/${tag}_size_sweep.txt 2>&1
    python3 tools/bench_aggregate.py 2>&1 | grep -v amdgpu.ids > $out/${tag}_aggregate.txt
    for c in grid random shuffled clustered clustered:0.5 clustered:0.3; do python3 tools/preprocess_cases.py $c 0.5 2>&1 | grep kappa; done > $out/${tag}_point_orders.txt
    for c in grid random; do python3 tools/preprocess_cases.py $c 1.3 2>&1 | grep kappa; done >> $out/${tag}_point_orders.txt
    (cd tools/ubench && ./atomics3; ./atomics4) > $out/${tag}_atomics_ubench.txt 2>&1
    python3 tools/aggregate_lists_probe.py 2>&1 | grep -v amdgpu.ids > $out/${tag}_aggregate_lists_probe.txt
    # the deferred tile lists against the default, every kind of step, same process (C3; BASELINE configs[1]'s sizes; kappa 1.3)
    (python3 tools/ab_defer.py 0.5 2>&1 | grep defer; python3 tools/ab_defer.py 0.5 90 256 2>&1 | grep defer | sed "s/^/c2 sizes: /"; \
     python3 tools/ab_defer.py 1.3 2>&1 | grep defer | sed "s/^/kappa 1.3: /") > $out/${tag}_ab_defer.txt
    # the N > 1 path of bench.py, two gloo ranks sharing the one GPU: a functional rehearsal, not a scaling number
    python3 -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29533 bench.py --gpus 2 --steps 5 --warmup 2 \
        --dist-backend gloo --workload c4 --no-cpu-baseline 2> $out/rehearsal.err | grep '^{' > $out/${tag}_rehearsal_c4_gloo_world2.json
    echo "== lines done"
fi
