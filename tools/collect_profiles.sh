#!/bin/bash
# usage (on the GPU box, from the repo root): tools/collect_profiles.sh <round tag, e.g. r03> [part]
# Collects everything profiles/README.md lists into gpurun_out/<tag>_profiles/ (copy what is to be judged into
# profiles/ afterwards).  part = stats | pmc | lines | all (default all).  rocprofv3 always with the program
# itself behind `--`, --pmc passes each in their own run with --kernel-trace only.
set -e
tag=${1:-r03}; part=${2:-all}
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
    stats cold $root/tools/prof_step.py cold --steps 200
    stats warm $root/tools/prof_step.py warm --steps 200
    stats fwdbwd $root/tools/prof_step.py fwdbwd --steps 200
    stats cold_k13 $root/tools/prof_step.py cold --steps 100 --kappa 1.3
    stats fwdbwd_k13 $root/tools/prof_step.py fwdbwd --steps 100 --kappa 1.3
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
    echo "== lines done"
fi
