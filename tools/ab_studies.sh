#!/bin/bash
# usage (GPU box, repo root): tools/ab_studies.sh > gpurun_out/r03_ab_studies.txt
# The two build knobs that were measured and left off, against the default build on the SAME box (variant
# libraries from tools/build_variant.sh, driven through the ctypes host; build them first, here in the
# container: block, fusedbuild, noatomics, norows, neither -- see the loop at the end of this file):
#   PIGS_BWD_BLOCK=1   backward over block lists (one wave = four tiles)
#   PIGS_FUSED_BUILD=1 Gaussian chain count + scan + scatter in one launch behind device-wide barriers
export PIGS_AMD_HOST=ctypes
echo "# backward kernel + unpermute at C3, kappa 0.5, by backward cut-off (tools/kernel_times.py): default (tile lists) vs PIGS_BWD_BLOCK=1"
for qb in 36 40 44; do
  python3 tools/kernel_times.py 0.5 $qb 2>&1 | grep kappa
  PIGS_AMD_LIB=build/variants/libpigs_block.so python3 tools/kernel_times.py 0.5 $qb 2>&1 | grep kappa
done
echo "# the backward kernel by compiled-out phases (probe builds -DPIGS_BWD_PROBE_NO_ATOMICS / -DPIGS_BWD_PROBE_NO_ROWS): everything, without the global atomics, without the row arithmetic, without both"
for v in default noatomics norows neither; do
  if [ $v = default ]; then python3 tools/kernel_times.py 0.5 2>&1 | grep kappa; else PIGS_AMD_LIB=build/variants/libpigs_$v.so python3 tools/kernel_times.py 0.5 2>&1 | grep kappa; fi
done
echo "# warm step (samples half reused) at C3, kappa 0.5 (tools/prof_step.py warm --steps 200): default (three launches) vs PIGS_FUSED_BUILD=1"
python3 tools/prof_step.py warm --steps 200 2>&1 | grep us/step
PIGS_AMD_LIB=build/variants/libpigs_fusedbuild.so python3 tools/prof_step.py warm --steps 200 2>&1 | grep us/step
echo "# conic-gradient error of the four worst tools/fuzz_big.py cases by backward cut-off (tools/fuzz_diag.py)"
unset PIGS_AMD_HOST
for k in 31 11 30 2; do python3 tools/fuzz_diag.py $k 2>&1 | grep -E "^case|conics"; done

# variant builds (run in the build container before shipping the tree to the GPU box):
#   tools/build_variant.sh block -DPIGS_BWD_BLOCK=1; tools/build_variant.sh fusedbuild -DPIGS_FUSED_BUILD=1
#   tools/build_variant.sh noatomics -DPIGS_BWD_PROBE_NO_ATOMICS; tools/build_variant.sh norows -DPIGS_BWD_PROBE_NO_ROWS
#   tools/build_variant.sh neither -DPIGS_BWD_PROBE_NO_ATOMICS -DPIGS_BWD_PROBE_NO_ROWS
