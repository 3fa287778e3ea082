#!/bin/bash
# usage: tools/build_variant.sh <name> <extra hipcc flags...>   -> build/variants/libpigs_<name>.so (same ABI)
# select it with PIGS_AMD_LIB=build/variants/libpigs_<name>.so
set -e
cd "$(dirname "$0")/.."
name=$1; shift
mkdir -p build/variants
hipcc -O3 -std=c++17 -fPIC -shared --offload-arch=gfx950 -ffp-contract=fast -fno-slp-vectorize -Wno-unused-function "$@" \
  -o build/variants/libpigs_$name.so pigs_amd/csrc/*.hip
echo build/variants/libpigs_$name.so
