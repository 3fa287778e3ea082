#!/bin/bash
# Build timing-only ablation variants of the library (outputs are wrong; only kernel time matters):
#   1 = no pair evaluation, 2 = no candidate test (nothing accepted), 3 = no traversal at all
set -e
cd "$(dirname "$0")/../pigs_amd/csrc"
for v in 1 2 3; do
  hipcc -O3 -std=c++17 -fPIC -shared --offload-arch=gfx950 -ffp-contract=fast -DPIGS_ABLATE=$v -o ../libpigs_amd_ablate$v.so *.hip
done
