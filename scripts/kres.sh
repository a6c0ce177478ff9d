#!/bin/bash
# Register / scratch / LDS use of every kernel of one source file: scripts/kres.sh csrc/<file>.hip [extra hipcc flags]
cd "$(dirname "$0")/.."
f=$1; shift
/opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -Iinclude -Idisentangle_mlp_amd/csrc -Wno-unused-result "$@" \
  -Rpass-analysis=kernel-resource-usage -c disentangle_mlp_amd/$f -o /dev/null 2>&1 |
  awk '/Function Name:/ {name=$NF} / VGPRs:/ {v=$(NF-1)} /AGPRs:/ {a=$(NF-1)} /ScratchSize/ {s=$(NF-1)} /LDS Size/ {l=$(NF-1); printf "%-110s vgpr %4s agpr %4s scratch %5s lds %7s\n", name, v, a, s, l}' |
  sed 's/\[-Rpass[^]]*\]//g'
