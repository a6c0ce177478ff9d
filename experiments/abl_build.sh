#!/bin/bash
# One-off libraries of the ring kernel with pieces of its K step compiled out (conv_ring.hip, VG_RING_ABL bits):
# timing experiments only -- results are wrong by construction.  Output: experiments/abl/libabl_<bits>.so
set -e
cd "$(dirname "$0")/.."
C=disentangle_mlp_amd/csrc
OBJS=$(ls $C/build/*.o | grep -v conv_ring.o)
for bits in "$@"; do
  ( /opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -Iinclude -I$C -Wno-unused-result -mllvm -pragma-unroll-threshold=131072 -DVG_RING_ABL=$bits -c $C/conv_ring.hip -o experiments/abl/ring_$bits.o &&
    /opt/rocm/bin/hipcc -shared -fPIC --offload-arch=gfx950 -o experiments/abl/libabl_$bits.so experiments/abl/ring_$bits.o $OBJS ) &
done
wait
ls -la experiments/abl/*.so
