#!/bin/bash
# One-off libraries of a kernel file with pieces of its inner loop compiled out (timing experiments only -- results are
# wrong by construction):  abl_build.sh <ring|wx> <bits> ...   ->  experiments/abl/libabl_<which>_<bits>.so
#   ring: conv_ring.hip,       -DVG_RING_ABL=<bits>
#   wx:   wgrad_bf16split.hip, -DVG_WX_ABL=<bits>
#   tfwd: conv_thin_fwd.hip,   -DVG_TF_ABL=<bits>
#   twg:  conv_thin_wgrad.hip, -DVG_TWG_ABL=<bits>
#   gemm: gemm_split.hip,      -DVG_GEMM_ABL=<bits>
# VG_ABL_EXTRA="-DVG_RING_REGF=1" abl_build.sh ring 512   (512: no ablation bit -- a full kernel with the extra defines)
set -e
cd "$(dirname "$0")/.."
C=disentangle_mlp_amd/csrc
which=$1; shift
case $which in
  ring) SRC=conv_ring; DEF=VG_RING_ABL ;;
  wx)   SRC=wgrad_bf16split; DEF=VG_WX_ABL ;;
  tfwd) SRC=conv_thin_fwd; DEF=VG_TF_ABL ;;
  twg)  SRC=conv_thin_wgrad; DEF=VG_TWG_ABL ;;
  gemm) SRC=gemm_split; DEF=VG_GEMM_ABL ;;
  *) echo "usage: $0 <ring|wx|tfwd|twg|gemm> <bits> ..."; exit 2 ;;
esac
mkdir -p experiments/abl
OBJS=$(ls $C/build/*.o | grep -v "/$SRC.o")
for bits in "$@"; do
  ( /opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -Iinclude -I$C -Wno-unused-result -mllvm -pragma-unroll-threshold=131072 -D$DEF=$bits $VG_ABL_EXTRA -c $C/$SRC.hip -o experiments/abl/${which}_$bits.o &&
    /opt/rocm/bin/hipcc -shared -fPIC --offload-arch=gfx950 -o experiments/abl/libabl_${which}_$bits.so experiments/abl/${which}_$bits.o $OBJS ) &
done
wait
ls -la experiments/abl/*.so
