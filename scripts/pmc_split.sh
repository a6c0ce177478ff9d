# PMC pass over the split-bf16 forward kernel on the dominant shape (128 -> 256, 32x32 -> 16x16, B = 128)
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
for MODE in bf16x3 bf16x6; do
  O=$R/gpurun_out/pmc_$MODE
  mkdir -p $O
  VG_CONV_ARITH=$MODE rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_LDS_BANK_CONFLICT GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d $O/sq -- python3 $R/scripts/one_conv.py fwd -1 128 256 32 > $O/sq.log 2>&1 &&
  VG_CONV_ARITH=$MODE rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_SALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_LDS_IDX_ACTIVE --kernel-trace --output-format csv -d $O/inst -- python3 $R/scripts/one_conv.py fwd -1 128 256 32 > $O/inst.log 2>&1
done
find $R/gpurun_out/pmc_bf16x3 $R/gpurun_out/pmc_bf16x6 -name "*counter_collection.csv"
