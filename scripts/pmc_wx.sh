# kernel trace + PMC passes over the split-bf16 weight gradient on two shapes (B = 128, bf16x6)
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/pmc_wx
mkdir -p $O
for shape in "128 256 32 2" "32 128 64 2"; do
  tag=$(echo $shape | tr ' ' '_')
  rocprofv3 --kernel-trace --stats --output-format csv -d $O/trace_$tag -- python3 $R/scripts/one_wgrad.py $shape > $O/trace_$tag.log 2>&1 &&
  rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_LDS_BANK_CONFLICT GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d $O/sq_$tag -- python3 $R/scripts/one_wgrad.py $shape > $O/sq_$tag.log 2>&1 &&
  rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_SALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_LDS_IDX_ACTIVE --kernel-trace --output-format csv -d $O/inst_$tag -- python3 $R/scripts/one_wgrad.py $shape > $O/inst_$tag.log 2>&1 &&
  rocprofv3 --pmc FETCH_SIZE WRITE_SIZE TCC_HIT_sum TCC_MISS_sum TCP_TCC_READ_REQ_sum --kernel-trace --output-format csv -d $O/mem_$tag -- python3 $R/scripts/one_wgrad.py $shape > $O/mem_$tag.log 2>&1 || exit 1
done
find $O -name "*counter_collection.csv" -o -name "*kernel_stats.csv"
