# PMC passes over the ring kernel on the dominant shape (128 -> 256, 32x32 -> 16x16, B = 128, bf16x6)
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/pmc_ring
mkdir -p $O
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_LDS_BANK_CONFLICT GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d $O/sq -- python3 $R/scripts/one_conv.py fwd -1 128 256 32 > $O/sq.log 2>&1 &&
rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_SALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_LDS_IDX_ACTIVE --kernel-trace --output-format csv -d $O/inst -- python3 $R/scripts/one_conv.py fwd -1 128 256 32 > $O/inst.log 2>&1 &&
rocprofv3 --pmc SQ_WAIT_INST_LDS SQ_LDS_ADDR_CONFLICT SQ_LDS_DATA_FIFO_FULL SQ_LDS_CMD_FIFO_FULL SQ_LDS_UNALIGNED_STALL SQ_INSTS_LDS SQ_ACTIVE_INST_MISC SQ_INST_CYCLES_VMEM --kernel-trace --output-format csv -d $O/lds -- python3 $R/scripts/one_conv.py fwd -1 128 256 32 > $O/lds.log 2>&1
find $O -name "*counter_collection.csv"
