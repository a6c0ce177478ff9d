# PMC passes over the 3-channel edge-layer kernels (scripts/one_thin.py)
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/pmc_thin
mkdir -p $O
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_LDS_BANK_CONFLICT GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d $O/sq -- python3 $R/scripts/one_thin.py > $O/sq.log 2>&1 &&
rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SALU SQ_ACTIVE_INST_LDS SQ_WAVES --kernel-trace --output-format csv -d $O/inst -- python3 $R/scripts/one_thin.py > $O/inst.log 2>&1 &&
rocprofv3 --pmc SQ_WAIT_INST_LDS SQ_INST_CYCLES_VMEM SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_SCA SQ_INSTS_SMEM SQ_IFETCH SQ_INSTS_BRANCH --kernel-trace --output-format csv -d $O/misc -- python3 $R/scripts/one_thin.py > $O/misc.log 2>&1
find $O -name "*counter_collection.csv"
