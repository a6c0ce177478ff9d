cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/pmc_v5
mkdir -p $O
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $O/fetch -- python3 $R/scripts/one_conv.py fwd -1 128 256 32 > $O/fetch.log 2>&1 &&
rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $O/write -- python3 $R/scripts/one_conv.py fwd -1 128 256 32 > $O/write.log 2>&1 &&
rocprofv3 --pmc TCC_HIT_sum TCC_MISS_sum --kernel-trace --output-format csv -d $O/tcc -- python3 $R/scripts/one_conv.py fwd -1 128 256 32 > $O/tcc.log 2>&1 &&
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_LDS_BANK_CONFLICT GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d $O/sq -- python3 $R/scripts/one_conv.py fwd -1 128 256 32 > $O/sq.log 2>&1
find $O -name "*.csv" | head -20
