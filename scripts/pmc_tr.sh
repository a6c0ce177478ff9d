cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r4_pmc_tr
mkdir -p $O
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $O/fetch -- python3 $R/scripts/one_conv.py tr -1 256 128 16 > $O/fetch.log 2>&1 &&
rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $O/write -- python3 $R/scripts/one_conv.py tr -1 256 128 16 > $O/write.log 2>&1
echo rc=$?
