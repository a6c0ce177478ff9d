# rocprofv3 passes over THE BENCH COMMAND (one MI355X): kernel-trace statistics, then hardware counters in separate
# passes (FETCH_SIZE | WRITE_SIZE | TCC hit/miss | SQ_* + GRBM_GUI_ACTIVE), each with --kernel-trace only, as
# MI355X_MICROARCH.md (HBM / rocprofv3 sections) prescribes.  scripts/pmc_summarize.py turns the counter CSVs into
# profiles/rNN_pmc_*.json for the dominant kernel.  Usage (on the GPU box): bash scripts/pmc_bench.sh <out-dir>
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/${1:-pmc_bench}
mkdir -p $O
CMD="python3 $R/bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-opt-in"
rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats -- $CMD > $O/stats.log 2>&1 &&
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $O/fetch -- $CMD > $O/fetch.log 2>&1 &&
rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $O/write -- $CMD > $O/write.log 2>&1 &&
rocprofv3 --pmc TCC_HIT_sum TCC_MISS_sum --kernel-trace --output-format csv -d $O/tcc -- $CMD > $O/tcc.log 2>&1 &&
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_LDS_BANK_CONFLICT GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d $O/sq -- $CMD > $O/sq.log 2>&1
echo "rc=$?"
find $O -name "*.csv" | head -30
