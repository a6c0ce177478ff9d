# The kernel-trace statistics and the FETCH_SIZE / WRITE_SIZE / L2 hit-miss passes of scripts/pmc_bench.sh only (for
# scripts/pmc_families.py).  Usage (GPU box): bash scripts/pmc_fetch.sh <out-dir>
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/${1:-pmc_fetch}
mkdir -p $O
CMD="python3 $R/bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-opt-in"
rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats -- $CMD > $O/stats.log 2>&1 &&
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $O/fetch -- $CMD > $O/fetch.log 2>&1 &&
rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $O/write -- $CMD > $O/write.log 2>&1 &&
rocprofv3 --pmc TCC_HIT_sum TCC_MISS_sum --kernel-trace --output-format csv -d $O/tcc -- $CMD > $O/tcc.log 2>&1
echo "rc=$?"
