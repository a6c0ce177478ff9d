# rocprofv3 --kernel-trace --stats over THE BENCH COMMAND (graph replay).  Usage (GPU box): bash scripts/prof_bench.sh <out-dir> [extra bench args]
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/${1:-prof_bench}
shift
mkdir -p $O
rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats -- python3 $R/bench.py --steps 10 --warmup 2 --no-cpu-baseline --no-opt-in "$@" > $O/stats.log 2>&1
echo "rc=$?"
find $O -name "*kernel_stats.csv" | head
