# kernel-trace timing of ablation libraries: abl_trace.sh <script.py> <kernel substring> <bits> ...
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
script=$1; sub=$2; shift 2
for bits in "$@"; do
  rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/abl_tr_$bits -- python3 $R/scripts/$script $bits > $R/gpurun_out/abl_tr_$bits.log 2>&1 || exit 1
  f=$(find $R/gpurun_out/abl_tr_$bits -name "*kernel_stats.csv" | head -1)
  echo "bits $bits: $(grep "$sub" $f | awk -F, '{printf "%s calls avg %.1f us", $2, $4/1000}')" >> $R/gpurun_out/abl_trace_summary.log
done
cat $R/gpurun_out/abl_trace_summary.log
