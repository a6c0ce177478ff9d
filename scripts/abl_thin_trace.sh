# kernel-trace timing of each thin-forward ablation library (python-side timing is launch-bound for kernels this short)
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
for bits in "$@"; do
  rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/abl_thin_$bits -- python3 $R/scripts/abl_thin.py $bits > $R/gpurun_out/abl_thin_$bits.log 2>&1 || exit 1
done
