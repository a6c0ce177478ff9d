# Builds and runs the f16 MFMA check on the GPU box:  bash scripts/mfma_f16.sh [out-name] [warm-up seconds]
# -> gpurun_out/<out-name>.jsonl
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
O=$R/gpurun_out/${1:-mfma_f16}.jsonl
mkdir -p $R/gpurun_out
/opt/rocm/bin/hipcc -O3 -std=c++17 --offload-arch=gfx950 $R/scripts/mfma_f16.hip -o /tmp/mfma_f16 && timeout -k 10 300 /tmp/mfma_f16 ${2:-2} > $O && cat $O
