# bash scripts/mfma_f16_prec.sh [out-name]  -> gpurun_out/<out-name>.jsonl
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
O=$R/gpurun_out/${1:-mfma_f16_prec}.jsonl
mkdir -p $R/gpurun_out
/opt/rocm/bin/hipcc -O3 -std=c++17 --offload-arch=gfx950 $R/scripts/mfma_f16_prec.hip -o /tmp/mfma_f16_prec && timeout -k 10 200 /tmp/mfma_f16_prec > $O && cat $O
