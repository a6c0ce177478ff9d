# Builds and runs the bf16 MFMA microbenchmark on the GPU box (hipcc is in the image):  bash scripts/mfma_peak.sh [out-name] [warm-up seconds]
# -> gpurun_out/<out-name>.jsonl, one JSON line per (operands, shape, wavefronts per SIMD).
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
O=$R/gpurun_out/${1:-mfma_peak}.jsonl
mkdir -p $R/gpurun_out
/opt/rocm/bin/hipcc -O3 -std=c++17 --offload-arch=gfx950 $R/scripts/mfma_peak.hip -o /tmp/mfma_peak && timeout -k 10 240 /tmp/mfma_peak ${2:-2} > $O && cat $O
