#!/bin/bash
# The Python-free kernel harnesses (experiments/): each runs in 1-3 s on an MI355X, so a whole batch costs about 20 s of gpurun budget.
# Build here (hipcc cross-compiles without a GPU; the binaries travel with gpurun), run there:
#   bash tools/standalone_harnesses.sh build && gpurun --timeout 120 -- 'bash tools/standalone_harnesses.sh run'
set -u
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}; cd $R
F="--offload-arch=gfx950 -O3 -std=c++17 -I controlar_amd/csrc"
if [ "${1:-run}" = build ]; then
  for h in small_chain gemm_mid t_check hfuse_check ws_check queue_check kbench; do /opt/rocm/bin/hipcc $F experiments/$h.hip -o experiments/$h & done; wait; ls -la experiments | grep -v '\.hip\|README'
  exit 0
fi
O=$R/gpurun_out; mkdir -p $O
( timeout 30 experiments/small_chain 2 631;  timeout 30 experiments/small_chain 2 200; timeout 30 experiments/small_chain 8 631; timeout 30 experiments/small_chain 16 631 ) > $O/small_chain.txt 2>&1
timeout 30 experiments/gemm_mid > $O/gemm_mid.txt 2>&1
timeout 60 experiments/t_check > $O/t_check.txt 2>&1
timeout 60 experiments/ws_check > $O/ws_check.txt 2>&1
timeout 30 experiments/hfuse_check > $O/hfuse_check.txt 2>&1
tail -40 $O/small_chain.txt; cat $O/gemm_mid.txt; tail -30 $O/t_check.txt; tail -14 $O/ws_check.txt; cat $O/hfuse_check.txt
