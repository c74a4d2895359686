#!/bin/bash
# Round-end measurement batch (run on the GPU box through gpurun).  Writes everything under gpurun_out/.
set -x
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out
mkdir -p $O
cd $R
timeout 400 python bench.py > $O/final_bench_default.json 2> $O/final_bench_default.err
timeout 200 python bench.py --cfg-scale 4 --batch 1 --steps 1 --warmup 1 --no-cpu-baseline > $O/final_config2_cfg4_b1.json 2>/dev/null
timeout 300 python bench.py --cfg-scale 4 --batch 32 --steps 1 --warmup 1 --no-cpu-baseline --condition-type depth --adapter-size base > $O/final_config3_cfg4_b32_depth_base.json 2>/dev/null
timeout 300 python bench.py --cfg-scale 4 --batch 1 --steps 1 --warmup 1 --no-cpu-baseline --image-h 768 --image-w 512 > $O/final_config4_mr768x512.json 2>/dev/null
timeout 300 python bench.py --batch 8 --steps 1 --warmup 1 --no-cpu-baseline --weights-fp8 --condition-type hed --adapter-size base > $O/final_config5_fp8_b8.json 2>/dev/null
cd /tmp && export TMPDIR=/tmp
timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/pb -o b -- python $R/bench.py --batch 128 --steps 1 --warmup 0 --no-cpu-baseline > $O/final_prof.log 2>&1
cp /tmp/pb/b_kernel_stats.csv $O/final_bench_b128_kernel_stats.csv
python $R/tools/trace_summary.py /tmp/pb/b_kernel_trace.csv 1.0 > $O/final_bench_b128_trace_summary.txt
