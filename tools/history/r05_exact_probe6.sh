#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}; cd $R
for ch in 16 32 48 96; do timeout 300 python tools/exact_probe.py 384 3 509 8 CAR_ENC_CHUNK=$ch 2>&1 | grep -E "^\{|rror" | sed "s/^/chunk$ch /"; done
timeout 300 python tools/exact_probe.py 384 3 509 20 CAR_KV_UNCACHED=1 2>&1 | grep -E "^\{|rror" | sed "s/^/kv_uncached /"
timeout 400 python bench.py --precision fp32 --steps 2 --warmup 1 --no-cpu-baseline > gpurun_out/r05_bench_fp32_v3.json 2> gpurun_out/r05_bench_fp32_v3.err; python -c "
import json; d=json.load(open('gpurun_out/r05_bench_fp32_v3.json')); print('plain', d['value'], d['roofline']['avg_launch_ms'], d['config'].get('stage_ms'), d['config']['self_check'])"
timeout 400 python bench.py --precision fp32 --steps 2 --warmup 1 --no-cpu-baseline --overlap-vq > gpurun_out/r05_bench_fp32_v3_ovq.json 2> gpurun_out/r05_bench_fp32_v3_ovq.err; python -c "
import json; d=json.load(open('gpurun_out/r05_bench_fp32_v3_ovq.json')); print('overlap-vq', d['value'], d['roofline']['avg_launch_ms'], d['config'].get('stage_ms'), d['config']['self_check'])"
