#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}; cd $R
experiments/f32_check quick 2>&1 | grep -E "FAIL|DIFFER|rejected|all checks|FAILED"
for rc in 22 1212 1114 1118; do
  timeout 300 python tools/exact_probe.py 384 3,2 509 20 CAR_F32_QKV_TILED_FROM=96 CAR_F32_RESID_CFG=$rc 2>&1 | grep -E "^\{|rror" | sed "s/^/qkv_tiled resid$rc /"
done
