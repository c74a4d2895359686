#!/bin/bash
# exact-mode decode step at the mean position: prefetch depth of the linears x chains (dev library)
R=${GRAFT_REPO_ROOT:-$(pwd)}; cd $R
for deep in 0 1 2; do
  timeout 600 python tools/exact_probe.py 384 2,3 509 20 CAR_F32_DEEP=$deep 2>&1 | grep -E "^\{|Error|error" | sed "s/^/deep$deep /"
done
