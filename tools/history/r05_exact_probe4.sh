#!/bin/bash
# exact-mode decode step by POSITION and chain count (dev library)
R=${GRAFT_REPO_ROOT:-$(pwd)}; cd $R
for skip in 0 100 250 700 1000; do
  timeout 600 python tools/exact_probe.py 384 1,2,3 $skip 20 2>&1 | grep -E "^\{|Error|error"
done
