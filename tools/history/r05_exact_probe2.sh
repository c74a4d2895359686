#!/bin/bash
# exact-mode decode step at the mean position: attention form x chains x linear priority (dev library)
R=${GRAFT_REPO_ROOT:-$(pwd)}; cd $R
for form in 3 4; do for prio in 0 1; do
  timeout 600 python tools/exact_probe.py 384 2,3,4,6 509 20 CAR_ATTN_F32_FORM=$form CAR_LINEAR_PRIO=$prio 2>&1 | grep -E "^\{|Error|error" | sed "s/^/form$form prio$prio /"
done; done
