#!/bin/bash
# exact-mode decode step at the mean position for larger batches (per-image cost)
R=${GRAFT_REPO_ROOT:-$(pwd)}; cd $R
timeout 600 python tools/exact_probe.py 480 3,2 509 20 2>&1 | grep -E "^\{|Error|error|rror"
timeout 600 python tools/exact_probe.py 512 4,2 509 20 2>&1 | grep -E "^\{|Error|error|rror"
