#!/bin/bash
# round 6: early launch (two-stream chain with arrival counters) in the small-batch layer
set -u
O=gpurun_out/r06_probe5; mkdir -p $O
true
L=experiments/lat_probe
run() { local name=$1; shift; timeout 60 $L "$@" > $O/$name.txt 2>&1; echo "$name rc=$?"; }
run rows2_base 2 631
run rows2_ovl 2 631 0 1 12 0 0 0 1 1
run rows2_ovl_pos200 2 200 0 1 12 0 0 0 1 1
run rows2_ovl_pos1100 2 1100 0 1 12 0 0 0 1 1
run rows8_base 8 631
run rows8_ovl 8 631 0 1 12 0 0 0 1 1
run rows1_ovl 1 631 0 1 12 0 0 0 1 1
grep -H "instrumented chain\|error word" $O/*.txt
cat $O/rows2_ovl.txt
