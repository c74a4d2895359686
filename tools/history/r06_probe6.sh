#!/bin/bash
set -u
O=gpurun_out/r06_probe6; mkdir -p $O
timeout 300 experiments/kbench check > $O/kbench_check.txt 2>&1; tail -1 $O/kbench_check.txt; grep FAIL $O/kbench_check.txt | head
L=experiments/lat_probe
run() { local name=$1; shift; timeout 60 $L "$@" > $O/$name.txt 2>&1; }
run rows2_nopf 2 631
run rows2_m2 2 631 0 1 12 2 192 160
run rows2_m2_a128 2 631 0 1 12 2 128 160
run rows2_pos200_m2 2 200 0 1 12 2 192 160
run rows2_pos1100_m2 2 1100 0 1 12 2 192 160
run rows8_nopf 8 631
run rows8_m2 8 631 0 1 12 2 96 160
run rows16_nopf 16 631
run rows64_nopf 64 631
L=experiments/lat_probe_el
run rows2_el 2 631 0 1 12 0 0 0 1 1
grep -H "instrumented chain" $O/*.txt
cat $O/rows2_m2.txt | tail -22
