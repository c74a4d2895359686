#!/bin/bash
# round 6, third GPU pass: kernarg-line prefetch + unconditional norm loads + hoisted epilogue operands + 16-wave wo / w2 + two-blocks-in-flight attention, with and without L2 run-ahead helpers
set -u
O=gpurun_out/r06_probe3; mkdir -p $O
timeout 300 experiments/kbench check > $O/kbench_check.txt 2>&1; tail -3 $O/kbench_check.txt; grep -c OK $O/kbench_check.txt; grep FAIL $O/kbench_check.txt | head -20
L=experiments/lat_probe
run() { local name=$1; shift; timeout 60 $L "$@" > $O/$name.txt 2>&1; }
run rows2_nopf_attn160 2 631 0 1 12 0 0 0 0
run rows2_nopf 2 631
run rows2_m1 2 631 0 1 12 1 192 160
run rows2_m2 2 631 0 1 12 2 192 160
run rows2_m2_l96 2 631 0 1 12 2 192 96
run rows2_m3 2 631 0 1 12 3 192 160
run rows2_pos200_nopf 2 200
run rows2_pos200_m2 2 200 0 1 12 2 192 160
run rows2_pos1100_nopf 2 1100
run rows2_pos1100_m2 2 1100 0 1 12 2 192 160
run rows8_nopf 8 631
run rows8_m2 8 631 0 1 12 2 96 160
run rows16_nopf 16 631
run rows64_nopf 64 631
grep -H "instrumented chain" $O/*.txt
cat $O/rows2_nopf.txt $O/rows2_m2.txt
