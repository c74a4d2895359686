#!/bin/bash
# round 6, second GPU pass: L2 run-ahead helper workgroups in the small-batch layer (experiments/lat_probe modes 1-3)
set -u
O=gpurun_out/r06_probe2; mkdir -p $O
L=experiments/lat_probe
run() { local name=$1; shift; timeout 60 $L "$@" > $O/$name.txt 2>&1; }
run rows2_base 2 631
run rows2_m3_a192 2 631 0 1 12 3 192
run rows2_m1_a192_l160 2 631 0 1 12 1 192 160
run rows2_m2_a192_l160 2 631 0 1 12 2 192 160
for a in 64 128 208; do run rows2_m1_a${a}_l160 2 631 0 1 12 1 $a 160; done
for l in 64 96 176; do run rows2_m2_a192_l$l 2 631 0 1 12 2 192 $l; done
run rows2_pos200_base 2 200
run rows2_pos200_m2 2 200 0 1 12 2 192 160
run rows2_pos1100_m2 2 1100 0 1 12 2 192 160
run rows8_base 8 631
run rows8_m2 8 631 0 1 12 2 96 160
run rows8_m1 8 631 0 1 12 1 96 160
grep -H "instrumented chain" $O/*.txt
cat $O/rows2_m2_a192_l160.txt
