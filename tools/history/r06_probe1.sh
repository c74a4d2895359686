#!/bin/bash
# round 6, first GPU pass: what a kernel boundary keeps (experiments/xk_cache), the small-batch layer timeline with the weight run-ahead kernel, HIP_FORCE_DEV_KERNARG
set -u
O=gpurun_out/r06_probe1; mkdir -p $O
L=experiments/lat_probe
timeout 300 experiments/xk_cache > $O/xk_cache.txt 2>&1
timeout 60 $L 2 631 > $O/lat_rows2_base.txt 2>&1
HIP_FORCE_DEV_KERNARG=1 timeout 60 $L 2 631 > $O/lat_rows2_devkernarg1.txt 2>&1
HIP_FORCE_DEV_KERNARG=0 timeout 60 $L 2 631 > $O/lat_rows2_devkernarg0.txt 2>&1
timeout 60 $L 2 631 0 1 4 > $O/lat_rows2_NL4.txt 2>&1
for g in 16 32 64 128; do timeout 60 $L 2 631 0 1 12 $g 2 > $O/lat_rows2_pf${g}_a2.txt 2>&1; done
for a in 1 3 4; do timeout 60 $L 2 631 0 1 12 32 $a > $O/lat_rows2_pf32_a$a.txt 2>&1; done
for r in 8 16 64; do timeout 60 $L $r 631 > $O/lat_rows${r}_base.txt 2>&1; timeout 60 $L $r 631 0 1 12 32 2 > $O/lat_rows${r}_pf32_a2.txt 2>&1; done
timeout 60 $L 2 1100 > $O/lat_rows2_pos1100_base.txt 2>&1
grep -H "instrumented chain" $O/lat_*.txt
tail -40 $O/xk_cache.txt
