#!/bin/bash
set -u
O=gpurun_out/r06_probe9; mkdir -p $O
for r in 16 24 32 48 64 128; do
  timeout 60 experiments/lat_probe_r05 $r 631 > $O/r05_rows$r.txt 2>&1
  timeout 60 experiments/lat_probe $r 631 > $O/r06_rows$r.txt 2>&1
done
grep -H "instrumented chain" $O/*.txt
