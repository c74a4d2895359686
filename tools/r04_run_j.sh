ROOT=$(pwd); OUT=$ROOT/gpurun_out
for r in 2 8 32 64 128; do timeout 120 experiments/lat_probe $r 631 > $OUT/r04_lat_probe_v5_rows$r.txt 2>&1; cat $OUT/r04_lat_probe_v5_rows$r.txt | cut -c1-150 | head -9; done
for r in 32 128; do timeout 120 experiments/lat_probe $r 631 0 0 > $OUT/r04_lat_probe_v5_rows${r}_no_normx.txt 2>&1; cat $OUT/r04_lat_probe_v5_rows${r}_no_normx.txt | cut -c1-150 | head -11; done
