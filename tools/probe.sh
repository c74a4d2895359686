#!/bin/bash
# tools/probe.sh <target> [out-dir] — the standalone measurement harnesses (experiments/*.hip) as named targets; run from the repo root on the GPU box
# (`gpurun -- 'bash tools/probe.sh small-chain'`).  Builds what it needs (hipcc, ~1-8 min per harness: they include the whole of decode2.hip), writes into
# gpurun_out/<target>/.  Copy what is to be judged into profiles/.
#   small-chain   lat_probe: phase timeline of a decode layer at 2 / 8 / 16 rows, with and without the L2 run-ahead helpers, three positions
#   mid-chain     lat_probe at 24 / 32 / 48 / 64 / 128 rows (the NORM == 2 regime and above)
#   early-launch  lat_probe built with -DCAR_EARLY_LAUNCH: two streams + arrival counters (measured, not shipped)
#   caches        xk_cache (what a kernel boundary keeps: L2 / Infinity Cache) and xk_fresh (first-load latency by what a kernel reads)
#   kbench        correctness of every decode2.hip kernel / tile configuration against host references (+ timing with `kbench-perf`)
#   twins         tools/twin_probe.py: two chains, free-running twin rows (GPT-B, 384 sequences)
set -u
T=${1:-small-chain}; O=${2:-gpurun_out/$T}; mkdir -p $O
H="hipcc --offload-arch=gfx950 -O3 -std=c++17 -I controlar_amd/csrc"
need() { local bin=$1; shift; [ -x experiments/$bin ] && [ experiments/$bin -nt controlar_amd/csrc/decode2.hip ] || $H "$@" -o experiments/$bin; }
lat() { local name=$1; shift; timeout 60 experiments/lat_probe "$@" > $O/$name.txt 2>&1; grep -H "instrumented chain" $O/$name.txt; }
case $T in
  small-chain)
    need lat_probe -DCAR_STAMP experiments/lat_probe.hip
    for pos in 200 631 1100; do lat rows2_pos${pos}_no_runahead 2 $pos; lat rows2_pos${pos} 2 $pos 0 1 12 2 192 160; done
    lat rows8_no_runahead 8 631; lat rows8 8 631 0 1 12 2 96 160; lat rows16 16 631 ;;
  mid-chain)
    need lat_probe -DCAR_STAMP experiments/lat_probe.hip
    for r in 24 32 48 64 128; do lat rows$r $r 631; done ;;
  early-launch)
    need lat_probe_el -DCAR_STAMP -DCAR_EARLY_LAUNCH experiments/lat_probe.hip
    for r in 2 8; do timeout 60 experiments/lat_probe_el $r 631 0 1 12 0 0 0 1 1 > $O/rows$r.txt 2>&1; grep -H "instrumented chain\|error word" $O/rows$r.txt; done ;;
  caches)
    need xk_cache experiments/xk_cache.hip; need xk_fresh experiments/xk_fresh.hip
    timeout 300 experiments/xk_cache | tee $O/xk_cache.txt; timeout 120 experiments/xk_fresh | tee $O/xk_fresh.txt ;;
  kbench|kbench-perf)
    need kbench experiments/kbench.hip
    if [ $T = kbench ]; then timeout 600 experiments/kbench check > $O/kbench_check.txt 2>&1; else timeout 900 experiments/kbench > $O/kbench.txt 2>&1; fi; tail -2 $O/kbench*.txt ;;
  twins)
    python tools/twin_probe.py b 384 96 2>&1 | grep call | tee $O/twins.txt ;;
  *) echo "unknown target $T"; exit 2 ;;
esac
