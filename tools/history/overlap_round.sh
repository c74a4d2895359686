#!/bin/bash
# GPU batch (gpurun): schedule-knob sweep of the multi-chain decode loop + the parity tests of the changed path under the phase offset.
set -u
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out; mkdir -p $O; cd $R
timeout 480 python tools/overlap_sweep.py 768 1024 > $O/overlap_sweep.txt 2>&1
tail -30 $O/overlap_sweep.txt
timeout 200 python -m pytest tests/test_parity_gpu.py -x -q -m gpu -k "schedule_knobs or two_chain or cfg_large" > $O/pytest_knobs.txt 2>&1
tail -5 $O/pytest_knobs.txt
CAR_PHASE_OFFSET=1 timeout 400 python -m pytest tests/test_bench_shapes_gpu.py -x -q -m gpu -k "two_chains" > $O/pytest_phase_bench_shapes.txt 2>&1
tail -5 $O/pytest_phase_bench_shapes.txt
