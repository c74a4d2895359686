#!/bin/bash
set -u
O=gpurun_out/r06_d; mkdir -p $O
python tools/twin_probe.py b 384 96 2>&1 | grep call
timeout 300 experiments/kbench check > $O/kbench_check.txt 2>&1; tail -1 $O/kbench_check.txt
timeout 2400 python -m pytest tests -x -q -m gpu > $O/pytest_gpu.txt 2>&1; tail -3 $O/pytest_gpu.txt
bash tools/r06_round_c.sh
