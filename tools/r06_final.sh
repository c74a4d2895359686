#!/bin/bash
# round 6 final pass on the shipped build: GPU suite, harness tables, the profiling round (traces + PMC), the default bench line
set -u
O=gpurun_out; mkdir -p $O/r06_final
timeout 2400 python -m pytest tests -x -q -m gpu > $O/r06_final/pytest_gpu.txt 2>&1; tail -3 $O/r06_final/pytest_gpu.txt
python __graft_entry__.py smoke > $O/r06_final/smoke.txt 2>&1; tail -3 $O/r06_final/smoke.txt
bash tools/probe.sh small-chain $O/r06_final/small > $O/r06_final/small_chain.log 2>&1
bash tools/probe.sh mid-chain $O/r06_final/mid > $O/r06_final/mid_chain.log 2>&1
bash tools/probe.sh kbench $O/r06_final/kbench > $O/r06_final/kbench.log 2>&1; tail -1 $O/r06_final/kbench.log
bash tools/profile_round.sh r06 > $O/r06_final/profile_round.log 2>&1
cp $O/pmc_decode_step.json profiles/pmc_decode_step.json 2>/dev/null; cp $O/pmc_decode_step_fp32.json profiles/pmc_decode_step_fp32.json 2>/dev/null
( time python bench.py --steps 3 --warmup 1 > $O/r06_final/r06_bench_b768.json 2> $O/r06_final/bench.err ) 2>> $O/r06_final/bench.err
tail -4 $O/r06_final/bench.err
cp profiles/pmc_decode_step.json $O/r06_final/ 2>/dev/null; cp profiles/pmc_decode_step_fp32.json $O/r06_final/ 2>/dev/null
cat $O/r06_final/small_chain.log $O/r06_final/mid_chain.log | grep instrumented
