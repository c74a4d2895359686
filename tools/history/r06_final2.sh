#!/bin/bash
# round 6, last pass after a comment-only change in csrc (the build id changed): GPU suite, smoke, the profiling round (traces + PMC), the default bench line
set -u
O=gpurun_out; mkdir -p $O/r06_final2
timeout 1500 python -m pytest tests -x -q -m gpu > $O/r06_final2/pytest_gpu.txt 2>&1; tail -3 $O/r06_final2/pytest_gpu.txt
python __graft_entry__.py smoke > $O/r06_final2/smoke.txt 2>&1; tail -2 $O/r06_final2/smoke.txt
bash tools/profile_round.sh r06 > $O/r06_final2/profile_round.log 2>&1; tail -3 $O/r06_final2/profile_round.log
cp $O/pmc_decode_step.json profiles/pmc_decode_step.json 2>/dev/null; cp $O/pmc_decode_step_fp32.json profiles/pmc_decode_step_fp32.json 2>/dev/null
( time python bench.py --steps 3 --warmup 1 > $O/r06_final2/r06_bench_b768.json 2> $O/r06_final2/bench.err ) 2>> $O/r06_final2/bench.err
tail -4 $O/r06_final2/bench.err
cp profiles/pmc_decode_step.json $O/r06_final2/ 2>/dev/null; cp profiles/pmc_decode_step_fp32.json $O/r06_final2/ 2>/dev/null
python - <<'PY'
import json
d=json.loads(open('gpurun_out/r06_final2/r06_bench_b768.json').read().strip().splitlines()[-1])
print('bench', d['value'], d['ms_per_step'], 'frac', d['roofline']['frac'], 'traffic', d['roofline'].get('traffic'), 'exact', d.get('exact',{}).get('value'), {k:(v.get('roofline',{}) or {}).get('avg_launch_ms') for k,v in d.get('configs',{}).items() if isinstance(v,dict)})
PY
