#!/bin/bash
# everything profiles/round6_* of the final state is made from, in one gpurun call (tools/r6_collect.sh copies it into profiles/)
# the tree that runs is commit 013818c (+ this script)
o=gpurun_out/r6_final; rm -rf $o; mkdir -p $o
R=$GRAFT_REPO_ROOT
timeout 2400 python -m pytest tests -m gpu -x -q > $o/pytest_gpu.txt 2>&1; tail -3 $o/pytest_gpu.txt
# HBM traffic first: the bench lines below quote it
bash tools/pmc_traffic.sh > /dev/null 2>&1
python tools/pmc_step_traffic.py gpurun_out/pmc_traffic/rd gpurun_out/pmc_traffic/wr > $o/step_traffic.txt 2>&1; head -3 $o/step_traffic.txt
python - <<PY
import json
d = json.load(open("gpurun_out/hbm_traffic.json")); d["_collected_at"] = "013818c"
json.dump(d, open("profiles/round6_hbm_traffic.json", "w"), indent=1)
PY
cp profiles/round6_hbm_traffic.json $o/hbm_traffic.json
p=$R/$o/prof; mkdir -p $p
( cd /tmp; export TMPDIR=/tmp
  rocprofv3 --kernel-trace --stats --output-format csv -d $p/trace -o t -- python $R/bench.py --steps 8 --warmup 4 --no-cpu-baseline --steady-steps 0 > $p/bench_under_profiler.log 2>&1 )
tr=$(find $p/trace -name "*kernel_trace.csv" | head -1)
python tools/steady_stats.py $tr 5 11 $p/kernel_stats_steady.csv > $p/summary.md
cp $(find $p/trace -name "*kernel_stats.csv" | head -1) $p/kernel_stats_whole_run.csv
python tools/queue_kernels.py $tr 5 11 > $p/queues.txt
python tools/queue_gaps.py $tr 5 11 > $p/gaps.txt
python tools/step_timeline.py $tr 6 > $p/timeline.txt
find $p/trace -name "*.csv" -size +3M -delete; find $p -name "*.db" -delete
python bench.py > $o/c2_bench.json 2> $o/c2_bench.err
IFSEG_LAB=1 IFSEG_PHASE_TIMING=1 IFSEG_DRAIN_TIMING=1 python bench.py --lab --steps 60 --warmup 10 --no-cpu-baseline --steady-steps 0 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(json.dumps({k: d.get(k) for k in ('value','ms_per_step','host_enqueue_ms_per_step','end_of_backward_wait_ms','phase_ms','phase_host_ms')}, indent=1))" > $o/phase_timing.json
python bench.py --config c3 --no-cpu-baseline > $o/c3_bench.json 2> $o/c3_bench.err
python bench.py --config c4 --no-cpu-baseline > $o/c4_bench.json 2> $o/c4_bench.err
for f in c2 c3 c4; do python -c "import json,sys; d=json.loads(open('$o/${f}_bench.json').read().strip().splitlines()[-1]); print('$f', d['value'], d['ms_per_step'], d['roofline']['kernel'], d['roofline']['frac'])"; done
# the round in one line pair: round 5's final tree (commit 7620dff, tools/bin/base) against this one, same box, interleaved
REPS=4 STEPS=30 bash tools/r6_ab2.sh > $o/ab_round5_vs_round6.txt 2>&1; cut -c1-40 $o/ab_round5_vs_round6.txt
