#!/bin/bash
# round 6: the GPU suite, the default bench line and a steady-state kernel trace of the current tree, in one gpurun call
o=gpurun_out/r6_art; rm -rf $o; mkdir -p $o
R=$GRAFT_REPO_ROOT
timeout 2400 python -m pytest tests -m gpu -x -q > $o/pytest_gpu.txt 2>&1; tail -3 $o/pytest_gpu.txt
python bench.py > $o/c2_bench.json 2> $o/c2_bench.err; cut -c1-400 $o/c2_bench.json
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
head -12 $p/summary.md | cut -c1-160
