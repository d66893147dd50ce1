#!/bin/bash
# round 6, call 1: dW-group release points -- bit-equality, same-box A/B, timeline of one variant
o=gpurun_out/r6_call1; rm -rf $o; mkdir -p $o
R=$GRAFT_REPO_ROOT
timeout 900 python tools/r6_dw_release_check.py 1 2 3 > $o/release_check.txt 2>&1; tail -4 $o/release_check.txt
REPS=2 STEPS=30 bash tools/r6_ab.sh "base:" "rel1:IFSEG_DW_RELEASE=1" "rel2:IFSEG_DW_RELEASE=2" "rel3:IFSEG_DW_RELEASE=3" > $o/ab.txt 2>&1
cat $o/ab.txt | cut -c1-60
for v in 0 2 3; do
p=$R/gpurun_out/r6_call1/prof_rel$v; mkdir -p $p
( cd /tmp; export TMPDIR=/tmp
  IFSEG_LAB=1 IFSEG_DW_RELEASE=$v rocprofv3 --kernel-trace --stats --output-format csv -d $p/trace -o t -- python $R/bench.py --lab --steps 8 --warmup 4 --no-cpu-baseline --steady-steps 0 > $p/bench_under_profiler.log 2>&1 )
tr=$(find $p/trace -name "*kernel_trace.csv" | head -1)
python tools/steady_stats.py $tr 5 11 $p/kernel_stats_steady.csv > $p/summary.md
python tools/step_timeline.py $tr 6 > $p/timeline.txt
python tools/queue_kernels.py $tr 5 11 > $p/queues.txt
rm -rf $p/trace
done
