#!/bin/bash
# round 6, call 10: where SegOFA-Large (C4) spends its step -- kernel trace of the steady state + stand-alone GEMM times at its shapes
o=gpurun_out/r6_call10; rm -rf $o; mkdir -p $o
R=$GRAFT_REPO_ROOT
( cd /tmp; export TMPDIR=/tmp
  rocprofv3 --kernel-trace --stats --output-format csv -d $R/$o/trace -o t -- python $R/bench.py --config c4 --steps 4 --warmup 3 --no-cpu-baseline --steady-steps 0 > $R/$o/bench_c4_under_profiler.log 2>&1 )
tr=$(find $R/$o/trace -name "*kernel_trace.csv" | head -1)
python tools/steady_stats.py $tr 4 8 $o/c4_kernel_stats_steady.csv > $o/c4_summary.md
python tools/queue_kernels.py $tr 4 8 > $o/c4_queues.txt
rm -rf $o/trace
head -30 $o/c4_summary.md | cut -c1-150
python tools/gemm_c4_bench.py > $o/gemm_c4.txt 2>&1; cat $o/gemm_c4.txt
