#!/bin/bash
# tools/profile_round.sh <tag>: rocprofv3 kernel trace + stats of bench.py (5+2 steps + the 3-step GEMM pass = 10 steps)
# and the HBM-traffic PMC passes; everything lands in gpurun_out/prof_<tag>/ (copy what should be judged into profiles/)
tag=${1:-x}
out=$GRAFT_REPO_ROOT/gpurun_out/prof_$tag
rm -rf $out; mkdir -p $out
cd /tmp; export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $out/trace -o t -- python $GRAFT_REPO_ROOT/bench.py --steps 5 --warmup 2 --no-cpu-baseline --steady-steps 0 > $out/bench_under_profiler.log 2>&1
python $GRAFT_REPO_ROOT/tools/prof_summary.py $out/trace 10 "rocprofv3 --kernel-trace --stats, bench.py --steps 5 --warmup 2 (10 steps incl. the 3-step GEMM pass), MI355X, round 4 ($tag)" > $out/summary.md
cp $(find $out/trace -name "*kernel_stats.csv" | head -1) $out/kernel_stats.csv
python $GRAFT_REPO_ROOT/tools/queue_kernels.py $(find $out/trace -name "*kernel_trace.csv" | head -1) 2 6 > $out/queues.txt
python $GRAFT_REPO_ROOT/tools/queue_gaps.py $(find $out/trace -name "*kernel_trace.csv" | head -1) 2 6 > $out/gaps.txt
python $GRAFT_REPO_ROOT/tools/step_timeline.py $(find $out/trace -name "*kernel_trace.csv" | head -1) 3 > $out/timeline.txt
find $out/trace -name "*.csv" -size +3M -delete; find $out -name "*.db" -delete
