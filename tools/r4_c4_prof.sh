#!/bin/bash
cd /tmp; export TMPDIR=/tmp
o=$GRAFT_REPO_ROOT/gpurun_out/r4_c4_trace; rm -rf $o; mkdir -p $o
rocprofv3 --kernel-trace --stats --output-format csv -d $o/trace -o t -- python $GRAFT_REPO_ROOT/bench.py --config c4 --steps 4 --warmup 2 --no-cpu-baseline --steady-steps 0 > $o/log.txt 2>&1
python $GRAFT_REPO_ROOT/tools/prof_summary.py $o/trace 6 "C4 under rocprofv3" > $o/summary.md
head -30 $o/summary.md
find $o/trace -name "*.csv" -size +3M -delete; find $o -name "*.db" -delete
