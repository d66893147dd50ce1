#!/bin/bash
# tools/r4_cfg_prof.sh <c2|c3|c4>: rocprofv3 kernel statistics of a bench.py configuration -> gpurun_out/r4_<cfg>_trace/summary.md
cfg=${1:-c3}
cd /tmp; export TMPDIR=/tmp
o=$GRAFT_REPO_ROOT/gpurun_out/r4_${cfg}_trace; rm -rf $o; mkdir -p $o
rocprofv3 --kernel-trace --stats --output-format csv -d $o/trace -o t -- python $GRAFT_REPO_ROOT/bench.py --config $cfg --steps 4 --warmup 2 --no-cpu-baseline --steady-steps 0 > $o/log.txt 2>&1
python $GRAFT_REPO_ROOT/tools/prof_summary.py $o/trace 9 "$cfg under rocprofv3 (bench.py --steps 4 --warmup 2: 9 steps incl. the profiling passes)" > $o/summary.md
head -44 $o/summary.md
find $o/trace -name "*.csv" -size +3M -delete; find $o -name "*.db" -delete
