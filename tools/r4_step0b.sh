#!/bin/bash
# Round 4, step 0b: where does the step's time go when kernels of several queues run side by side?
out=gpurun_out/r4_step0b.txt; : > $out
NB=$PWD/ifseg_amd/lib/variants/nobias_all.so
b() { python bench.py --steps 40 --warmup 8 --no-cpu-baseline --steady-steps 0 $2 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('%-40s %7.2f img/s  %7.3f ms' % ('$1', d['value'], d['ms_per_step']))" >> $out 2>&1; }
b baseline
IFSEG_NO_OVERLAP=1 b no_overlap
IFSEG_NO_OVERLAP=1 IFSEG_LIB=$NB b no_overlap+nobias
IFSEG_NO_OVERLAP=1 b no_overlap+no_prefetch --no-prefetch
IFSEG_NO_OVERLAP=1 IFSEG_LIB=$NB b no_overlap+no_prefetch+nobias --no-prefetch
b no_prefetch --no-prefetch
IFSEG_LIB=$NB b no_prefetch+nobias --no-prefetch
IFSEG_EXP_SKIP=dw b skip_dw
IFSEG_EXP_SKIP=dw IFSEG_LIB=$NB b skip_dw+nobias
IFSEG_DQ_SERIAL=1 b dq_serial
IFSEG_DQ_SERIAL=1 IFSEG_LIB=$NB b dq_serial+nobias
cat $out
# kernel trace of the step with both libraries: in-step kernel durations
cd /tmp; export TMPDIR=/tmp
for v in base nobias; do
  o=$GRAFT_REPO_ROOT/gpurun_out/r4_trace_$v; rm -rf $o; mkdir -p $o
  if [ $v = nobias ]; then export IFSEG_LIB=$NB; fi
  rocprofv3 --kernel-trace --stats --output-format csv -d $o/trace -o t -- python $GRAFT_REPO_ROOT/bench.py --steps 5 --warmup 2 --no-cpu-baseline --steady-steps 0 > $o/log.txt 2>&1
  python $GRAFT_REPO_ROOT/tools/queue_kernels.py $(find $o/trace -name "*kernel_trace.csv" | head -1) 2 6 > $o/queues.txt
  find $o/trace -name "*.csv" -size +3M -delete; find $o -name "*.db" -delete
done
