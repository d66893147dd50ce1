#!/bin/bash
# A/B of environment switches inside ONE gpurun call: tools/r4_ab.sh "NAME=ENV=VAL,ENV=VAL" ...   (NAME=- : no switch)
out=gpurun_out/r4_ab.txt
for spec in "$@"; do
  name=${spec%%=*}; envs=${spec#*=}
  ( if [ "$envs" != "-" ]; then IFS=','; for e in $envs; do export "$e"; done; unset IFS; fi
    python bench.py --steps 40 --warmup 8 --no-cpu-baseline --steady-steps 0 $R4_BENCH_ARGS 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('%-36s %7.2f img/s  %7.3f ms' % ('$name', d['value'], d['ms_per_step']))" ) | tee -a $out
done
