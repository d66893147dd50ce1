#!/bin/bash
# tools/sweep_env.sh VAR v1 v2 ... : bench.py (20 steps, no CPU baseline) once per value of the environment variable VAR
var=$1; shift
for v in "$@"; do
  env $var=$v python bench.py --no-cpu-baseline --steps 60 --warmup 10 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$var=$v', d['value'], 'img/s', d['ms_per_step'], 'ms', d['kernel_families_ms_per_step'])"
done
