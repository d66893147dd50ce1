#!/bin/bash
# A/B of laboratory settings in the training step (one box, one call): ms_per_step per setting, REPS repetitions interleaved.
# usage: r6_ab.sh "NAME:ENV=VAL,ENV=VAL" ...      (IFSEG_LAB=1 is set for every setting that names a variable)
cd $GRAFT_REPO_ROOT
for rep in $(seq 1 ${REPS:-2}); do
for s in "$@"; do
  name=${s%%:*}; envs=${s#*:}
  ( IFS=','; for kv in $envs; do [ -n "$kv" ] && export "$kv" && export IFSEG_LAB=1; done
    out=$(timeout 600 python bench.py --lab --steps ${STEPS:-30} --warmup 6 --no-cpu-baseline --steady-steps 0 ${BENCH_ARGS} 2>/dev/null | tail -1)
    echo "$name $(echo "$out" | python -c 'import sys,json; d=json.loads(sys.stdin.read()); print(d["ms_per_step"], d["value"], d.get("host_enqueue_ms_per_step"), json.dumps(d.get("kernel_families_ms_per_step")))')" )
done
done
