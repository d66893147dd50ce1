#!/bin/bash
# A/B of settings in the training step (one box, one call): prints ms_per_step per setting.  usage: r5_ab.sh "NAME=ENV=VAL,ENV=VAL" ...
cd $GRAFT_REPO_ROOT
for rep in 1 2; do
for s in "$@"; do
  name=${s%%:*}; envs=${s#*:}
  ( IFS=','; for kv in $envs; do [ -n "$kv" ] && export "$kv"; done
    out=$(timeout 600 python bench.py --steps ${STEPS:-20} --warmup 6 2>/dev/null | tail -1)
    echo "$name $(echo "$out" | python -c 'import sys,json; d=json.loads(sys.stdin.read()); print(d["ms_per_step"], d["value"], d.get("host_enqueue_ms_per_step"))')" )
done
done
