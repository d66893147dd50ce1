#!/bin/bash
# same-box A/B of two TREES (builds), interleaved: the working tree against tools/bin/base (an export of an earlier commit with
# its own library).  usage: r6_ab2.sh [bench args]   env: REPS, STEPS
cd $GRAFT_REPO_ROOT
for rep in $(seq 1 ${REPS:-3}); do
  for t in base new; do
    d=$GRAFT_REPO_ROOT; [ $t = base ] && d=$GRAFT_REPO_ROOT/tools/bin/base
    out=$(cd $d && timeout 600 python bench.py --steps ${STEPS:-30} --warmup 6 --no-cpu-baseline --steady-steps 0 "$@" 2>/dev/null | tail -1)
    echo "$t $(echo "$out" | python -c 'import sys,json; d=json.loads(sys.stdin.read()); print(d["ms_per_step"], d["value"], json.dumps(d.get("kernel_families_ms_per_step")))')"
  done
done
