#!/bin/bash
# HBM traffic per kernel family: two separate --pmc passes (never combined with trace domains), then tools/pmc_traffic.py
out=$GRAFT_REPO_ROOT/gpurun_out/pmc_traffic
rm -rf $out; mkdir -p $out
cd /tmp; export TMPDIR=/tmp
rocprofv3 --output-format csv --pmc FETCH_SIZE -d $out/rd -o rd -- python $GRAFT_REPO_ROOT/bench.py --steps 2 --warmup 1 --no-cpu-baseline --steady-steps 0 > $out/rd.log 2>&1
rocprofv3 --output-format csv --pmc WRITE_SIZE -d $out/wr -o wr -- python $GRAFT_REPO_ROOT/bench.py --steps 2 --warmup 1 --no-cpu-baseline --steady-steps 0 > $out/wr.log 2>&1
python $GRAFT_REPO_ROOT/tools/pmc_traffic.py $out/rd $out/wr > $GRAFT_REPO_ROOT/gpurun_out/hbm_traffic.json
find $out -name "*.csv" -size +4M -delete; find $out -name "*.db" -delete
