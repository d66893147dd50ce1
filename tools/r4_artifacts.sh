#!/bin/bash
# everything profiles/round4_* is made from, in one gpurun call (tools/r4_collect.sh copies the results into profiles/)
o=gpurun_out/r4_art; rm -rf $o; mkdir -p $o
python -m pytest tests -m gpu -x -q > $o/pytest_gpu.txt 2>&1; tail -3 $o/pytest_gpu.txt
bash tools/profile_round.sh r4final > /dev/null 2>&1
bash tools/pmc_traffic.sh > /dev/null 2>&1
python tools/pmc_step_traffic.py gpurun_out/pmc_traffic/rd gpurun_out/pmc_traffic/wr 7 > $o/step_traffic.txt 2>&1
bash tools/attn_bi_pmc.sh enc > $o/attn_bi_pmc_enc.txt 2>&1
python tools/attn_bi_bench.py enc dec cross 2>/dev/null > $o/attn_bi_bench_base.txt
ATTN_BENCH_LARGE=1 ATTN_BENCH_B=8 python tools/attn_bi_bench.py enc dec cross 2>/dev/null > $o/attn_bi_bench_large.txt
python bench.py > $o/c2_bench.json 2> $o/c2_bench.err
IFSEG_PHASE_TIMING=1 IFSEG_DRAIN_TIMING=1 python bench.py --steps 60 --warmup 10 --no-cpu-baseline --steady-steps 0 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(json.dumps({k: d[k] for k in ('value','ms_per_step','host_enqueue_ms_per_step','end_of_backward_wait_ms','phase_ms','phase_host_ms')}, indent=1))" > $o/phase_timing.json
python bench.py --config c3 --no-cpu-baseline > $o/c3_bench.json 2> $o/c3_bench.err
python bench.py --config c4 --no-cpu-baseline > $o/c4_bench.json 2> $o/c4_bench.err
for f in c2 c3 c4; do python -c "import json,sys; d=json.loads(open('$o/${f}_bench.json').read().strip().splitlines()[-1]); print('$f', d['value'], d['ms_per_step'], d['roofline']['kernel'], d['roofline']['frac'])"; done
