#!/bin/bash
# copy what tools/r4_artifacts.sh produced (gpurun_out/) into profiles/round4_* (tracked)
set -e
cd "$(dirname "$0")/.."
a=gpurun_out/r4_art; p=gpurun_out/prof_r4final
cp $p/kernel_stats.csv profiles/round4_kernel_stats.csv
cp $p/summary.md profiles/round4_summary.md
cp $p/queues.txt profiles/round4_queues.txt
cp $p/gaps.txt profiles/round4_gaps.txt
cp $p/timeline.txt profiles/round4_timeline.txt
cp gpurun_out/hbm_traffic.json profiles/round4_hbm_traffic.json
cp $a/step_traffic.txt profiles/round4_step_traffic.txt
cp $a/attn_bi_pmc_enc.txt profiles/round4_attn_pmc_enc.txt
cp $a/attn_bi_bench_base.txt profiles/round4_attn_bi_bench_base.txt
cp $a/attn_bi_bench_large.txt profiles/round4_attn_bi_bench_large.txt
for c in c2 c3 c4; do tail -1 $a/${c}_bench.json > profiles/round4_${c}_bench.json; done
cp $a/phase_timing.json profiles/round4_phase_timing.json
tail -3 $a/pytest_gpu.txt > profiles/round4_pytest_gpu.txt
cp gpurun_out/r4_step0.txt profiles/round4_step0_bound_kernels.txt
cp gpurun_out/r4_step0b.txt profiles/round4_step0_bound_step.txt
cp gpurun_out/r4_ab.txt profiles/round4_ab_log.txt
cp gpurun_out/r4_bi_exp2.txt profiles/round4_attn_bi_ablation.txt
ls -la profiles/round4_*
