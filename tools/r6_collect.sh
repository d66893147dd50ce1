#!/bin/bash
# copy what tools/r6_artifacts.sh produced (gpurun_out/r6_final) into profiles/round6_* (tracked)
set -e
cd "$(dirname "$0")/.."
a=gpurun_out/r6_final; p=$a/prof
cp $p/kernel_stats_steady.csv profiles/round6_kernel_stats.csv
cp $p/kernel_stats_whole_run.csv profiles/round6_kernel_stats_whole_run.csv
cp $p/summary.md profiles/round6_summary.md
cp $p/queues.txt profiles/round6_queues.txt
cp $p/gaps.txt profiles/round6_gaps.txt
cp $p/timeline.txt profiles/round6_timeline.txt
cp $a/hbm_traffic.json profiles/round6_hbm_traffic.json
cp $a/step_traffic.txt profiles/round6_step_traffic.txt
for c in c2 c3 c4; do tail -1 $a/${c}_bench.json > profiles/round6_${c}_bench.json; done
cp $a/phase_timing.json profiles/round6_phase_timing.json
cut -c1-40 $a/ab_round5_vs_round6.txt > profiles/round6_ab_round5_vs_round6.txt
tail -3 $a/pytest_gpu.txt > profiles/round6_pytest_gpu.txt
ls -la profiles/round6_*
