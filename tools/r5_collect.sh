#!/bin/bash
# copy what tools/r5_artifacts.sh produced (gpurun_out/) into profiles/round5_* (tracked)
set -e
cd "$(dirname "$0")/.."
a=gpurun_out/r5_art; p=gpurun_out/prof_r5
cp $p/kernel_stats_steady.csv profiles/round5_kernel_stats.csv
cp $p/kernel_stats_whole_run.csv profiles/round5_kernel_stats_whole_run.csv
cp $p/summary.md profiles/round5_summary.md
cp $p/queues.txt profiles/round5_queues.txt
cp $p/gaps.txt profiles/round5_gaps.txt
cp $p/timeline.txt profiles/round5_timeline.txt
cp gpurun_out/hbm_traffic.json profiles/round5_hbm_traffic.json
cp $a/step_traffic.txt profiles/round5_step_traffic.txt
for c in c2 c3 c4; do tail -1 $a/${c}_bench.json > profiles/round5_${c}_bench.json; done
cp $a/phase_timing.json profiles/round5_phase_timing.json
tail -3 $a/pytest_gpu.txt > profiles/round5_pytest_gpu.txt
ls -la profiles/round5_*
