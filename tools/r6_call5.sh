#!/bin/bash
# round 6, call 5: the full GPU suite after the clean-up + the resized-grid training test; bench line
o=gpurun_out/r6_call5; rm -rf $o; mkdir -p $o
timeout 3000 python -m pytest tests -m gpu -x -q -s > $o/pytest_gpu.txt 2>&1; tail -6 $o/pytest_gpu.txt
grep -n "resized-grid training\|gradients behind" $o/pytest_gpu.txt
python bench.py --steps 30 --warmup 6 --no-cpu-baseline --steady-steps 0 > $o/bench.json 2> $o/bench.err; cut -c1-200 $o/bench.json
