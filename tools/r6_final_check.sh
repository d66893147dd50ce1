#!/bin/bash
# the GPU suite and the default bench line on the round's last commit (the row-map host change came after the artifacts call)
o=gpurun_out/r6_final_check; rm -rf $o; mkdir -p $o
timeout 2400 python -m pytest tests -m gpu -x -q > $o/pytest_gpu.txt 2>&1; tail -3 $o/pytest_gpu.txt
python bench.py > $o/c2_bench.json 2> $o/c2_bench.err; cut -c1-200 $o/c2_bench.json
python -c "import __graft_entry__ as g; g.smoke()" > $o/smoke.txt 2>&1; tail -2 $o/smoke.txt
