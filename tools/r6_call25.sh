#!/bin/bash
# round 6, call 25: the clip + Adam launch alone -- grid size sweep (library variants of optim.hip), two repetitions
o=gpurun_out/r6_call25; mkdir -p $o
( for rep in 1 2; do python tools/adam_bench.py | head -1
  for v in g256 g512 g768 g1024 g1280 g1536 g4096; do IFSEG_LIB=$GRAFT_REPO_ROOT/ifseg_amd/lib/variants/adam_$v.so python tools/adam_bench.py | head -1; done; done ) > $o/adam_grid.txt 2>&1
grep adam $o/adam_grid.txt
