#!/bin/bash
# were any waves context-switched (CWSR) while the coefficient kernel ran next to the GEMM stream?
out=$GRAFT_REPO_ROOT/gpurun_out/cwsr
rm -rf $out; mkdir -p $out
cd /tmp; export TMPDIR=/tmp
rocprofv3 --output-format csv --pmc SQ_WAVES_SAVED SQ_WAVES_RESTORED SQ_WAVES -d $out -o c -- python $GRAFT_REPO_ROOT/tools/probe/coef_under_gemm.py linear 4096,768,768 > $out/run.log 2>&1
tail -1 $out/run.log
python - <<PY
import csv, glob, collections
acc = collections.defaultdict(float)
for f in glob.glob("$out/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        acc[(r["Kernel_Name"][:48], r["Counter_Name"])] += float(r["Counter_Value"])
for k, v in sorted(acc.items()): print(k, v)
PY
find $out -name "*.csv" -size +2M -delete
