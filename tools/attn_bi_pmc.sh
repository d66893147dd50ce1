#!/bin/bash
# instruction-mix / stall counters of the batch-inner attention kernels alone (tools/attn_bi_bench.py): separate --pmc passes
out=$GRAFT_REPO_ROOT/gpurun_out/attn_bi_pmc
rm -rf $out; mkdir -p $out
cd /tmp; export TMPDIR=/tmp
i=0
for set in "SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_INSTS_SALU" "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_BUSY_CU_CYCLES SQ_WAVE_CYCLES" "SQ_VALU_MFMA_BUSY_CYCLES SQ_LDS_BANK_CONFLICT SQ_WAIT_INST_LDS SQ_WAIT_ANY" "SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VMEM SQ_LDS_IDX_ACTIVE" "GRBM_GUI_ACTIVE SQ_INSTS_VALU_TRANS_F32"; do
  i=$((i+1))
  rocprofv3 --output-format csv --pmc $set -d $out/p$i -o p -- python $GRAFT_REPO_ROOT/tools/attn_bi_bench.py ${1:-enc} > $out/p$i.log 2>&1
done
python - <<PY
import csv, glob, collections, re
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob("$out/p*/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        k = re.search(r"(attn_\w+kernel)", r["Kernel_Name"])
        if k: acc[k.group(1)][r["Counter_Name"]].append(float(r["Counter_Value"]))
for k, d in acc.items():
    print(k)
    for c, v in sorted(d.items()):
        print("   %-32s %14.0f  (avg over %d launches)" % (c, sum(v) / len(v), len(v)))
PY
find $out -name "*.csv" -size +2M -delete; find $out -name "*.db" -delete
