#!/bin/bash
# PMC counters of single GEMM configurations through the laboratory binary.  usage: tools/pmc_lab.sh "<cfgs>" <layout> <M> <N> <K> [epi]
# (separate passes per counter group, no tracing domains beside --pmc)
R=$GRAFT_REPO_ROOT; out=$R/gpurun_out/pmc_lab; mkdir -p $out
cd /tmp; export TMPDIR=/tmp
tag="$2_$3_$4_$5"
run() { # name counters...
  n=$1; shift
  rm -rf $out/$tag.$n
  timeout 120 rocprofv3 --output-format csv --pmc "$@" -d $out/$tag.$n -o p -- $R/tools/bin/gemm_lab one 3 "$CFGS" $LAY $M $N $K $EPI > $out/$tag.$n.log 2>&1
}
CFGS=$1; LAY=$2; M=$3; N=$4; K=$5; EPI=${6:-0}
run a SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_ACTIVE_INST_LDS SQ_LDS_IDX_ACTIVE SQ_LDS_BANK_CONFLICT SQ_BUSY_CU_CYCLES
run b SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_LDS
run c SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_MISC SQ_INSTS_VMEM_RD SQ_INSTS_SALU SQ_INSTS_MFMA SQ_INST_CYCLES_VMEM SQ_INSTS_SMEM
run d TCP_PENDING_STALL_CYCLES_sum TCP_TCC_READ_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum TA_BUSY_avr
run e TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum TCC_EA0_RDREQ_sum
python3 - <<PY
import csv, glob, collections, re
acc = collections.OrderedDict()
for f in sorted(glob.glob("$out/$tag.*/**/*counter_collection.csv", recursive=True)):
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"]
        if "gemm" not in k: continue
        m = re.search(r"RingCfg<([^>]*)>", k)
        name = ("ring " + m.group(1)) if m else re.sub(r"\(anonymous namespace\)::", "", k)[:60]
        a = acc.setdefault(name, collections.defaultdict(lambda: [0.0, 0]))
        a[r["Counter_Name"]][0] += float(r["Counter_Value"]); a[r["Counter_Name"]][1] += 1
for k, a in acc.items():
    print(k)
    for c in sorted(a): print("    %-34s %16.0f per launch (%d)" % (c, a[c][0] / a[c][1], a[c][1]))
PY
