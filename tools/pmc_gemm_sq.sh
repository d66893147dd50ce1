#!/bin/bash
# SQ busy counters of the GEMM kernels in tools/gemm_bench.py (is the kernel LDS- or MFMA-bound?)
out=$GRAFT_REPO_ROOT/gpurun_out/pmc_gemm_sq
rm -rf $out; mkdir -p $out
cd /tmp; export TMPDIR=/tmp
timeout 240 rocprofv3 --output-format csv --pmc SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_ACTIVE_INST_LDS SQ_LDS_IDX_ACTIVE SQ_LDS_BANK_CONFLICT -d $out/p -o p -- python $GRAFT_REPO_ROOT/tools/gemm_bench.py > $out/log.txt 2>&1
timeout 240 rocprofv3 --output-format csv --pmc SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_VALU SQ_INSTS_LDS -d $out/q -o q -- python $GRAFT_REPO_ROOT/tools/gemm_bench.py > $out/log2.txt 2>&1
python - <<PY
import csv, glob, collections
rows = [r for f in glob.glob("$out/*/**/*counter_collection.csv", recursive=True) for r in csv.DictReader(open(f))]
acc = collections.OrderedDict()
for r in rows:
    if "gemm_kernel" not in r["Kernel_Name"]:
        continue
    k = (r["Kernel_Name"][29:66], r["Grid_Size"])
    a = acc.setdefault(k, collections.defaultdict(lambda: [0.0, 0]))
    a[r["Counter_Name"]][0] += float(r["Counter_Value"]); a[r["Counter_Name"]][1] += 1
for k, a in acc.items():
    g = lambda n: a[n][0] / max(1, a[n][1])
    busy = g("SQ_BUSY_CYCLES") / 32.0          # per-SE sum -> cycles of the kernel
    print("%-38s grid %-8s cycles %8.0f  MFMA busy %4.1f%%  LDS inst-active %4.1f%%  LDS idx-active %4.1f%%  bank-conflict cyc %4.1f%%  wait-LDS/wave %4.1f%%  wait-any/wave %4.1f%%"
          % (k[0], k[1], busy, 100 * g("SQ_VALU_MFMA_BUSY_CYCLES") / 1024 / busy, 100 * 4 * g("SQ_ACTIVE_INST_LDS") / 256 / busy,
             100 * g("SQ_LDS_IDX_ACTIVE") / 256 / busy, 100 * g("SQ_LDS_BANK_CONFLICT") / 256 / busy,
             100 * g("SQ_WAIT_INST_LDS") / max(1, g("SQ_WAVE_CYCLES")), 100 * g("SQ_WAIT_INST_ANY") / max(1, g("SQ_WAVE_CYCLES"))))
PY
