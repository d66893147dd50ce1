#!/bin/bash
# FETCH_SIZE / WRITE_SIZE per launch of tools/gemm_bench.py (one kernel instance per shape, in call order)
out=$GRAFT_REPO_ROOT/gpurun_out/pmc_gemm
rm -rf $out; mkdir -p $out
cd /tmp; export TMPDIR=/tmp
# one counter per pass: FETCH_SIZE and WRITE_SIZE together exceed what one pass can collect (rocprofv3 aborts and hangs)
timeout 240 rocprofv3 --output-format csv --pmc FETCH_SIZE -d $out/p -o p -- python $GRAFT_REPO_ROOT/tools/gemm_bench.py > $out/log.txt 2>&1
timeout 240 rocprofv3 --output-format csv --pmc WRITE_SIZE -d $out/q -o q -- python $GRAFT_REPO_ROOT/tools/gemm_bench.py > $out/log2.txt 2>&1
python - <<PY
import csv, glob, collections
rows = [r for f in glob.glob("$out/*/**/*counter_collection.csv", recursive=True) for r in csv.DictReader(open(f))]
acc = collections.OrderedDict()
for r in rows:
    if "gemm_kernel" not in r["Kernel_Name"] and "reduce_parts" not in r["Kernel_Name"]:
        continue
    k = (r["Kernel_Name"][:70], r["Grid_Size"], r.get("LDS_Block_Size", ""))
    a = acc.setdefault(k, collections.defaultdict(lambda: [0.0, 0]))
    a[r["Counter_Name"]][0] += float(r["Counter_Value"]); a[r["Counter_Name"]][1] += 1
for k, a in acc.items():
    f = a["FETCH_SIZE"]; w = a["WRITE_SIZE"]
    print("%-72s grid %-8s n=%3d  read %7.1f MB  write %6.1f MB" % (k[0], k[1], f[1], 2 * f[0] / max(1, f[1]) / 1024, w[0] / max(1, w[1]) / 1024))
PY
