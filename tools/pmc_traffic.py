"""HBM bytes per launch of each kernel family from two rocprofv3 --pmc passes of bench.py (FETCH_SIZE, WRITE_SIZE).
usage: pmc_traffic.py <dir with the FETCH_SIZE pass> <dir with the WRITE_SIZE pass> > profiles/roundN_hbm_traffic.json
bytes = 2 x FETCH_SIZE KiB (gfx950: FETCH_SIZE tallies 128-B requests at 64 B, MI355X_MICROARCH.md HBM section) + WRITE_SIZE KiB"""
import csv, glob, json, re, sys, collections

FAM = [("attn_bwd_dkv", r"attn_(bwd|bi)_dkv_kernel"), ("attn_bwd_dq", r"attn_(bwd|bi)_dq_kernel"), ("attn_fwd", r"attn_(bi_)?fwd_kernel"),
       ("attn_dense_bias", r"attn_dense_bias_kernel"), ("attn_dbias_grads", r"attn_dbias_grads_kernel"), ("attn_dbias_tables", r"attn_dbias_tables_kernel"),
       ("attn_bwd_reduce", r"attn_bwd_reduce_kernel"), ("gemm_nn_gln", r"gemm_nn_gln_kernel"),
       ("gemm_nt", r"gemm_kernel<0, false"), ("gemm_nn", r"gemm_kernel<0, true"), ("gemm_tn", r"gemm_kernel<1, true"), ("gemm_tn_group", r"gemm_tn_group_kernel"),
       ("conv", r"gemm_kernel<2, "), ("ln_fwd", r"ln_fwd_kernel"), ("ln_bwd", r"ln_bwd(_drop|_lean)?_kernel"), ("adam", r"adam_kernel")]

def collect(d, counter):
    acc = collections.defaultdict(lambda: [0.0, 0])
    for f in glob.glob(d + "/**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            if r["Counter_Name"] != counter:
                continue
            for fam, pat in FAM:
                if re.search(pat, r["Kernel_Name"]):
                    a = acc[fam]; a[0] += float(r["Counter_Value"]); a[1] += 1
                    break
    return acc

rd, wr = collect(sys.argv[1], "FETCH_SIZE"), collect(sys.argv[2], "WRITE_SIZE")
out = {"_method": "rocprofv3 --pmc FETCH_SIZE and, in a separate run, --pmc WRITE_SIZE of `bench.py --steps 2 --warmup 1 "
                  "--no-cpu-baseline` (tools/pmc_traffic.sh, MI355X); per-launch averages over all launches of "
                  "the family; bytes = 2 x FETCH_SIZE KiB (gfx950 correction of MI355X_MICROARCH.md, HBM section: FETCH_SIZE "
                  "tallies 128-B requests at 64 B) + WRITE_SIZE KiB; the adam entry is the calibration (30 B/param algorithmic)"}
for fam, _ in FAM:
    if rd[fam][1] and wr[fam][1]:
        r = 2.0 * rd[fam][0] / rd[fam][1] * 1024.0
        w = wr[fam][0] / wr[fam][1] * 1024.0
        out[fam] = {"bytes_per_launch": int(r + w), "read_bytes_per_launch": int(r), "write_bytes_per_launch": int(w),
                    "launches_sampled": rd[fam][1]}
print(json.dumps(out, indent=1))
