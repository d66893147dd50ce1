"""Sum rocprofv3 --pmc counter_collection CSVs per kernel name.  usage: pmc_summary.py <dir> [substr ...]"""
import csv, glob, re, sys, collections
d = sys.argv[1]
subs = sys.argv[2:]
acc = collections.defaultdict(lambda: collections.defaultdict(float))
cnt = collections.Counter()
for f in glob.glob(d + "/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        k = r.get("Kernel_Name", "")
        if subs and not any(s in k for s in subs):
            continue
        m = re.search(r"(\w+_kernel\w*(<[^>]*>)?)", k)
        k = m.group(1) if m else k[:60]
        acc[k][r["Counter_Name"]] += float(r["Counter_Value"])
        cnt[(k, r["Counter_Name"])] += 1
for k in sorted(acc):
    print(k)
    for c in sorted(acc[k]):
        n = cnt[(k, c)]
        print("   %-32s %18.0f  per launch %16.0f  (%d launches)" % (c, acc[k][c], acc[k][c] / n, n))
