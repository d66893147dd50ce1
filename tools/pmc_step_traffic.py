"""HBM bytes per STEP by kernel from the two --pmc passes of tools/pmc_traffic.sh (FETCH_SIZE x2 on gfx950 + WRITE_SIZE).
The number of executed steps is COUNTED (one adam_kernel launch per step: the timed steps plus bench.py's warm-up, event-timed and
host-timing passes), so launches / step here equal the kernel trace's (round 5's file divided by the --steps argument: 2 x).
usage: pmc_step_traffic.py <rd dir> <wr dir>"""
import csv, glob, collections, re, sys

def short(k):
    k = k.replace("void ", "").replace("(anonymous namespace)::", "").replace("at::native::", "torch:")
    m = re.match(r"([\w:]+(<[^(]*>)?)", k)
    return (m.group(1) if m else k)[:70]

def tot(d, c):
    acc = collections.defaultdict(float); n = collections.defaultdict(int)
    for f in glob.glob(d + "/**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            if r["Counter_Name"] == c:
                k = short(r["Kernel_Name"]); acc[k] += float(r["Counter_Value"]); n[k] += 1
    return acc, n

rd, nr = tot(sys.argv[1], "FETCH_SIZE"); wr, nw = tot(sys.argv[2], "WRITE_SIZE")
steps = float(max(1, sum(v for k, v in nr.items() if "adam_kernel" in k)))
print("executed steps in the run (adam_kernel launches): %d" % steps)
rows = []
for k in set(rd) | set(wr):
    r = 2 * rd.get(k, 0) * 1024 / steps; w = wr.get(k, 0) * 1024 / steps
    rows.append((r + w, r, w, nr.get(k, 0) / steps, k))
rows.sort(reverse=True)
print("total per step: read %.2f GB  write %.2f GB" % (sum(r[1] for r in rows) / 1e9, sum(r[2] for r in rows) / 1e9))
for t, r, w, n, k in rows[:40]:
    print("%8.1f MB/step (rd %8.1f wr %7.1f) %6.1f launches/step  %s" % (t / 1e6, r / 1e6, w / 1e6, n, k))
