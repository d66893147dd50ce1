"""Busy intervals of every queue over one steady-state step of a bench.py kernel trace (kernels closer than 30 us merged).
usage: queue_timeline.py <trace.csv> <step index>"""
import csv, sys, re, collections
f, step = sys.argv[1], int(sys.argv[2])
rows = list(csv.DictReader(open(f)))
adam = sorted(int(r["Start_Timestamp"]) for r in rows if "adam_kernel" in r["Kernel_Name"])
w0, w1 = adam[step], adam[step + 1]
def short(n):
    m = re.search(r"(\w+_kernel\w*|__amd_rocclr_\w+)", n)
    return ("torch" if "at::native" in n else (m.group(1) if m else n[:30]))[:28]
byq = collections.defaultdict(list)
for r in rows:
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    if w0 - 3_000_000 <= s < w1:
        byq[r["Queue_Id"]].append((s, e, short(r["Kernel_Name"])))
print("step window: 0 .. %.1f us" % ((w1 - w0) / 1e3))
for q in sorted(byq):
    ks = sorted(byq[q])
    segs = []
    for s, e, n in ks:
        if segs and s - segs[-1][1] < 30_000:
            segs[-1][1] = max(segs[-1][1], e); segs[-1][2] += 1; segs[-1][3][n] += 1
        else:
            segs.append([s, e, 1, collections.Counter({n: 1})])
    print("queue %s: %d kernels" % (q, len(ks)))
    for s, e, c, names in segs:
        if e - s > 40_000 or c > 3:
            print("   %9.1f .. %9.1f us  (%4d kernels)  %s" % ((s - w0) / 1e3, (e - w0) / 1e3, c, ", ".join("%s x%d" % kv for kv in names.most_common(3))))
