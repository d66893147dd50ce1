"""Idle time on the busiest queue of a bench.py kernel trace, attributed to the (previous kernel -> next kernel) pair.
usage: queue_gaps.py <trace.csv> <first step> <last step>"""
import csv, sys, collections, re
f, first, last = sys.argv[1], int(sys.argv[2]), int(sys.argv[3])
rows = list(csv.DictReader(open(f)))
adam = sorted(int(r["Start_Timestamp"]) for r in rows if "adam_kernel" in r["Kernel_Name"])
w0, w1 = adam[first], adam[last]
steps = last - first
def short(n):
    m = re.search(r"(\w+_kernel\w*(<[^>]*>)?|__amd_rocclr_\w+)", n)
    k = m.group(1) if m else n[:40]
    if "at::native" in n:
        k = "torch"
    return k[:34]
byq = collections.defaultdict(list)
for r in rows:
    s = int(r["Start_Timestamp"])
    if w0 <= s < w1:
        byq[r["Queue_Id"]].append((s, int(r["End_Timestamp"]), short(r["Kernel_Name"])))
q = max(byq, key=lambda k: sum(e - s for s, e, _ in byq[k]))
ks = sorted(byq[q])
gaps = collections.defaultdict(lambda: [0, 0.0])
tot = 0.0
hist = collections.Counter()
for (s0, e0, n0), (s1, e1, n1) in zip(ks, ks[1:]):
    g = max(0, s1 - e0)
    tot += g
    a = gaps[(n0, n1)]
    a[0] += 1; a[1] += g
    hist[min(int(g / 2000), 15)] += 1
busy = sum(e - s for s, e, _ in ks)
print("queue %s: %d kernels/step, busy %.2f ms/step, idle between kernels %.2f ms/step, window %.2f ms/step"
      % (q, len(ks) // steps, busy / steps / 1e6, tot / steps / 1e6, (w1 - w0) / steps / 1e6))
print("gap histogram (2 us bins, count/step):", " ".join("%d:%.0f" % (2 * b, c / steps) for b, c in sorted(hist.items())))
for (n0, n1), (c, t) in sorted(gaps.items(), key=lambda kv: -kv[1][1])[:30]:
    print("  %-34s -> %-34s %5.1f /step  avg %6.1f us  total %6.3f ms/step" % (n0, n1, c / steps, t / c / 1e3, t / steps / 1e6))
