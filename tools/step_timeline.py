"""One steady step of a bench.py kernel trace as a timeline of the busiest (main) queue: every kernel with its start offset,
duration and the idle time in front of it; for idle stretches >= MIN_GAP us, what the other queues were running meanwhile.
usage: step_timeline.py <trace.csv> <step index> [min_gap_us=5]"""
import csv, sys, collections, re
f, step = sys.argv[1], int(sys.argv[2])
min_gap = float(sys.argv[3]) if len(sys.argv) > 3 else 5.0
rows = list(csv.DictReader(open(f)))
adam = sorted(int(r["Start_Timestamp"]) for r in rows if "adam_kernel" in r["Kernel_Name"])
w0, w1 = adam[step], adam[step + 1]
def short(n):
    m = re.search(r"(\w+_kernel\w*(<[^>]*>)?|__amd_rocclr_\w+)", n)
    k = m.group(1) if m else n[:40]
    if "at::native" in n:
        k = "torch:" + (re.search(r"(\w+Functor|\w+_kernel_cuda|copy_kernel|\w+_kernel)", n) or [None, "?"])[1][:24]
    return k[:44]
byq = collections.defaultdict(list)
for r in rows:
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    if w0 <= s < w1:
        byq[r["Queue_Id"]].append((s, e, short(r["Kernel_Name"])))
main = max(byq, key=lambda k: sum(e - s for s, e, _ in byq[k]))
ks = sorted(byq[main])
others = sorted((s, e, n, q) for q in byq if q != main for s, e, n in byq[q])
print("step window %.3f ms, main queue %s: %d kernels" % ((w1 - w0) / 1e6, main, len(ks)))
prev_end = w0
for s, e, n in ks:
    gap = (s - prev_end) / 1e3
    line = "%9.1f us  +%7.1f  %-44s" % ((s - w0) / 1e3, (e - s) / 1e3, n)
    if gap >= min_gap:
        act = collections.Counter()
        for os_, oe, on, oq in others:
            ov = min(oe, s) - max(os_, prev_end)
            if ov > 0:
                act["q%s %s" % (oq, on)] += ov / 1e3
        print("      idle %7.1f us   meanwhile: %s" % (gap, "; ".join("%s %.0f us" % kv for kv in act.most_common(4)) or "-"))
    print(line)
    prev_end = max(prev_end, e)
