"""Markdown summary of a `rocprofv3 --kernel-trace --stats --output-format csv` run of bench.py.

usage: prof_summary.py <dir with *_kernel_stats.csv and *_kernel_trace.csv> <steps profiled> [title]"""
import csv, glob, sys, collections

d, steps = sys.argv[1], int(sys.argv[2])
title = sys.argv[3] if len(sys.argv) > 3 else "rocprofv3 --kernel-trace --stats"
stats = glob.glob(d + "/**/*kernel_stats.csv", recursive=True)[0]
rows = list(csv.DictReader(open(stats)))
tot = sum(float(r["TotalDurationNs"]) for r in rows)
print("# %s\n" % title)
print("total kernel time per step (all streams, summed over concurrent queues): %.2f ms\n" % (tot / steps / 1e6))
trace = glob.glob(d + "/**/*kernel_trace.csv", recursive=True)
if trace:
    q = collections.defaultdict(lambda: [0, 0.0])
    t0, t1 = None, None
    for r in csv.DictReader(open(trace[0])):
        b, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
        k = r.get("Queue_Id", "?")
        q[k][0] += 1
        q[k][1] += (e - b)
        t0 = b if t0 is None else min(t0, b)
        t1 = e if t1 is None else max(t1, e)
    print("queues over the whole run (%.1f ms span, %d steps): " % ((t1 - t0) / 1e6, steps)
          + "; ".join("queue %s: %d kernels / %.1f ms busy per step" % (k, v[0] // steps, v[1] / steps / 1e6)
                      for k, v in sorted(q.items(), key=lambda kv: -kv[1][1])) + "\n")
print("| kernel | calls/step | ms/step | avg us | % |\n|---|---|---|---|---|")
for r in sorted(rows, key=lambda r: -float(r["TotalDurationNs"]))[:40]:
    n = int(r["Calls"])
    t = float(r["TotalDurationNs"])
    print("| `%s` | %d | %.3f | %.1f | %.1f |" % (r["Name"][:96], round(n / steps), t / steps / 1e6, t / n / 1e3, 100 * t / tot))
