"""Per-kernel statistics of the STEADY-STATE steps of a rocprofv3 kernel trace of bench.py: the window between the optimizer
kernels of step <first> and step <last> (0-based over all steps of the run, warm-up included), so that calls/step, ms/step and
the count of foreign (torch / runtime copy) kernels are per executed step and not inflated by packing, warm-up or the bench's
event-timed pass.  usage: steady_stats.py <kernel_trace.csv> <first> <last> [out.csv]   (markdown table on stdout)"""
import csv, sys, collections, re
f, first, last = sys.argv[1], int(sys.argv[2]), int(sys.argv[3])
rows = list(csv.DictReader(open(f)))
adam = sorted(int(r["Start_Timestamp"]) for r in rows if "adam_kernel" in r["Kernel_Name"])
w0, w1 = adam[first], adam[last]
steps = last - first
acc = collections.defaultdict(lambda: [0, 0])
for r in rows:
    if w0 <= int(r["Start_Timestamp"]) < w1:
        a = acc[r["Kernel_Name"]]
        a[0] += 1; a[1] += int(r["End_Timestamp"]) - int(r["Start_Timestamp"])
tot = sum(v[1] for v in acc.values())
own = lambda n: "anonymous namespace" in n
foreign_n = sum(v[0] for n, v in acc.items() if not own(n)); foreign_t = sum(v[1] for n, v in acc.items() if not own(n))
print("# steady-state window: steps %d..%d of the run, %.2f ms per step under the profiler" % (first, last, (w1 - w0) / 1e6 / steps))
print("total kernel time per step (all queues summed): %.2f ms; kernels per step: %.1f; foreign kernels (torch element-wise / "
      "fill / runtime copies) per step: %.1f launches, %.3f ms" % (tot / steps / 1e6, sum(v[0] for v in acc.values()) / steps,
                                                               foreign_n / steps, foreign_t / steps / 1e6))
print("\n| kernel | calls/step | ms/step | avg us | % |\n|---|---|---|---|---|")
for n, v in sorted(acc.items(), key=lambda kv: -kv[1][1])[:45]:
    print("| `%s` | %.1f | %.3f | %.1f | %.1f |" % (n[:100], v[0] / steps, v[1] / steps / 1e6, v[1] / v[0] / 1e3, 100.0 * v[1] / tot))
if len(sys.argv) > 4:
    with open(sys.argv[4], "w") as o:
        w = csv.writer(o)
        w.writerow(["Name", "Calls_per_step", "ms_per_step", "AverageNs", "Percentage"])
        for n, v in sorted(acc.items(), key=lambda kv: -kv[1][1]):
            w.writerow([n, "%.2f" % (v[0] / steps), "%.4f" % (v[1] / steps / 1e6), "%.0f" % (v[1] / v[0]), "%.2f" % (100.0 * v[1] / tot)])
