"""Per-queue kernel counts / time from a rocprofv3 kernel trace csv of bench.py, restricted to the steady-state window
between the optimizer kernels of step <first> and step <last> (0-based, over all steps of the run incl. warm-up).
usage: queue_kernels.py <trace.csv> <first> <last>"""
import csv, sys, collections, re
f, first, last = sys.argv[1], int(sys.argv[2]), int(sys.argv[3])
rows = list(csv.DictReader(open(f)))
adam = sorted(int(r["Start_Timestamp"]) for r in rows if "adam_kernel" in r["Kernel_Name"])
w0, w1 = adam[first], adam[last]
steps = last - first
print("window: %.2f ms for %d steps = %.2f ms/step (under the profiler)" % ((w1 - w0) / 1e6, steps, (w1 - w0) / 1e6 / steps))
acc = collections.defaultdict(lambda: collections.defaultdict(lambda: [0, 0.0]))
for r in rows:
    if not (w0 <= int(r["Start_Timestamp"]) < w1):
        continue
    n = r["Kernel_Name"]
    m = re.search(r"(\w+_kernel\w*|__amd_rocclr_\w+|at::native::\w+(<[^,>]*)?)", n)
    k = m.group(1) if m else n[:50]
    if "vectorized_elementwise" in n or "elementwise_kernel" in n:
        mm = re.search(r"at::native::(\w+Functor|\w+_kernel_cuda|\w+_kernel)", n)
        k = "torch:" + (mm.group(1) if mm else "elementwise")
    a = acc[r["Queue_Id"]][k]
    a[0] += 1; a[1] += int(r["End_Timestamp"]) - int(r["Start_Timestamp"])
for q in sorted(acc, key=lambda q: -sum(v[1] for v in acc[q].values())):
    tot_n = sum(v[0] for v in acc[q].values()); tot_t = sum(v[1] for v in acc[q].values())
    print("queue %s: %d kernels/step, %.2f ms/step" % (q, tot_n // steps, tot_t / steps / 1e6))
    for k, v in sorted(acc[q].items(), key=lambda kv: -kv[1][0])[:28]:
        print("   %-44s %6.1f /step  %8.3f ms/step" % (k[:44], v[0] / steps, v[1] / steps / 1e6))
