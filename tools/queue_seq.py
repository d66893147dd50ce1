"""Kernel sequence of the busiest queue over one steady-state step of a bench.py kernel trace, with the idle gap in
front of every kernel and what the OTHER queues were running during gaps > 20 us.
usage: queue_seq.py <trace.csv> <step index>"""
import csv, sys, re
f, step = sys.argv[1], int(sys.argv[2])
rows = list(csv.DictReader(open(f)))
adam = sorted(int(r["Start_Timestamp"]) for r in rows if "adam_kernel" in r["Kernel_Name"])
w0, w1 = adam[step], adam[step + 1]
def short(n):
    m = re.search(r"(\w+_kernel\w*(<[^>]*>)?|__amd_rocclr_\w+)", n)
    k = m.group(1) if m else n[:40]
    if "at::native" in n:
        mm = re.search(r"at::native::(\w+Functor<\w+|\w+_kernel_cuda|\w+_kernel\w*|\w+)", n)
        k = "torch:" + (mm.group(1) if mm else "?")
    return k[:46]
ks = [(int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Queue_Id"], short(r["Kernel_Name"])) for r in rows
      if w0 <= int(r["Start_Timestamp"]) < w1]
busy = {}
for s, e, q, n in ks:
    busy[q] = busy.get(q, 0) + e - s
main = max(busy, key=busy.get)
mk = sorted(k for k in ks if k[2] == main)
prev_end = w0
for s, e, q, n in mk:
    gap = (s - prev_end) / 1e3
    line = "%9.1f us  +%7.1f gap  %7.1f us  %s" % ((s - w0) / 1e3, gap, (e - s) / 1e3, n)
    if gap > 20:
        others = sorted(set("%s:%s" % (k[2], k[3]) for k in ks if k[2] != main and k[0] < s and k[1] > prev_end))
        line += "    || during gap: " + ", ".join(others[:6])
    print(line)
    prev_end = max(prev_end, e)
