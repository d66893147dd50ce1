"""Tabulates tools/bin/gemm_lab perf output: one row per shape, one column per configuration (us)."""
import re, sys, collections
rows = collections.OrderedDict()
for l in open(sys.argv[1] if len(sys.argv) > 1 else 'gpurun_out/lab_perf.txt'):
    m = re.match(r'(\w\w) M(\d+)\s+N(\d+)\s+K(\d+)\s+epi\d\s+(.*?)\s+([\d.]+) us\s+([\d.]+) TF/s\s+(\S+)', l)
    if m:
        key = (m.group(1), m.group(2), m.group(3), m.group(4))
        rows.setdefault(key, []).append((m.group(5).strip(), float(m.group(6)), float(m.group(7)), m.group(8)))
cfgs = [c for c, _, _, _ in list(rows.values())[0]]
short = lambda c: c.replace("ring ", "").replace("tile 128x128 (gemm.hip)", "tile").replace(" 2/CU", "*2").replace("x64 ", "/").replace("x32 ", "k32/")
print("%-16s" % "shape" + "".join("%13s" % short(c)[:13] for c in cfgs))
for k, v in rows.items():
    best = min(u for _, u, _, _ in v)
    print("%-16s" % ("%s N%s K%s" % (k[0], k[2], k[3]) if k[1] == '8480' else "%s %s^3" % (k[0], k[1])) +
          "".join("%12.1f%s" % (u, "*" if u == best else " ") for _, u, _, _ in v), "" if all(x[3] != 'MISMATCH' for x in v) else "MISMATCH")
