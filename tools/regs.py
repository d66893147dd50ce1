"""Register / spill / occupancy summary per kernel of one source: python tools/regs.py attention.hip "-DFOO=1" [filter]"""
import os, re, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from ifseg_amd import build as B
src, flags = sys.argv[1], (sys.argv[2].split() if len(sys.argv) > 2 else [])
flt = sys.argv[3] if len(sys.argv) > 3 else ""
r = subprocess.run(["/opt/rocm/bin/hipcc"] + B.FLAGS + flags + ["-c", os.path.join(B.CSRC, src), "-o", "/dev/null",
                    "-Rpass-analysis=kernel-resource-usage"], capture_output=True, text=True)
cur, row = None, {}
def flush():
    if cur and flt in cur:
        print("%-70s VGPR %3s spill %3s sgpr-spill %3s scratch %4s occ %s" % (cur[:70], row.get("VGPRs"), row.get("VGPRs Spill"),
              row.get("SGPRs Spill"), row.get("ScratchSize [bytes/lane]"), row.get("Occupancy [waves/SIMD]")))
for l in r.stderr.splitlines():
    m = re.search(r"Function Name: (\S+)", l)
    if m: flush(); cur, row = m.group(1), {}
    m = re.search(r"remark:\s+([A-Za-z \[\]/]+): (\d+)", l)
    if m: row[m.group(1).strip()] = m.group(2)
flush()
if r.returncode: print(r.stderr[-2000:])
