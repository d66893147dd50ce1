"""Disassemble the device code of the built library and look for the one instruction form that went wrong next to a running
GEMM (DESIGN.md "Round 3" (6), tools/probe/README.md): a packed-fp32 multiply / FMA / add whose LOW lane reads the HIGH half
of its second source (`op_sel:[x,1...]`), e.g. `v_pk_mul_f32 v[a:b], v[c:d], v[e:f] op_sel:[0,1] op_sel_hi:[1,0]`.
usage: python tools/check_isa.py [lib.so]      exit code 1 if the form occurs"""
import os, re, shutil, subprocess, sys, tempfile
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
OBJDUMP = "/opt/rocm/lib/llvm/bin/llvm-objdump"
PAT = re.compile(r"v_pk_(mul|fma|add)_f32\b.*\bop_sel:\[[01],1")


def scan(lib):
    """-> (number of gfx950 code objects, list of (kernel, instruction))"""
    hits, nobj = [], 0
    with tempfile.TemporaryDirectory() as td:
        so = os.path.join(td, "lib.so")
        shutil.copy(lib, so)                                 # --offloading extracts next to its input
        subprocess.run([OBJDUMP, "--offloading", so], stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL, check=True)
        for f in sorted(os.listdir(td)):
            if "gfx950" not in f:
                continue
            nobj += 1
            dis = subprocess.run([OBJDUMP, "-d", "--mcpu=gfx950", os.path.join(td, f)], capture_output=True, text=True).stdout
            kernel = "?"
            for line in dis.splitlines():
                m = re.match(r"^[0-9a-f]+ <(.+)>:", line)
                if m:
                    kernel = m.group(1)
                elif PAT.search(line):
                    hits.append((kernel, " ".join(line.split("//")[0].split())))
    return nobj, hits


if __name__ == "__main__":
    lib = sys.argv[1] if len(sys.argv) > 1 else os.path.join(ROOT, "ifseg_amd", "lib", "libifseg_hip.so")
    nobj, hits = scan(lib)
    print("%s: %d gfx950 code objects, %d packed-fp32 instructions whose low lane reads the high half of src1" % (lib, nobj, len(hits)))
    for k, ins in hits[:20]:
        print("  %s: %s" % (k[:60], ins))
    sys.exit(1 if hits or not nobj else 0)
