"""Build an experimental variant of the library: `python tools/variant.py NAME file.hip "-DFOO=1 -DBAR"` recompiles ONE
source with extra flags and links it with the current objects of the others into ifseg_amd/lib/variants/NAME.so
(select at run time with IFSEG_LIB=...).  A/B kernel experiments in one gpurun call."""
import os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from ifseg_amd import build as B

name, src, flags = sys.argv[1], sys.argv[2], (sys.argv[3].split() if len(sys.argv) > 3 else [])
B.build(verbose=False)
vdir = os.path.join(B.LIBDIR, "variants"); os.makedirs(vdir, exist_ok=True)
odir = os.path.join(B.OBJDIR, "variants"); os.makedirs(odir, exist_ok=True)
o = os.path.join(odir, "%s_%s.o" % (name, src[:-4]))
hipcc = "/opt/rocm/bin/hipcc"
subprocess.run([hipcc] + B.FLAGS + B.PER_FILE_FLAGS.get(src, []) + flags + ["-c", os.path.join(B.CSRC, src), "-o", o], check=True)
objs = [o if s == src else os.path.join(B.OBJDIR, s[:-4] + ".o") for s in B._sources()]
out = os.path.join(vdir, name + ".so")
subprocess.run([hipcc, "--offload-arch=gfx950", "-shared", "-fPIC", "-no-hip-rt", "-o", out] + objs, check=True)
print(out)
