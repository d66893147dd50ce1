"""Build libifseg_hip.so (gfx950) in-tree with hipcc -- no JIT cache, no torch extension.

    python -m ifseg_amd.build [--force]

The shared library exposes the plain-C ABI declared in include/ifseg_hip.h and is
loaded with ctypes by ifseg_amd.hip (PyTorch is only used for device memory and
streams).  hipcc cross-compiles for gfx950 without a GPU.
"""
import os
import subprocess
import sys
from concurrent.futures import ThreadPoolExecutor

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
LIBDIR = os.path.join(HERE, "lib")
LIB = os.path.join(LIBDIR, "libifseg_hip.so")
OBJDIR = os.path.join(HERE, "build")
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-Wno-unused-value"] + os.environ.get("IFSEG_EXTRA_FLAGS", "").split()


# per-file flags on top of FLAGS (see the build note at the top of csrc/ffn_ln.hip)
PER_FILE_FLAGS = {"ffn_ln.hip": ["-fno-slp-vectorize"]}


def _sources():
    return sorted(f for f in os.listdir(CSRC) if f.endswith(".hip"))


def _stale(target, deps):
    if not os.path.exists(target):
        return True
    t = os.path.getmtime(target)
    return any(os.path.getmtime(d) > t for d in deps)


def _torch_libdir():
    import importlib.util
    spec = importlib.util.find_spec("torch")
    if spec is None or not spec.submodule_search_locations:
        return None
    d = os.path.join(list(spec.submodule_search_locations)[0], "lib")
    return d if os.path.exists(os.path.join(d, "libamdhip64.so")) else None


def build(force=False, verbose=True):
    hipcc = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
    if not os.path.exists(hipcc):
        hipcc = "hipcc"
    os.makedirs(LIBDIR, exist_ok=True)
    os.makedirs(OBJDIR, exist_ok=True)
    headers = [os.path.join(CSRC, f) for f in os.listdir(CSRC) if f.endswith(".h")]
    headers.append(os.path.join(os.path.dirname(HERE), "include", "ifseg_hip.h"))
    jobs = []
    objs = []
    for src in _sources():
        s = os.path.join(CSRC, src)
        o = os.path.join(OBJDIR, src[:-4] + ".o")
        objs.append(o)
        if force or _stale(o, [s] + headers):
            jobs.append([hipcc] + FLAGS + PER_FILE_FLAGS.get(src, []) + ["-c", s, "-o", o])

    def run(cmd):
        if verbose:
            print("[ifseg_amd.build]", " ".join(cmd[-4:]), flush=True)
        r = subprocess.run(cmd, capture_output=True, text=True)
        if r.returncode != 0:
            raise RuntimeError("hipcc failed:\n%s\n%s" % (" ".join(cmd), r.stderr[-4000:]))

    with ThreadPoolExecutor(max_workers=min(8, max(1, len(jobs)))) as ex:
        list(ex.map(run, jobs))
    if force or jobs or _stale(LIB, objs):
        # No DT_NEEDED on a HIP runtime: PyTorch bundles its own copy (torch/lib/libamdhip64.so) next to
        # /opt/rocm's, and two runtimes in one process do not share devices/streams (observed: "no
        # ROCm-capable device is detected" from ours).  ifseg_amd.hip preloads torch's copy RTLD_GLOBAL
        # and the HIP symbols of this library bind to it at dlopen time.
        run([hipcc, "--offload-arch=gfx950", "-shared", "-fPIC", "-no-hip-rt", "-o", LIB] + objs)
    return LIB


if __name__ == "__main__":
    print(build(force="--force" in sys.argv))
