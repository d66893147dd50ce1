"""The ONE gate of the laboratory: measurement / A-B switches of the engine, the trainer and the kernels' host side.

Every such switch is an environment variable `IFSEG_<NAME>` that is honoured only when `IFSEG_LAB=1` is set as well;
without the gate the variable is ignored and the product runs its default path.  `bench.py` refuses to report a
number under `IFSEG_LAB=1` unless it is started with `--lab` (the line is then marked `"lab": true`), exactly as it
refuses `IFSEG_EXP_*` builds.  (csrc/common.h `lab_env` is the same gate for the C++ host code.)"""
import os

def on():
    return os.environ.get("IFSEG_LAB") == "1"


def get(name, default=None):
    """value of the laboratory switch IFSEG_<name> (a string), or `default` outside the laboratory"""
    if not on():
        return default
    return os.environ.get("IFSEG_" + name, default)


def flag(name):
    return get(name) not in (None, "", "0")


def seen():
    """the IFSEG_* variables of this process (bench.py prints them into the line)"""
    return {k: v for k, v in sorted(os.environ.items()) if k.startswith("IFSEG_")}
