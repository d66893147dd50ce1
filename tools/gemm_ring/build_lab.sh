#!/bin/bash
# Builds the stand-alone GEMM laboratory (tools/gemm_lab.cpp) against the in-tree library.
set -e
cd "$(dirname "$0")/.."
python -m ifseg_amd.build >/dev/null
mkdir -p tools/bin
/opt/rocm/bin/hipcc -O2 -std=c++17 tools/gemm_lab.cpp -o tools/bin/gemm_lab -Lifseg_amd/lib -lifseg_hip -Wl,-rpath,'$ORIGIN/../../ifseg_amd/lib'
echo tools/bin/gemm_lab
