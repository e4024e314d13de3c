#!/bin/bash
# HISTORIC (rounds 3 / 4 experiments): builds the WHOLE library with 512 threads into /tmp/b and prints k_solve's resources + spill map.  Only k_solve is
# valid in such a build (the landmark-sharded kernels assume 256 threads); the product builds the 512-thread k_solve as its own translation unit
# (csrc/uvs_solve512.hip, __graft_entry__.build).  Kept because the experiment records in tools/experiments/README.md were made with it.
# usage: tools/build512.sh [extra -D flags...]
set -e
mkdir -p /tmp/b && cd /tmp/b
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -shared -fPIC -w -DUVS_NT=512 -DUVS_ALLOW_EXPERIMENTAL_NT "$@" -gline-tables-only --save-temps \
  -Rpass-analysis=kernel-resource-usage /root/repo/uv-slam_amd/csrc/uvs_solver.hip -o /tmp/b/lib512.so -ldl -pthread 2> /tmp/b/remarks.txt || { tail -30 /tmp/b/remarks.txt; exit 1; }
grep -A12 "Function Name: _ZN6uvsdev7k_solve" /tmp/b/remarks.txt | grep -E "VGPRs:|AGPRs|Scratch|Spill" | sed 's/.*remark: *//; s/ \[-Rpass.*//' | tr '\n' ';'; echo
python3 /root/repo/tools/spill_map.py /tmp/b/uvs_solver-hip-amdgcn-amd-amdhsa-gfx950.s k_solve --lines | head -45
