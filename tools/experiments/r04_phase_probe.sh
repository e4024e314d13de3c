#!/bin/bash
NT=$1; shift
D=$(mktemp -d /tmp/pp.XXXXXX); cd $D
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -c -fPIC -w --cuda-device-only -DUVS_NT=$NT -DUVS_ALLOW_EXPERIMENTAL_NT "$@" -Rpass-analysis=kernel-resource-usage /root/repo/uv-slam_amd/csrc/uvs_solver.hip -o $D/probe.o 2> $D/remarks.txt || { grep -m3 "error" $D/remarks.txt | cut -c1-200; rm -rf $D; exit 1; }
grep -A12 "Function Name: _ZN6uvsdev7k_solve" $D/remarks.txt | grep -E " VGPRs:| AGPRs| ScratchSize|VGPRs Spill" | sed 's/.*remark: *//; s/ \[-Rpass.*//; s/.*:[0-9]*:[0-9]*: *//' | tr '\n' ';'; echo
rm -rf $D
