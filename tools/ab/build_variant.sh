#!/bin/bash
# builds a variant of the solver library into ab/lib_<name>.so:  tools/ab/build_variant.sh <name> [-D... flags for BOTH translation units]
# (same flags as __graft_entry__.build; run A/B with tools/ab/run_ab.sh cur ab/lib_<name>.so on the GPU box)
set -e
name=$1; shift
R=$(cd "$(dirname "$0")/../.." && pwd); C=$R/uv-slam_amd/csrc; mkdir -p $R/ab /tmp/abv_$name
common="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -w -mllvm -disable-machine-licm"
/opt/rocm/bin/hipcc $common "$@" -c $C/uvs_solver.hip -o /tmp/abv_$name/a.o &
/opt/rocm/bin/hipcc $common -mllvm -sink-insts-to-avoid-spills "$@" -c $C/uvs_solve512.hip -o /tmp/abv_$name/b.o &
wait
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC /tmp/abv_$name/a.o /tmp/abv_$name/b.o -o $R/ab/lib_$name.so -ldl -pthread
echo built ab/lib_$name.so
