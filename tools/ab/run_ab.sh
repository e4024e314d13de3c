#!/bin/bash
# A/B of solver library builds on ONE box: bash tools/ab/run_ab.sh libA.so libB.so ...  (paths relative to the repo root; "cur" = the in-tree build)
for rep in 1 2; do
for lib in "$@"; do
  if [ "$lib" = "cur" ]; then unset UVS_SOLVER_LIB; else export UVS_SOLVER_LIB=$PWD/$lib; fi
  python bench.py --steps 20 --warmup 3 --no-cpu-baseline --no-replay --no-large --no-stream 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('$lib', 'batch ms %.4f (kernel %.4f) value %.0f  single %.4f ms' % (d['ms_per_step'], d['roofline']['kernel_ms_per_launch'], d['value'], d['single_window_ms']))"
done; done
