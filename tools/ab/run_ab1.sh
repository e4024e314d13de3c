#!/bin/bash
# one pass A/B (bench once per lib)
for lib in "$@"; do
  export UVS_SOLVER_LIB=$PWD/$lib
  python bench.py --steps 20 --warmup 3 --no-cpu-baseline --no-replay --no-large 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('$lib', 'batch ms %.4f (kernel %.4f) value %.0f  single %.4f ms' % (d['ms_per_step'], d['roofline']['kernel_ms_per_launch'], d['value'], d['single_window_ms']))"
done
