#!/bin/bash
# usage: bash ab/run_ph.sh lib.so   -> parity tests + bench with UVS_PHASED=1 and without
export UVS_SOLVER_LIB=$PWD/$1
echo "== phased: tests"; UVS_PHASED=1 timeout 600 python -m pytest tests/test_gpu_parity.py tests/test_golden.py tests/test_gpu_edge_cases.py -m gpu -x -q 2>&1 | grep -E "passed|failed|Error|error" | head -5
for rep in 1 2; do
for ph in 1 0; do
  if [ "$ph" = "1" ]; then export UVS_PHASED=1; else unset UVS_PHASED; fi
  python bench.py --steps 20 --warmup 3 --no-cpu-baseline --no-replay --no-large 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('phased=$ph', 'batch ms %.4f (kernel %.4f) value %.0f  single %.4f ms  its %.2f cost %.6f' % (d['ms_per_step'], d['roofline']['kernel_ms_per_launch'], d['value'], d['single_window_ms'], d['lm_iterations_mean'], d['final_cost_mean']))"
done; done
