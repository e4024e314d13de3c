#!/bin/bash
# The randomized soaks of a round, one record: bash profiles/soak.sh r03  ->  gpurun_out/<tag>_soak.txt (copy it to profiles/).  Run on the GPU box.
tag=${1:-rXX}
mkdir -p gpurun_out
(echo "== gpu_soak.py (persistent kernel)"; python tests/gpu_soak.py 300 31000 persistent 2>&1 | tail -2
 echo "== gpu_soak.py fused"; python tests/gpu_soak.py 300 31000 fused 2>&1 | tail -2
 echo "== gpu_soak_relo.py"; python tests/gpu_soak_relo.py 200 32000 2>&1 | tail -2
 echo "== gpu_soak_relo.py with ESTIMATE_TD"; python tests/gpu_soak_relo.py 150 33000 td 2>&1 | tail -2
 echo "== gpu_soak_relo.py with ESTIMATE_EXTRINSIC"; python tests/gpu_soak_relo.py 150 34000 ex 2>&1 | tail -2
 echo "== gpu_soak_relo.py with ESTIMATE_TD + ESTIMATE_EXTRINSIC"; python tests/gpu_soak_relo.py 150 35000 tdex 2>&1 | tail -2
 echo "== gpu_soak_relo.py with ESTIMATE_EXTRINSIC, fused multi-workgroup loop (relo_Pose as a second-level block of k_large_solve)"; python tests/gpu_soak_relo.py 150 36000 ex fused 2>&1 | tail -2
 echo "== gpu_soak_relo.py with ESTIMATE_TD + ESTIMATE_EXTRINSIC, fused multi-workgroup loop"; python tests/gpu_soak_relo.py 150 37000 tdex fused 2>&1 | tail -2
 echo "== gpu_soak_options.py"; python tests/gpu_soak_options.py 2>&1 | tail -5
 echo "== gpu_soak_more.py"; python tests/gpu_soak_more.py 2>&1 | tail -4
 echo "== gpu_soak_marg_batch.py (uvs_marginalize_batch vs the one-window call)"; python tests/gpu_soak_marg_batch.py 192 41000 2>&1 | tail -4
 echo "== gpu_soak_replay.py"; python tests/gpu_soak_replay.py 2>&1 | tail -5
 echo "== gpu_soak_batch256.py"; python tests/gpu_soak_batch256.py 0 2>&1 | tail -3
 echo "== gpu_soak_rejections.py"; python tests/gpu_soak_rejections.py 2>&1 | tail -70
 echo "== cpu_soak_oracle_variants.py (no HIP code: two CPU builds of the oracle on the prior-free stress windows)"; python tests/cpu_soak_oracle_variants.py 2>&1 | tail -20 | cut -c1-420) > gpurun_out/${tag}_soak.txt 2>&1
