#!/bin/bash
# after `gpurun -- 'bash profiles/collect.sh rNN'` has merged gpurun_out/ back: bash profiles/copy_back.sh rNN   (the summaries the judge reads live in profiles/, gpurun_out/ is scratch)
TAG=${1:?tag}
cd "$(dirname "$0")/.."
python profiles/summarize.py $TAG
cp gpurun_out/prof_kt/runc_kernel_stats.csv profiles/${TAG}_kernel_stats_bench256.csv
cp gpurun_out/prof_large/runc_kernel_stats.csv profiles/${TAG}_kernel_stats_large.csv
cp gpurun_out/prof_single_fused/runc_kernel_stats.csv profiles/${TAG}_kernel_stats_single_window_fused.csv
for f in phase_cycles phase_cycles_256_threads lin_timeline ab_512_vs_256_threads batch_tail large_timeline icache_l2_counters \
         stream_timeline stream_timeline_d2h_copy stream_ab micro_overlap soak; do
  [ -f gpurun_out/${TAG}_$f.txt ] && cp gpurun_out/${TAG}_$f.txt profiles/
done
ls profiles | grep "^${TAG}_"
