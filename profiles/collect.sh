#!/bin/bash
# Collects the rocprofv3 evidence summarised under profiles/ (run on the GPU box through gpurun from the repo root):
#   gpurun --timeout 1500 -- 'bash profiles/collect.sh r01b'
# One pass for the kernel trace, separate passes per PMC group (FETCH_SIZE and WRITE_SIZE do not fit one pass; PMC is never
# combined with sys/hip/hsa traces).  Raw CSVs stay in gpurun_out/ (scratch); profiles/summarize.py writes the tracked summary.
set -u
TAG=${1:-r01}
R=$(pwd)
export TMPDIR=/tmp
CMD="python $R/bench.py --steps 10 --warmup 2 --no-cpu-baseline --no-replay --no-large --no-stream"
LCMD="python $R/bench.py --steps 2 --warmup 1 --batch 4 --no-cpu-baseline --no-replay --no-fused-single --no-stream --no-scale-point"      # the configs[3] window through the fused loop (kernel trace only)
cd /tmp
rm -rf $R/gpurun_out/prof_kt $R/gpurun_out/prof_fetch $R/gpurun_out/prof_write $R/gpurun_out/prof_sq $R/gpurun_out/prof_sq2
rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_kt -o runc -- $CMD > $R/gpurun_out/prof_kt.log 2>&1
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $R/gpurun_out/prof_fetch -o runc -- $CMD > $R/gpurun_out/prof_fetch.log 2>&1
rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $R/gpurun_out/prof_write -o runc -- $CMD > $R/gpurun_out/prof_write.log 2>&1
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_INSTS_VALU_MFMA_MOPS_F64 SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES --kernel-trace --output-format csv -d $R/gpurun_out/prof_sq -o runc -- $CMD > $R/gpurun_out/prof_sq.log 2>&1
rocprofv3 --pmc SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_INSTS_SALU SQ_INSTS_SMEM --kernel-trace --output-format csv -d $R/gpurun_out/prof_sq2 -o runc -- $CMD > $R/gpurun_out/prof_sq2.log 2>&1
rm -rf $R/gpurun_out/prof_large
rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_large -o runc -- $LCMD > $R/gpurun_out/prof_large.log 2>&1
rm -rf $R/gpurun_out/prof_single_fused
rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_single_fused -o runc -- python $R/tools/single_vs_fused.py > $R/gpurun_out/prof_single_fused.log 2>&1      # ONE canonical window through both single-window forms
# PMC passes for the landmark-sharded kernels (configs[3] fused loop): traffic of the loop (FETCH_SIZE / WRITE_SIZE in separate passes) + wave / wait counters
rm -rf $R/gpurun_out/prof_large_fetch $R/gpurun_out/prof_large_write $R/gpurun_out/prof_large_sq
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $R/gpurun_out/prof_large_fetch -o runc -- $LCMD > $R/gpurun_out/prof_large_fetch.log 2>&1
rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $R/gpurun_out/prof_large_write -o runc -- $LCMD > $R/gpurun_out/prof_large_write.log 2>&1
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_BUSY_CYCLES --kernel-trace --output-format csv -d $R/gpurun_out/prof_large_sq -o runc -- $LCMD > $R/gpurun_out/prof_large_sq.log 2>&1
cd $R
# per-phase cycles of one window with the n = 75 prior (debug launch, thread 0), the batch tail (slowest window / contention) and the instruction-cache / L2 counters:
# plain text files of the round, not derived from a trace
python tests/gpu_debug_prior.py > gpurun_out/${TAG}_phase_cycles.txt 2>&1
UVS_KSOLVE_NT=256 python tests/gpu_debug_prior.py > gpurun_out/${TAG}_phase_cycles_256_threads.txt 2>&1      # the one-wave-per-SIMD instantiation of the persistent kernel, same window
# per-wave step log of one LM iteration (second linearization of the canonical window with its prior), both instantiations
UVS_DEBUG_LIN_TIMELINE=$R/gpurun_out/${TAG}_tl512.bin python tests/gpu_debug_prior.py > /dev/null 2>&1; python tools/lin_timeline.py gpurun_out/${TAG}_tl512.bin 2 > gpurun_out/${TAG}_lin_timeline.txt 2>&1
UVS_KSOLVE_NT=256 UVS_DEBUG_LIN_TIMELINE=$R/gpurun_out/${TAG}_tl256.bin python tests/gpu_debug_prior.py > /dev/null 2>&1; echo "---- UVS_KSOLVE_NT=256" >> gpurun_out/${TAG}_lin_timeline.txt; python tools/lin_timeline.py gpurun_out/${TAG}_tl256.bin 2 >> gpurun_out/${TAG}_lin_timeline.txt 2>&1
# same-box A/B of the two instantiations (batch of 256 + one window), twice each
for rep_ in 1 2; do for nt_ in 512 256; do UVS_KSOLVE_NT=$nt_ python bench.py --steps 20 --warmup 3 --no-cpu-baseline --no-replay --no-large --no-stream 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('UVS_KSOLVE_NT=$nt_', 'batch ms %.4f (kernel %.4f) value %.0f  single window %.4f ms' % (d['ms_per_step'], d['roofline']['kernel_ms_per_launch'], d['value'], d['single_window_ms']))"; done; done > gpurun_out/${TAG}_ab_512_vs_256_threads.txt 2>&1
python tools/batch_tail.py > gpurun_out/${TAG}_batch_tail.txt 2>&1
python tools/large_timeline.py config3 > gpurun_out/${TAG}_large_timeline.txt 2>&1; python tools/large_timeline.py canonical >> gpurun_out/${TAG}_large_timeline.txt 2>&1
bash tools/pmc_icache.sh > gpurun_out/${TAG}_icache_l2_counters.txt 2>&1
# ---- the end-to-end stream: host calls / copies / kernels timeline of the default form and of the form before round 5's fix (results fetched by a copy, two sets), the switches' A/B, the overlap probe
# (round 5 also ran the opt-in dense landmark path here; it left the tree in round 6: tools/experiments/r05_dense/)
(cd /tmp; rm -rf $R/gpurun_out/prof_stream; rocprofv3 --kernel-trace --memory-copy-trace --hip-runtime-trace --output-format csv -d $R/gpurun_out/prof_stream -o s -- python $R/tools/stream_rate.py 16 > /dev/null 2>&1)
python tools/stream_trace.py gpurun_out/prof_stream 14 > gpurun_out/${TAG}_stream_timeline.txt 2>&1
(cd /tmp; rm -rf $R/gpurun_out/prof_stream_d2h; UVS_STREAM_D2H_COPY=1 UVS_STREAM_SETS=2 rocprofv3 --kernel-trace --memory-copy-trace --hip-runtime-trace --output-format csv -d $R/gpurun_out/prof_stream_d2h -o s -- python $R/tools/stream_rate.py 16 > /dev/null 2>&1)
python tools/stream_trace.py gpurun_out/prof_stream_d2h 14 > gpurun_out/${TAG}_stream_timeline_d2h_copy.txt 2>&1
# (round 6: the switches are read per call; ONE process alternates the configurations on the same windows, ten runs each: median / quartiles)
python tools/stream_ab.py 10 32 256 > gpurun_out/${TAG}_stream_ab.txt 2>&1
(cd tools && hipcc --offload-arch=gfx950 -O2 -o micro_overlap micro_overlap.hip -lpthread 2>/dev/null; ./micro_overlap 46 1.5) > gpurun_out/${TAG}_micro_overlap.txt 2>&1
python profiles/summarize.py $TAG      # printed for the log; gpurun only merges gpurun_out/ back, so re-run these two lines locally afterwards:
#   bash profiles/copy_back.sh $TAG
