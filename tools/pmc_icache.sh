#!/bin/bash
# instruction-cache / L2 / scalar-cache counters of k_solve (batch of 256 and one window): gpurun -- 'bash tools/pmc_icache.sh'
set -u
R=$(pwd); export TMPDIR=/tmp
cd /tmp
rocprofv3 -L 2>/dev/null | grep -oE "\b(SQC_[A-Z_0-9]+|SQ_IFETCH[A-Z_0-9]*|SQ_WAIT[A-Z_0-9]*|SQ_INST_LEVEL[A-Z_0-9]*|TCC_(HIT|MISS|REQ|READ)[A-Z_0-9]*sum|TCP_[A-Z_0-9]*sum|SQ_INSTS_[A-Z_0-9]+|SQ_BUSY[A-Z_0-9_]*|SQ_LEVEL_WAVES|SQ_WAVES[A-Z_]*)\b" | sort -u > $R/gpurun_out/counters_avail.txt
for B in 256 1; do
  CMD="python $R/bench.py --steps 6 --warmup 2 --batch $B --no-cpu-baseline --no-replay --no-large"
  i=0
  for G in "SQC_ICACHE_REQ SQC_ICACHE_HITS SQC_ICACHE_MISSES SQC_ICACHE_MISSES_DUPLICATE" "SQ_IFETCH SQ_IFETCH_LEVEL SQ_WAIT_INST_ANY SQ_WAVE_CYCLES" "SQC_DCACHE_REQ SQC_DCACHE_HITS SQC_DCACHE_MISSES SQC_TC_INST_REQ SQC_TC_DATA_READ_REQ" "TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum" "TCP_TCC_READ_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_LATENCY_sum" "SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SMEM SQ_INSTS_FLAT SQ_WAIT_ANY"; do
    i=$((i+1)); rm -rf $R/gpurun_out/pmci_${B}_$i
    rocprofv3 --pmc $G --kernel-trace --output-format csv -d $R/gpurun_out/pmci_${B}_$i -o run -- $CMD > $R/gpurun_out/pmci_${B}_$i.log 2>&1
  done
done
cd $R
python - <<'PY'
import csv, glob, collections
# the bench command launches k_solve both on the batch (grid = 256 x batch threads) and on one resident window (grid = 256): split by grid size
for B in (256, 1):
    for d in sorted(glob.glob(f"gpurun_out/pmci_{B}_*/")):
        f = glob.glob(d + "**/*counter_collection.csv", recursive=True)
        if not f: print(d, "no csv"); continue
        acc = collections.defaultdict(list)
        for r in csv.DictReader(open(f[0])):
            if "k_solve" in r["Kernel_Name"]: acc[(int(r["Grid_Size"]) // 256, r["Counter_Name"])].append(float(r["Counter_Value"]))
        for (wgs, k), v in sorted(acc.items()):
            if B == 256 and wgs != 256: continue      # the single-window launches of this run are covered by the batch-1 run
            print(f"workgroups {wgs:4d}  {k:36s} mean/launch {sum(v)/len(v):16.1f}  launches {len(v)}")
PY
