#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/gpurun_out/pmc2
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rocprofv3 -L > $OUT/counters.txt 2>&1
for set in "SQ_WAVES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_THREAD_CYCLES_VALU SQ_ACTIVE_INST_VALU SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_LDS" "SQ_WAVES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_INST_CYCLES_SALU SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_MISC SQ_IFETCH"; do
  tag=$(echo $set | cut -d' ' -f3)
  rocprofv3 --pmc $set --kernel-trace --output-format csv -d $OUT/bench_$tag -- python3 $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline > $OUT/bench_$tag.log 2>&1 || echo "bench pass $tag failed"
  rocprofv3 --pmc $set --kernel-trace --output-format csv -d $OUT/micro_$tag -- $R/tools/microbench > $OUT/micro_$tag.log 2>&1 || echo "micro pass $tag failed"
done
find $OUT -name "*counter_collection.csv" | head
