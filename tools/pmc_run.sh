#!/bin/bash
# rocprofv3 passes for the bench command: kernel stats, SQ counters, HBM traffic (separate --pmc passes; see MI355X_MICROARCH.md)
set -o pipefail
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/gpurun_out/${1:-prof}
ARGS="${2:---steps 3 --warmup 1 --no-cpu-baseline --serial}"
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats -- python3 $R/bench.py $ARGS > $OUT/stats.log 2>&1 || echo "stats pass failed"
rocprofv3 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_THREAD_CYCLES_VALU --kernel-trace --output-format csv -d $OUT/pmc_sq -- python3 $R/bench.py $ARGS > $OUT/pmc_sq.log 2>&1 || echo "sq pass failed"
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $OUT/pmc_fetch -- python3 $R/bench.py $ARGS > $OUT/pmc_fetch.log 2>&1 || echo "fetch pass failed"
rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $OUT/pmc_write -- python3 $R/bench.py $ARGS > $OUT/pmc_write.log 2>&1 || echo "write pass failed"
find $OUT -name "*.csv" | wc -l
