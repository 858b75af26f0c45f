#!/bin/bash
# kernel stats of the column workload for one curve / witness shape: tools/r3_prof_col.sh <out> <curve> [extra bench args]
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/$1
C=$2
shift 2
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
timeout -k 10 200 rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats_$C -- python3 $R/bench.py --workload column --curve $C --serial --no-cpu-baseline --steps 10 --warmup 2 "$@" > $O/stats_$C.log 2>&1
cd $R
f=$(find $O/stats_$C -name "*kernel_stats.csv" | head -1)
python - <<PY
import csv
rows=list(csv.DictReader(open("$f")))
print("== $C $@")
for r in rows[:16]:
    print("%-58s calls %4s avg %9.1f us  %5.2f%%"%(r['Name'].split('(')[0].replace('void zk::','')[:58],r['Calls'],float(r['AverageNs'])/1e3,float(r['Percentage'])))
PY
rm -f $(find $O/stats_$C -name "*kernel_trace.csv")
