#!/bin/bash
# one GPU call of round 3: selected tests, the default bench, a kernel-stats profile of the same command
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/${1:-r3}
K="${2:-eval_polynomial or halo2_ipa or kate or halo2_domain or halo2_expression or shim_contract or bench_work}"
mkdir -p $O
cd $R
timeout -k 10 500 python -m pytest tests -q -m gpu -x -k "$K" > $O/pytest.txt 2>&1; echo rc=$? >> $O/pytest.txt; tail -4 $O/pytest.txt
timeout -k 10 250 python bench.py --steps 5 --warmup 2 > $O/bench.json 2> $O/bench.err; echo bench_rc=$?; tail -c 1500 $O/bench.err
python -c "
import json;l=json.load(open('$O/bench.json'));print(l['ms_per_step'],l['phases_ms'],l.get('msm_mops'));print(l['cpu_baseline']['value'], l['cpu_baseline']['sample'])"
timeout -k 10 120 python bench.py --steps 5 --warmup 2 --no-cpu-baseline --expr-limbs 32 2>/dev/null | python -c "
import json,sys;l=json.loads(sys.stdin.read().strip().splitlines()[-1]);print('expr32', l['ms_per_step'],l['phases_ms'])"
cd /tmp && export TMPDIR=/tmp
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats -- python3 $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline > $O/stats.log 2>&1; echo prof_rc=$?
cd $R; f=$(find $O/stats -name "*kernel_stats.csv" | head -1); head -45 $f | cut -c1-170; rm -f $(find $O/stats -name "*kernel_trace.csv")
