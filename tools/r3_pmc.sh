#!/bin/bash
# rocprofv3 kernel stats + PMC passes (SQ, FETCH_SIZE, WRITE_SIZE: separate passes) of the default bench command and of the BLS12-381
# column workloads; summaries + traffic.json land in gpurun_out/r3pmc/ (copied to profiles/ by hand)
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r3pmc
mkdir -p $O
cd $R
bash tools/pmc_run.sh r3pmc/halo2 "--steps 3 --warmup 1 --no-cpu-baseline" > $O/halo2_run.log 2>&1
python tools/pmc_summary.py $O/halo2 $O/r03_f_halo2_pmc_summary.txt halo2_2p20 "rocprofv3 passes of: python bench.py --steps 3 --warmup 1 --no-cpu-baseline (round 3 final build); per-launch averages" > /dev/null
cp $(find $O/halo2/stats -name "*kernel_stats.csv" | head -1) $O/r03_f_halo2_kernel_stats.csv
for c in Bls381G1 Bls381G2; do
  bash tools/pmc_run.sh r3pmc/col_$c "--workload column --curve $c --steps 6 --warmup 1 --no-cpu-baseline --serial" > $O/col_${c}_run.log 2>&1
  python tools/pmc_summary.py $O/col_$c $O/r03_f_column_$(echo $c | tr 'A-Z' 'a-z')_pmc_summary.txt column_${c}_2p20 "rocprofv3 passes of: python bench.py --workload column --curve $c --serial (round 3 final build); per-launch averages" > /dev/null
  cp $(find $O/col_$c/stats -name "*kernel_stats.csv" | head -1) $O/r03_f_column_$(echo $c | tr 'A-Z' 'a-z')_kernel_stats.csv
done
# the raw counter files are large: keep the summaries only
rm -rf $O/halo2 $O/col_Bls381G1 $O/col_Bls381G2
ls -la $O; cat $O/traffic.json | head -30
