#!/bin/bash
# every bench line quoted in DESIGN.md / README.md, one file per line under gpurun_out/final/
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/final
mkdir -p $O
cd $R
python bench.py > $O/bench_vesta.json 2> $O/bench_vesta.err
python bench.py --curve Pallas --no-cpu-baseline 2>/dev/null | tail -1 > $O/bench_pallas.json
python bench.py --realistic --no-cpu-baseline 2>/dev/null | tail -1 > $O/bench_realistic.json
python bench.py --curve Bn254G1 --no-cpu-baseline 2>/dev/null | tail -1 > $O/bench_bn254.json
python bench.py --curve Bls381G1 --no-cpu-baseline 2>/dev/null | tail -1 > $O/bench_bls381.json
python bench.py --logn 22 --steps 5 --no-cpu-baseline 2>/dev/null | tail -1 > $O/bench_vesta_2e22.json
python bench.py --workload halo2 --steps 5 2>/dev/null | tail -1 > $O/bench_halo2.json
python bench.py --workload groth16 --curve Bn254G1 --logn 22 --steps 3 --warmup 1 2>/dev/null | tail -1 > $O/groth16_bn254_2e22.json
python bench.py --workload groth16 --curve Bls381G1 --logn 22 --steps 3 --warmup 1 2>/dev/null | tail -1 > $O/groth16_bls381_2e22.json
python bench.py --workload groth16 --curve Bls381G1 --logn 20 --steps 3 --warmup 1 2>/dev/null | tail -1 > $O/groth16_bls381_2e20.json
SPLITS=auto python tools/shard_model.py 2>/dev/null | grep ranks > $O/shard_model.txt
ls $O
