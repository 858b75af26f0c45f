#!/bin/bash
# every bench line quoted in DESIGN.md / README.md for round 3, one file per line under gpurun_out/final3/ (copied to profiles/r03_g_*)
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/final3
mkdir -p $O
cd $R
python bench.py > $O/bench_halo2.json 2> $O/bench_halo2.err
python bench.py --serial --no-cpu-baseline 2>/dev/null | tail -1 > $O/bench_halo2_serial.json
python bench.py --expr-limbs 32 --no-cpu-baseline 2>/dev/null | tail -1 > $O/bench_halo2_expr32.json
python bench.py --expr-kernel never --no-cpu-baseline 2>/dev/null | tail -1 > $O/bench_halo2_expr_interpreter.json
python bench.py --ipa virtual --no-cpu-baseline 2>/dev/null | tail -1 > $O/bench_halo2_ipa_virtual.json
python bench.py --ipa fold --no-cpu-baseline 2>/dev/null | tail -1 > $O/bench_halo2_ipa_fold.json
python bench.py --workload column 2>/dev/null | tail -1 > $O/bench_column_vesta.json
python bench.py --workload column --curve Pallas --no-cpu-baseline 2>/dev/null | tail -1 > $O/bench_column_pallas.json
python bench.py --workload column --realistic --no-cpu-baseline 2>/dev/null | tail -1 > $O/bench_column_realistic.json
python bench.py --workload column --curve Bn254G1 --no-cpu-baseline 2>/dev/null | tail -1 > $O/bench_column_bn254.json
python bench.py --workload column --curve Bls381G1 --serial --no-cpu-baseline 2>/dev/null | tail -1 > $O/bench_column_bls381_g1.json
python bench.py --workload column --curve Bls381G2 --serial --no-cpu-baseline 2>/dev/null | tail -1 > $O/bench_column_bls381_g2.json
python bench.py --workload column --curve Bls381G1 --serial --realistic --no-cpu-baseline 2>/dev/null | tail -1 > $O/bench_column_bls381_g1_realistic.json
python bench.py --workload column --curve Bls381G2 --serial --realistic --no-cpu-baseline 2>/dev/null | tail -1 > $O/bench_column_bls381_g2_realistic.json
python bench.py --workload column --curve Bn254G2 --serial --no-cpu-baseline 2>/dev/null | tail -1 > $O/bench_column_bn254_g2.json
python bench.py --workload column --logn 22 --steps 5 --no-cpu-baseline 2>/dev/null | tail -1 > $O/bench_column_vesta_2e22.json
python bench.py --workload groth16 --curve Bls381G1 --logn 20 --steps 4 --warmup 1 2>/dev/null | tail -1 > $O/groth16_bls381_2e20.json
python bench.py --workload groth16 --curve Bls381G1 --logn 22 --steps 3 --warmup 1 --no-cpu-baseline 2>/dev/null | tail -1 > $O/groth16_bls381_2e22.json
python bench.py --workload groth16 --curve Bn254G1 --logn 22 --steps 3 --warmup 1 2>/dev/null | tail -1 > $O/groth16_bn254_2e22.json
SPLITS=auto python tools/shard_model.py Vesta 20 2>/dev/null | grep ranks > $O/shard_model_vesta_2e20.txt
SPLITS=auto python tools/shard_model.py Bn254G1 22 2>/dev/null | grep ranks > $O/shard_model_bn254_2e22.txt
python - <<PY
import json,glob,os
for f in sorted(glob.glob("$O/*.json")):
    try:
        l=json.loads(open(f).read().strip().splitlines()[-1])
        print(os.path.basename(f), "ms/step %.2f"%l["ms_per_step"], "value %.3g"%l["value"], l.get("phases_ms") or l.get("msm_phases_ms") or "", "msm_mops", l.get("msm_mops"))
    except Exception as e: print(f, "FAILED", e)
PY
