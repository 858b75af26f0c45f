#!/bin/bash
# BLS12-381 (the reference's curve) and cache-footprint measurements of round 3
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/${1:-r3bls}
mkdir -p $O
cd $R
for c in Bls381G1 Bls381G2; do
  for wb in 0 15 14; do
    timeout -k 10 120 python bench.py --workload column --curve $c --serial --no-cpu-baseline --window-bits $wb --realistic 2>/dev/null | tail -1 > $O/column_${c}_wb${wb}_real.json
    timeout -k 10 120 python bench.py --workload column --curve $c --serial --no-cpu-baseline --window-bits $wb 2>/dev/null | tail -1 > $O/column_${c}_wb${wb}.json
  done
done
python - <<PY
import json,glob
for f in sorted(glob.glob("$O/column_*.json")):
    try:
        l=json.load(open(f)); print(f.split("/")[-1], "msm_ms %.3f"%l["msm_ms"], {k:round(v,3) for k,v in l["msm_phases_ms"].items()})
    except Exception as e: print(f, "failed", e)
PY
timeout -k 10 200 python bench.py --workload groth16 --curve Bls381G1 --logn 20 --steps 5 --warmup 2 --no-cpu-baseline 2>$O/g16.err | tail -1 > $O/groth16_bls381_2e20.json; python -c "
import json;l=json.load(open('$O/groth16_bls381_2e20.json'));print('groth16 bls 2^20', l['ms_per_step'], l['phases_ms'])"
timeout -k 10 200 python bench.py --workload groth16 --curve Bn254G1 --logn 22 --steps 3 --warmup 1 --no-cpu-baseline 2>>$O/g16.err | tail -1 > $O/groth16_bn254_2e22.json; python -c "
import json;l=json.load(open('$O/groth16_bn254_2e22.json'));print('groth16 bn254 2^22', l['ms_per_step'], l['phases_ms'])"
timeout -k 10 200 python tools/batch_vs_single.py Vesta 20 > $O/batch_vs_single_vesta.txt 2>&1; cat $O/batch_vs_single_vesta.txt
