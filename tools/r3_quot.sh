#!/bin/bash
# the quotient by sub-cosets: parity, then the work-list at 1 / 2 / 4 / 8 parts
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r3quot
mkdir -p $O
cd $R
timeout -k 10 600 python -m pytest tests/test_gpu_parity.py -x -q -k "quotient_by_parts" > $O/pytest.txt 2>&1 || { tail -30 $O/pytest.txt; exit 1; }
tail -3 $O/pytest.txt
for qp in 8 1 4 2; do
  timeout -k 10 300 python bench.py --no-cpu-baseline --quotient-parts $qp 2>$O/err_$qp.txt | tail -1 > $O/halo2_qp$qp.json || { tail -20 $O/err_$qp.txt; exit 1; }
done
timeout -k 10 300 python bench.py --no-cpu-baseline --serial 2>$O/err_serial.txt | tail -1 > $O/halo2_qp8_serial.json
python - <<PY
import json,glob,os
for f in sorted(glob.glob("$O/*.json")):
    try:
        l=json.loads(open(f).read().strip().splitlines()[-1])
        print(os.path.basename(f), "ms/step %.3f"%l["ms_per_step"], {k: round(v,2) for k,v in l["phases_ms"].items()}, "mops", round(l.get("msm_mops") or 0,1))
    except Exception as e: print(f, "FAILED", e)
PY
