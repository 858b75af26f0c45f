#!/bin/bash
# the specialised (hiprtc) quotient evaluator: parity (interpreter and specialised kernel), then the work-list with each
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r3jit
mkdir -p $O
cd $R
ZK_EXPR_STATS=1 timeout -k 10 900 python -m pytest tests/test_gpu_parity.py -x -q -k "halo2_expression" > $O/pytest.txt 2>&1 || { tail -40 $O/pytest.txt; exit 1; }
grep -E "expr jit|passed|failed" $O/pytest.txt | sort | uniq -c | tail -8
for w in 2 3 4; do
  ZK_EXPR_JIT_WAVES=$w ZK_EXPR_STATS=1 timeout -k 10 400 python bench.py --no-cpu-baseline --steps 4 2>$O/err_w$w.txt | tail -1 > $O/halo2_jit_w$w.json || { tail -20 $O/err_w$w.txt; exit 1; }
  grep -E "expr jit" $O/err_w$w.txt | sort | uniq -c | head -3
done
python - <<PY
import json,glob,os
for f in sorted(glob.glob("$O/*.json")):
    try:
        l=json.loads(open(f).read().strip().splitlines()[-1])
        print(os.path.basename(f), "ms/step %.3f"%l["ms_per_step"], l["digest"], {k: round(v,2) for k,v in l["phases_ms"].items()})
    except Exception as e: print(f, "FAILED", e)
PY
