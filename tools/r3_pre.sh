#!/bin/bash
# one bucket set over the table of window multiples (zk_bases_precompute) with windows wider than 16 bits: parity, then A/B bench lines
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r3pre
mkdir -p $O
cd $R
timeout -k 10 600 python -m pytest tests/test_gpu_parity.py -x -q -k "precomputed" > $O/pytest.txt 2>&1 || { tail -30 $O/pytest.txt; exit 1; }
tail -3 $O/pytest.txt
for wb in 0 16 18 19 20; do
  python bench.py --workload column --no-cpu-baseline --precomputed --window-bits $wb 2>$O/err_$wb.txt | tail -1 > $O/column_pre_wb$wb.json
done
python bench.py --workload column --no-cpu-baseline 2>/dev/null | tail -1 > $O/column_plain.json
python bench.py --workload column --no-cpu-baseline --curve Bls381G1 --precomputed 2>/dev/null | tail -1 > $O/column_bls_pre.json
python bench.py --workload column --no-cpu-baseline --curve Bls381G1 2>/dev/null | tail -1 > $O/column_bls_plain.json
python bench.py --no-cpu-baseline --precomputed 2>$O/err_halo2.txt | tail -1 > $O/halo2_pre.json
python bench.py --no-cpu-baseline 2>/dev/null | tail -1 > $O/halo2_plain.json
python - <<PY
import json,glob,os
for f in sorted(glob.glob("$O/*.json")):
    try:
        l=json.loads(open(f).read().strip().splitlines()[-1])
        print(os.path.basename(f), "ms/step %.3f"%l["ms_per_step"], l.get("phases_ms") or l.get("msm_phases_ms") or "", "wb", l["config"].get("msm_window_bits"), "win", l["config"].get("msm_windows_done"), "mops", l.get("msm_mops"))
    except Exception as e: print(f, "FAILED", e)
PY
