#!/bin/bash
# the quotient by sub-cosets: same proof elements (digest) from upstream's order, from 8 sub-cosets on one GPU, and from 2 / 4 ranks
# (gloo rehearsal: the ranks share this box's one GPU, so the times of the multi-rank lines mean nothing)
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r3quot2
mkdir -p $O
cd $R
python bench.py --no-cpu-baseline --steps 3 2>$O/err_q1.txt | tail -1 > $O/q1.json
python bench.py --no-cpu-baseline --steps 3 --quotient-parts 8 2>$O/err_q8.txt | tail -1 > $O/q8.json
python bench.py --no-cpu-baseline --steps 3 --quotient-parts 2 2>$O/err_q2.txt | tail -1 > $O/q2.json
for w in 2 4; do
  ZK_BENCH_BACKEND=gloo timeout -k 10 500 python -m torch.distributed.run --nnodes=1 --nproc-per-node $w --master-addr 127.0.0.1 --master-port 2950$w bench.py --gpus $w --steps 1 --warmup 1 --no-cpu-baseline 2>$O/err_w$w.txt | tail -1 > $O/w$w.json || { tail -30 $O/err_w$w.txt; exit 1; }
done
python - <<PY
import json,glob,os
for f in sorted(glob.glob("$O/*.json")):
    try:
        l=json.loads(open(f).read().strip().splitlines()[-1])
        print(os.path.basename(f), "ms/step %.3f"%l["ms_per_step"], l["digest"], {k: round(v,1) for k,v in l["phases_ms"].items()})
    except Exception as e: print(f, "FAILED", e)
PY
