#!/bin/bash
# round 3, visit c: GPU suite; A/B of the staged-entry layout (ab/*.so, two rounds); A/B of the saved dcolor/dpw (env knob)
cd $GRAFT_REPO_ROOT; O=$GRAFT_REPO_ROOT/gpurun_out/r3c; mkdir -p $O; export PYTHONDONTWRITEBYTECODE=1
( timeout 1500 python -m pytest tests -m gpu -x -q --durations=5 > $O/pytest.log 2>&1; echo "pytest rc=$?" ); tail -8 $O/pytest.log
bash tools/ab_bench.sh 2 --steps 30 --no-ops 2>&1 | tail -8
for v in 1 0 1 0; do
  EGS_SAVE_DCOLOR=$v timeout 200 python bench.py --cpu-sample 0 --steps 30 2>/dev/null | tail -1 > /tmp/x.json
  python - $v <<'PY'
import json, sys
d = json.load(open("/tmp/x.json"))
k = {n: round(v["avg_us"], 1) for n, v in d["kernels"].items() if "draw" in n or "preprocess" in n}
print("SAVE_DCOLOR", sys.argv[1], "ms/step %.4f" % d["ms_per_step"], "fwd %.4f" % d["fwd_only"]["ms"], "ops %.4f" % d["ops_ms_per_step"], k, flush=True)
PY
done | tee $O/dcolor_ab.txt
cp gpurun_out/ab_bench.txt $O/ 2>/dev/null
