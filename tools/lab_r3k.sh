#!/bin/bash
cd $GRAFT_REPO_ROOT; O=$GRAFT_REPO_ROOT/gpurun_out/r3k; mkdir -p $O; export PYTHONDONTWRITEBYTECODE=1
( timeout 1500 python -m pytest tests -m gpu -q -x --durations=3 > $O/pytest.log 2>&1; echo "pytest rc=$?" ); tail -5 $O/pytest.log
timeout 300 python bench.py --cpu-sample 0 --steps 30 > $O/bench.json 2> $O/bench.err
python - <<PY
import json
d=json.loads(open("$O/bench.json").read().strip().splitlines()[-1])
print("ms/step", d["ms_per_step"], "gpu_busy", d["gpu_busy_ms_per_step"], "fwd", d["fwd_only"]["ms"], "ops", d["ops_ms_per_step"])
for k,v in d["kernels"].items(): print("   %-22s x%d %8.1f us"%(k, v["launches_per_step"], v["avg_us"]))
PY
OUT=r3k bash tools/lab_timeline.sh | tail -22
