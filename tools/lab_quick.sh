#!/bin/bash
# quick GPU visit: GPU suite (optionally a -k filter in $1) + the default bench line (no CPU leg)
cd $GRAFT_REPO_ROOT; O=$GRAFT_REPO_ROOT/gpurun_out/${OUT:-quick}; mkdir -p $O; export PYTHONDONTWRITEBYTECODE=1
if [ -z "$SKIP_TESTS" ]; then
( timeout 1500 python -m pytest tests -m gpu -x -q --durations=5 ${1:+-k "$1"} > $O/pytest.log 2>&1; echo "pytest rc=$?" ); tail -12 $O/pytest.log
fi
timeout 600 python bench.py --cpu-sample 0 $BENCH_ARGS > $O/bench.json 2> $O/bench.err; tail -2 $O/bench.err
python - <<PY
import json
d=json.loads(open("$O/bench.json").read().strip().splitlines()[-1])
print("ms/step", d["ms_per_step"], "gpu_busy", d["gpu_busy_ms_per_step"], "fwd", d["fwd_only"]["ms"], "ops", d["ops_ms_per_step"])
for k,v in d["kernels"].items(): print("   %-22s x%d %8.1f us"%(k, v["launches_per_step"], v["avg_us"]))
if d.get("ops_kernels"):
    print("ops kernels:")
    for k,v in d["ops_kernels"].items(): print("   %-22s x%d %8.1f us  %s"%(k, v["launches_per_step"], v["avg_us"], v.get("algorithmic_GBs")))
PY
