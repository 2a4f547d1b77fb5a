#!/bin/bash
# round 3, visit d: footprint-culled lists -- whole GPU suite (all failures), then A/B by env knob
cd $GRAFT_REPO_ROOT; O=$GRAFT_REPO_ROOT/gpurun_out/r3d; mkdir -p $O; export PYTHONDONTWRITEBYTECODE=1
( timeout 1500 python -m pytest tests -m gpu -q --durations=5 > $O/pytest.log 2>&1; echo "pytest rc=$?" ); tail -15 $O/pytest.log
for v in 1 0 1 0; do
  EGS_CULL_LISTS=$v timeout 200 python bench.py --cpu-sample 0 --steps 30 --no-ops 2>$O/bench_$v.err | tail -1 > /tmp/x.json
  python - $v <<'PY'
import json, sys
d = json.load(open("/tmp/x.json"))
k = {n: round(v["avg_us"], 1) for n, v in d["kernels"].items()}
print("CULL", sys.argv[1], "ms/step %.4f" % d["ms_per_step"], "busy %.4f" % d["gpu_busy_ms_per_step"], "fwd %.4f" % d["fwd_only"]["ms"], d["config"]["patches"], d["config"]["patches_drawn"], k, flush=True)
PY
done | tee $O/cull_ab.txt
tail -3 $O/bench_1.err
