#!/bin/bash
# same-box A/B of ab/*.so with the full kernel table
rounds=${1:-2}
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out; export PYTHONDONTWRITEBYTECODE=1
cp easygaussiansplatting_amd/libegs_hip.so /tmp/libegs_keep.so
for r in $(seq 1 "$rounds"); do for so in ab/*.so; do
  cp "$so" easygaussiansplatting_amd/libegs_hip.so
  timeout 200 python bench.py --cpu-sample 0 --steps 30 --no-ops --no-ring8 2>/dev/null | tail -1 > /tmp/ab.json
  python - "$so" "$r" <<'PY'
import json, sys
d = json.load(open("/tmp/ab.json"))
k = {n: round(v["avg_us"], 1) for n, v in d["kernels"].items()}
print(sys.argv[1], "round", sys.argv[2], "ms/step %.4f" % d["ms_per_step"], "busy %.4f" % d["gpu_busy_ms_per_step"], k, flush=True)
PY
done; done | tee gpurun_out/ab_bench2.txt
cp /tmp/libegs_keep.so easygaussiansplatting_amd/libegs_hip.so
