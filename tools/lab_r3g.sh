#!/bin/bash
# GPU suite, then same-box A/B of ab/*.so (ops figure included: k_sh2color reads the same rows)
cd $GRAFT_REPO_ROOT; O=$GRAFT_REPO_ROOT/gpurun_out/r3g; mkdir -p $O; export PYTHONDONTWRITEBYTECODE=1
( timeout 1500 python -m pytest tests -m gpu -q --durations=5 > $O/pytest.log 2>&1; echo "pytest rc=$?" ); tail -6 $O/pytest.log
cp easygaussiansplatting_amd/libegs_hip.so /tmp/libegs_keep.so
for r in 1 2; do for so in ab/*.so; do
  cp "$so" easygaussiansplatting_amd/libegs_hip.so
  timeout 200 python bench.py --cpu-sample 0 --steps 30 2>/dev/null | tail -1 > /tmp/ab.json
  python - "$so" "$r" <<'PY'
import json, sys
d = json.load(open("/tmp/ab.json"))
k = {n: round(v["avg_us"], 1) for n, v in d["kernels"].items() if "draw" in n or "preprocess" in n}
ok = {n: round(v["avg_us"], 1) for n, v in (d.get("ops_kernels") or {}).items() if "sh2" in n or "chain" in n}
print(sys.argv[1], "round", sys.argv[2], "ms/step %.4f" % d["ms_per_step"], "fwd %.4f" % d["fwd_only"]["ms"], "ops %.4f" % d["ops_ms_per_step"], k, ok, flush=True)
PY
done; done | tee $O/ab.txt
cp /tmp/libegs_keep.so easygaussiansplatting_amd/libegs_hip.so
