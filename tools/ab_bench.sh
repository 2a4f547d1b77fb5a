#!/bin/bash
# A/B of library builds on ONE box (box-to-box variance is ~3 %, more than most kernel tweaks):
# every ab/*.so is copied over the in-tree libegs_hip.so in turn, ROUNDS times, and benched.
# Usage (on the GPU box, from the repo root): tools/ab_bench.sh [rounds] [extra bench.py flags...]
rounds=${1:-2}; shift
mkdir -p gpurun_out
cp easygaussiansplatting_amd/libegs_hip.so /tmp/libegs_keep.so
for r in $(seq 1 "$rounds"); do
  for so in ab/*.so; do
    cp "$so" easygaussiansplatting_amd/libegs_hip.so
    timeout 150 python bench.py --cpu-sample 0 "$@" 2>/dev/null | tail -1 > /tmp/ab.json
    python - "$so" "$r" <<'PY'
import json, sys
d = json.load(open("/tmp/ab.json"))
k = {n: round(v["avg_us"], 1) for n, v in d["kernels"].items() if "draw" in n or "preprocess" in n}
print(sys.argv[1], "round", sys.argv[2], "ms/step %.4f" % d["ms_per_step"], "fwd %.4f" % d["fwd_only"]["ms"], k, flush=True)
PY
  done
done | tee gpurun_out/ab_bench.txt
cp /tmp/libegs_keep.so easygaussiansplatting_amd/libegs_hip.so
