#!/bin/bash
# two-stream forward (EGS_FWD_SPLIT): same-box A/B of the bench step + the timeline of the split step
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/split; export PYTHONDONTWRITEBYTECODE=1
python -m pytest tests -m gpu -x -q -k "fused or parity or ring" 2>&1 | grep -E "passed|failed|rror" | tail -2
EGS_FWD_SPLIT=1 python -m pytest tests -m gpu -x -q -k "fused or parity or ring or raw" 2>&1 | grep -E "passed|failed|rror" | tail -2
for r in 1 2 3; do for v in 0 1; do
  EGS_FWD_SPLIT=$v timeout 200 python bench.py --cpu-sample 0 --steps 30 --no-ops --no-ring8 2>/dev/null | tail -1 > /tmp/ab.json
  python - "$v" "$r" <<'PY'
import json, sys
d = json.load(open("/tmp/ab.json"))
k = {n: round(v["avg_us"], 1) for n, v in d["kernels"].items()}
print("split", sys.argv[1], "round", sys.argv[2], "ms/step %.4f" % d["ms_per_step"], "fwd %.4f" % d["fwd_only"]["ms"], k, flush=True)
PY
done; done | tee gpurun_out/split/ab.txt
cd /tmp && export TMPDIR=/tmp
EGS_FWD_SPLIT=1 rocprofv3 --kernel-trace --output-format csv -d /tmp/trs -- python $GRAFT_REPO_ROOT/tools/profile_step.py --steps 160 > /tmp/trs.log 2>&1
python $GRAFT_REPO_ROOT/tools/trace_timeline.py /tmp/trs > $GRAFT_REPO_ROOT/gpurun_out/split/timeline.txt 2>&1; tail -22 $GRAFT_REPO_ROOT/gpurun_out/split/timeline.txt
