#!/bin/bash
# round 4, lab I: content-validated public pair: tests + the two seven-op figures, memo on / off
mkdir -p gpurun_out/r4i; O=gpurun_out/r4i
timeout 900 python -m pytest tests/test_gpu_memo.py tests/test_gpu_parity.py -m gpu -q 2>&1 | tail -30 > $O/tests.log
for rep in 1 2; do
python bench.py --cpu-sample 0 --no-ring8 2>/dev/null | tail -1 > /tmp/b.json
python - <<'PY'
import json
d = json.load(open("/tmp/b.json"))
print("ms/step", d["ms_per_step"], "ops(handle)", d["ops_ms_per_step"], "public pair", d["ops_public_pair_ms_per_step"], flush=True)
PY
done | tee $O/bench.txt
tail -4 $O/tests.log
