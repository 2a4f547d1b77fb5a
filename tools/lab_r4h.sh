#!/bin/bash
# round 4, lab H: the default bench line with the public-pair seven-op leg
mkdir -p gpurun_out/r4h; O=gpurun_out/r4h
python bench.py --cpu-sample 0 > $O/bench.json 2> $O/bench.err
python - <<'PY'
import json
d = json.loads(open("gpurun_out/r4h/bench.json").read().strip().splitlines()[-1])
print(d["ms_per_step"], d["ops_ms_per_step"], d["ops_public_pair_ms_per_step"], d["ring_views_8"]["ms_per_step"], d["fwd_only"])
PY
