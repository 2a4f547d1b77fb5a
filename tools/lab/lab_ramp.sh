#!/bin/bash
# how the timed step depends on what precedes it (idle clocks, the collector): bench.py at several --steps / --warmup / --ramp-steps
cd $GRAFT_REPO_ROOT; O=gpurun_out/ramp; mkdir -p $O; export PYTHONDONTWRITEBYTECODE=1
B="python bench.py --cpu-sample 0 --no-ops"
for spec in "$@"; do
  name="${spec%%:*}"; args="${spec#*:}"
  $B $args > $O/$name.json 2> $O/$name.err
done
python tools/lab/lab_summ.py $O/*.json
