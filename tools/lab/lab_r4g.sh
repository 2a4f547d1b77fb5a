#!/bin/bash
# round 4, lab G: do the lanes of ViewStreams gain from being kept OUT of phase? (stream priorities, a staggered start)
mkdir -p gpurun_out/r4g; O=gpurun_out/r4g
run() { tag=$1; shift; env "$@" timeout 300 python bench.py --views-per-rank 8 --view-streams ${LANES:-3} --steps 20 --warmup 5 --no-ops --no-ring8 --cpu-sample 0 2>/dev/null | tail -1 > /tmp/b.json; python -c "
import json; d=json.load(open('/tmp/b.json')); print('$tag', d['ms_per_step'], d['value'])"; }
for rep in 1 2; do
  run base X=1
  run prio EGS_VIEW_STREAM_PRIO=1
  run stag150 EGS_VIEW_STAGGER_US=150
  run stag300 EGS_VIEW_STAGGER_US=300
  run prio_stag300 EGS_VIEW_STREAM_PRIO=1 EGS_VIEW_STAGGER_US=300
  LANES=2 run lanes2 X=1
  LANES=2 run lanes2_stag400 EGS_VIEW_STAGGER_US=400
  LANES=4 run lanes4 X=1
  LANES=4 run lanes4_stag200 EGS_VIEW_STAGGER_US=200
done | tee $O/view_streams.txt
