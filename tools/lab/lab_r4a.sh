#!/bin/bash
# round 4, lab A: gradient-error dump under the relative rule, what k_draw_bwd walks, XCD-banded dispatch under ViewStreams
mkdir -p gpurun_out/r4a; O=gpurun_out/r4a
rm -f $O/grad_stats.jsonl
EGS_GRAD_STATS=$O/grad_stats.jsonl EGS_GRAD_STATS_ONLY=1 timeout 900 python -m pytest tests -m gpu -q -x 2>&1 | tail -40 > $O/tests.log
timeout 300 python tools/lab/bwd_hit_stats.py > $O/bwd_hit_stats.txt 2>&1
for rep in 1 2; do
for f in 1 3; do for b in 1 3; do
  EGS_TILE_ORDER_F=$f EGS_TILE_ORDER_B=$b timeout 300 python bench.py --views-per-rank 8 --view-streams 3 --steps 20 --warmup 5 --no-ops --no-ring8 --cpu-sample 0 > $O/v8_f${f}_b${b}_$rep.json 2> $O/v8_f${f}_b${b}_$rep.err
done; done; done
tail -5 $O/tests.log
