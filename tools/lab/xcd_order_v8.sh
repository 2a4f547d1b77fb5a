#!/bin/bash
# forward tile order global (1) vs sorted inside 8 XCD bands (3) with 8 views on 4 streams: step time and
# the forward draw kernel's fetch traffic per view
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/xcd; mkdir -p $O
export PYTHONDONTWRITEBYTECODE=1
for m in 1 3 1 3; do
  EGS_TILE_ORDER_F=$m timeout 120 python $R/bench.py --views-per-rank 8 --view-streams 4 --steps 10 --warmup 3 --ramp-steps 20 --cpu-sample 0 --no-ops --no-skewed 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('order_f=$m', d['ms_per_step'], {k:v['avg_us'] for k,v in d['kernels'].items() if k in ('k_draw','k_draw_bwd')})"
done | tee $O/steps.txt
for m in 1 3; do
  EGS_TILE_ORDER_F=$m timeout 150 rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d /tmp/f$m -- python $R/bench.py --views-per-rank 8 --view-streams 4 --steps 2 --warmup 1 --ramp-steps 2 --cpu-sample 0 --no-ops --no-skewed --no-prof > /tmp/f$m.log 2>&1
  echo "order_f=$m FETCH_SIZE per launch (KB):"; python $R/tools/pmc_summary.py $(dirname $(find /tmp/f$m -name "*counter_collection.csv" | head -1)) --all | grep -A2 "k_draw<" 
done | tee $O/fetch.txt
