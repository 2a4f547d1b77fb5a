#!/bin/bash
# round 2, lab A: parity of the new paths + same-box A/B of the tile dispatch orders and the validation modes
cd $GRAFT_REPO_ROOT
O=gpurun_out/labA; mkdir -p $O
export PYTHONDONTWRITEBYTECODE=1
( timeout 900 python -m pytest tests -m gpu -x -q > $O/pytest_default.log 2>&1; echo "pytest default rc=$?" ) 
tail -3 $O/pytest_default.log
( EGS_TILE_ORDER_F=4 EGS_TILE_ORDER_B=1 timeout 600 python -m pytest tests/test_gpu_parity.py tests/test_gpu_fused_ahead.py -x -q > $O/pytest_order41.log 2>&1; echo "pytest order F4 B1 rc=$?" )
tail -2 $O/pytest_order41.log
( EGS_TILE_ORDER_F=2 EGS_TILE_ORDER_B=3 timeout 600 python -m pytest tests/test_gpu_parity.py -x -q > $O/pytest_order23.log 2>&1; echo "pytest order F2 B3 rc=$?" )
tail -2 $O/pytest_order23.log
B="python bench.py --steps 20 --warmup 5 --cpu-sample 0 --no-ops"
$B > $O/base.json 2> $O/base.err
$B --immediate > $O/immediate.json 2> $O/immediate.err
for f in 1 2 3 4; do EGS_TILE_ORDER_F=$f $B > $O/F$f.json 2>/dev/null; done
for b in 1 2 3 4; do EGS_TILE_ORDER_B=$b $B > $O/B$b.json 2>/dev/null; done
EGS_TILE_ORDER_F=2 EGS_TILE_SERP=512 $B > $O/F2s512.json 2>/dev/null
EGS_TILE_ORDER_F=4 EGS_TILE_SERP=64 $B > $O/F4s64.json 2>/dev/null
$B > $O/base2.json 2> $O/base2.err
python bench.py --steps 50 --warmup 5 > $O/full.json 2> $O/full.err
python tools/lab_summ.py $O/base.json $O/immediate.json $O/F?.json $O/B?.json $O/F2s512.json $O/F4s64.json $O/base2.json $O/full.json
cat $O/base.err | tail -3
