#!/bin/bash
# As tools/lab/knob_sweep.sh, for the run-time knobs of the LIBRARY (alternative kernel variants and dispatch maps that
# are compiled into libegs_hip.so): every variant kept in the shipped binary must give a correct path.
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out; export PYTHONDONTWRITEBYTECODE=1
O=gpurun_out/knob_sweep_lib.txt; : > $O
run() {
  echo "=== $*" >> $O
  env "$@" timeout 900 python -m pytest tests -m gpu -q -p no:cacheprovider 2>&1 | grep -E "^FAILED|passed|failed" >> $O
}
run EGS_DRAWB_RED=0 EGS_TILE_MAP=1
run EGS_DRAWB_RED=3 EGS_TILE_MAP=2 EGS_DRAWB_BY_WORK=0
run EGS_TILE_ORDER_F=0 EGS_TILE_ORDER_B=0 EGS_TILE_MAP=0
run EGS_TILE_ORDER_F=2 EGS_TILE_ORDER_B=2 EGS_TILE_SERP=64
run EGS_TILE_ORDER_F=3 EGS_TILE_ORDER_B=3
run EGS_TILE_ORDER_F=4 EGS_TILE_ORDER_B=4 EGS_PRE_STAGE_IN=1
run EGS_DRAW_LDS_PAD=8192 EGS_DRAWB_LDS_PAD=8192 EGS_PRE_LDS_PAD=8192
cat $O
