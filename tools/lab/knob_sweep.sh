#!/bin/bash
# The whole GPU suite under every A/B knob of the host layer and under forced segment settings: a knob that is kept in
# the tree must give a correct path.  Expected failures: assertions about the DEFAULT path itself (culled lists present,
# bit-equality between a first-sight and a with-history render under forced segments).  -> gpurun_out/knob_sweep.txt
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out; export PYTHONDONTWRITEBYTECODE=1
O=gpurun_out/knob_sweep.txt; : > $O
run() {
  echo "=== $*" >> $O
  env "$@" timeout 900 python -m pytest tests -m gpu -q -p no:cacheprovider 2>&1 | grep -E "^FAILED|passed|failed" >> $O
}
run EGS_SEGMENTS=1 EGS_SEG_SPECULATE=1
run EGS_SEGMENTS=1 EGS_SEG_L=128 EGS_SEG_MIN=128 EGS_SEG_SPECULATE=0
run EGS_ENQUEUE_AHEAD=0
run EGS_MAILBOX_COPY=1
run EGS_TILE_WORK_CACHE=0 EGS_BWD_REUSE_ORDER=0
run EGS_SAVE_DCOLOR=0
run EGS_CULL_LISTS=0
run EGS_TILE_ORDER_REFRESH=2
cat $O
