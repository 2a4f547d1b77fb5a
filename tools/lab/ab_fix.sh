#!/bin/bash
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out; export PYTHONDONTWRITEBYTECODE=1
cp easygaussiansplatting_amd/libegs_hip.so /tmp/libegs_keep.so
for r in 1 2; do for so in ab/*.so; do
  cp "$so" easygaussiansplatting_amd/libegs_hip.so
  timeout 200 python tools/lab/skew_dbg.py 2>/dev/null | tail -1 | sed "s#^#$so round $r #"
done; done | tee gpurun_out/ab_fix.txt
cp /tmp/libegs_keep.so easygaussiansplatting_amd/libegs_hip.so
