#!/bin/bash
# same-box A/B of the builds under ab/ on an arbitrary lab script:  gpurun -- 'bash tools/lab/ab_any.sh tools/lab/<script>.py [args]'
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out; export PYTHONDONTWRITEBYTECODE=1
cp easygaussiansplatting_amd/libegs_hip.so /tmp/libegs_keep.so
for r in 1 2; do for so in ab/*.so; do
  cp "$so" easygaussiansplatting_amd/libegs_hip.so
  timeout 300 python "$@" 2>/dev/null | grep -v amdgpu.ids | sed "s#^#$so round $r #"
done; done | tee gpurun_out/ab_any.txt
cp /tmp/libegs_keep.so easygaussiansplatting_amd/libegs_hip.so
