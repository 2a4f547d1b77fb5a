#!/bin/bash
# the kernel-stats piece of collect_profiles.sh alone
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/prof; mkdir -p $O; export PYTHONDONTWRITEBYTECODE=1
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/ks -- python $R/bench.py --cpu-sample 0 --no-ops --no-ring8 > $O/bench_under_rocprof.json 2>/tmp/ks.err
find /tmp/ks -name "*kernel_stats.csv" -exec cp {} $O/kernel_stats.csv \;
head -8 $O/kernel_stats.csv | cut -c1-60,200-330
tail -1 $O/bench_under_rocprof.json | cut -c1-250
