#!/bin/bash
# Collect the round's profile artefacts on the GPU box (run through gpurun); results land in gpurun_out/prof/.
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
mkdir -p $R/gpurun_out/prof
# the default bench line (with the bounded CPU baseline), the same command under rocprofv3, and the extras
python $R/bench.py > $R/gpurun_out/prof/bench.json 2> $R/gpurun_out/prof/bench.err
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/ks -- python $R/bench.py --cpu-sample 0 > $R/gpurun_out/prof/bench_under_rocprof.json 2>/tmp/ks.err
python $R/bench.py --cpu-sample 0 --extras > $R/gpurun_out/prof/bench_extras.json 2>/dev/null
find /tmp/ks -name "*kernel_stats.csv" -exec cp {} $R/gpurun_out/prof/kernel_stats.csv \;
rocprofv3 --kernel-trace --output-format csv -d /tmp/tr -- python $R/tools/profile_step.py --steps 6 > /tmp/tr.log 2>&1
python $R/tools/trace_timeline.py /tmp/tr > $R/gpurun_out/prof/timeline.txt
# HBM traffic counters (separate passes, no tracing besides kernel-trace)
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d /tmp/p1 -- python $R/tools/profile_step.py --steps 3 > /tmp/p1.log 2>&1
rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d /tmp/p2 -- python $R/tools/profile_step.py --steps 3 > /tmp/p2.log 2>&1
python $R/tools/pmc_summary.py $(dirname $(find /tmp/p1 -name "*counter_collection.csv" | head -1)) $(dirname $(find /tmp/p2 -name "*counter_collection.csv" | head -1)) --all > $R/gpurun_out/prof/pmc_fetch_write.txt
cp /tmp/pmc_summary.json $R/gpurun_out/prof/pmc_fetch_write.json
tail -1 $R/gpurun_out/prof/bench.json | cut -c1-300
tail -1 $R/gpurun_out/prof/bench_under_rocprof.json | cut -c1-200
head -12 $R/gpurun_out/prof/kernel_stats.csv
tail -3 $R/gpurun_out/prof/timeline.txt
