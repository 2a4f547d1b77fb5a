#!/bin/bash
# round 4, lab K: every Gaussian's gradient at 1 M / 1080p against the all-tile oracle (parallel on the host cores)
mkdir -p gpurun_out/r4k; O=gpurun_out/r4k; rm -f $O/grad_stats.jsonl
nproc > $O/nproc.txt; free -g | head -2 >> $O/nproc.txt
( time EGS_GRAD_STATS=$O/grad_stats.jsonl EGS_GRAD_STATS_ONLY=1 timeout 1500 python -m pytest tests/test_gpu_parity.py -m gpu -q -k "every_gaussian" 2>&1 | tail -15 ) > $O/tests.log 2>&1
cat $O/nproc.txt; tail -20 $O/tests.log; cat $O/grad_stats.jsonl | cut -c1-300
