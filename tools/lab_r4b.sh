#!/bin/bash
# round 4, lab B: the GPU suite under the relative gradient rule, every number dumped (nothing stops at a failure)
mkdir -p gpurun_out/r4b; O=gpurun_out/r4b
rm -f $O/grad_stats.jsonl
EGS_GRAD_STATS=$O/grad_stats.jsonl EGS_GRAD_STATS_ONLY=1 timeout 1200 python -m pytest tests -m gpu -q 2>&1 | tail -60 > $O/tests.log
tail -8 $O/tests.log
