#!/bin/bash
# round 4, lab D: VERDICT r3 task 4(a) priced from the backward side (hit-bit probe)
mkdir -p gpurun_out/r4d; O=gpurun_out/r4d
timeout 600 python tools/bwd_hit_stats.py --time > $O/bwd_hit_bits.txt 2>&1
tail -8 $O/bwd_hit_bits.txt
