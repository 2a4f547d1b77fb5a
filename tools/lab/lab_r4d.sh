#!/bin/bash
# round 4, lab D: VERDICT r3 task 4(a) priced from the backward side (hit-bit probe)
# (needs a library built with the probe compiled in: cd easygaussiansplatting_amd/csrc && make clean &&
#  make FLAGS="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -munsafe-fp-atomics -fno-slp-vectorize -DEGS_PROBE_HIT_BITS=1")
mkdir -p gpurun_out/r4d; O=gpurun_out/r4d
timeout 600 python tools/lab/bwd_hit_stats.py --time > $O/bwd_hit_bits.txt 2>&1
tail -8 $O/bwd_hit_bits.txt
