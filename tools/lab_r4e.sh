#!/bin/bash
# round 4, lab E: colour + dcolor/dpw in one pass over the SH row (k_preprocess_fwd 82 -> 60 VGPRs): parity + same-box A/B
mkdir -p gpurun_out/r4e; O=gpurun_out/r4e
timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_raw.py tests/test_gpu_fused_ahead.py tests/test_numeric_diff.py -m gpu -q -x 2>&1 | tail -5 > $O/tests.log
bash tools/ab_bench2.sh 2 > $O/ab.log 2>&1
cp gpurun_out/ab_bench2.txt $O/ab_bench2.txt
tail -3 $O/tests.log; cat $O/ab_bench2.txt | cut -c1-400
