#!/bin/bash
# round 4, lab F: k_sh2color restructured (basis row out first, one pass for colour + dcolor/ddir): parity, op timing per
# library variant; k_preprocess_fwd under an occupancy cap (dynamic LDS pad)
mkdir -p gpurun_out/r4f; O=gpurun_out/r4f
timeout 600 python -m pytest tests/test_gpu_parity.py -m gpu -q -x -k "sh_degrees or stages_vs_oracle or backward_gpu_script or gsfunction_fused or chain_rule" 2>&1 | tail -4 > $O/tests.log
cp easygaussiansplatting_amd/libegs_hip.so /tmp/keep.so
for so in ab/a_old.so ab/b_new.so ab/c_new6.so ab/a_old.so ab/b_new.so ab/c_new6.so; do
  cp $so easygaussiansplatting_amd/libegs_hip.so
  echo "== $so"; timeout 200 python tools/time_ops.py 2>&1 | grep sh2Color
done > $O/sh2color.txt 2>&1
cp /tmp/keep.so easygaussiansplatting_amd/libegs_hip.so
for rep in 1 2; do for pad in 0 16384 24000 40000; do
  EGS_PRE_LDS_PAD=$pad timeout 200 python bench.py --cpu-sample 0 --steps 30 --no-ops --no-ring8 2>/dev/null | tail -1 > /tmp/b.json
  python - $pad <<'PY'
import json, sys
d = json.load(open("/tmp/b.json"))
print("pad", sys.argv[1], "ms/step %.4f" % d["ms_per_step"], "k_preprocess_fwd %.1f" % d["kernels"]["k_preprocess_fwd"]["avg_us"], "fwd_only %.4f" % d["fwd_only"]["ms"], flush=True)
PY
done; done > $O/pre_pad.txt 2>&1
cat $O/tests.log $O/sh2color.txt $O/pre_pad.txt
