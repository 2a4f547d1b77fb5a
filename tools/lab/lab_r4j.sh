#!/bin/bash
# round 4, lab J: k_draw_bwd with the two blocks of a tile row as one straight-line body (RED = 15) against RED = 7
mkdir -p gpurun_out/r4j; O=gpurun_out/r4j
cp easygaussiansplatting_amd/libegs_hip.so /tmp/keep.so
run() { env "$@" timeout 200 python bench.py --cpu-sample 0 --steps 30 --no-ops --no-ring8 2>/dev/null | tail -1 > /tmp/b.json; python - "$*" <<'PY'
import json, sys
d = json.load(open("/tmp/b.json"))
print(sys.argv[1], "ms/step %.4f" % d["ms_per_step"], "k_draw_bwd %.1f" % d["kernels"]["k_draw_bwd"]["avg_us"], flush=True)
PY
}
EGS_DRAWB_RED=15 timeout 600 python -m pytest tests/test_gpu_parity.py -m gpu -q -x -k "splat_10k or ragged or dense or needles or adversarial or g5 or gsfunction_fused or full_size_fused" 2>&1 | tail -3 > $O/tests.log
for rep in 1 2; do
  for so in ab/a_w5.so ab/b_w4.so; do
    cp $so easygaussiansplatting_amd/libegs_hip.so
    run LIB=$so EGS_DRAWB_RED=7
    run LIB=$so EGS_DRAWB_RED=15
  done
done | tee $O/pair.txt
cp /tmp/keep.so easygaussiansplatting_amd/libegs_hip.so
cat $O/tests.log
