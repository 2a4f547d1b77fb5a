#!/bin/bash
R=$GRAFT_REPO_ROOT
cd $R; mkdir -p gpurun_out/sweep
for rep in 1 2; do
for cfg in "256 1024" "256 768" "256 512" "128 512" "512 1024"; do
  set -- $cfg
  EGS_SEG_L=$1 EGS_SEG_MIN=$2 timeout 120 python bench.py --scene skewed_reset --steps 30 --cpu-sample 0 --no-ops --no-ring8 2>/dev/null | tail -1 > gpurun_out/sweep/s_$1_$2_$rep.json
  python - <<PY
import json
d=json.loads(open("gpurun_out/sweep/s_$1_$2_$rep.json").read())
k=d["kernels"]
g=lambda n: k.get(n,{}).get("avg_us",0)
print("L=$1 min=$2 rep $rep: step %.4f  seg %.1f fix %.1f compose %.1f plan %.1f bwd %.1f" % (d["ms_per_step"], g("k_draw_seg"), g("k_draw_seg_fix"), g("k_draw_seg_compose"), g("k_seg_plan"), g("k_draw_bwd_seg")))
PY
done; done
