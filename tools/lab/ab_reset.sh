#!/bin/bash
# A/B of two builds on ONE box, alternating: ab/old (here, before the call:  mkdir -p ab/old && git archive <rev> | tar -x -C ab/old
# && make -C ab/old/easygaussiansplatting_amd/csrc ; ab/ is git-ignored but travels with gpurun) against the working tree.
#   gpurun --timeout 600 -- 'timeout 500 bash tools/lab/ab_reset.sh'
R=$GRAFT_REPO_ROOT
mkdir -p $R/gpurun_out/ab
for i in 1 2 3; do
  for w in old new; do
    D=$R; [ $w = old ] && D=$R/ab/old
    (cd $D && timeout 120 python bench.py --scene skewed_reset --steps 30 --cpu-sample 0 --no-ops --no-ring8 2>/dev/null | tail -1 > $R/gpurun_out/ab/${w}_$i.json)
    python - <<PY
import json
d=json.loads(open("$R/gpurun_out/ab/${w}_$i.json").read())
k=d["kernels"]
f=sum(k[n]["avg_us"]*k[n]["launches_per_step"] for n in k if n.startswith("k_draw_seg") or n in ("k_seg_report",))
pl=k["k_seg_plan"]["avg_us"]*k["k_seg_plan"]["launches_per_step"]
print("$w $i step %.4f ms  fwd-seg kernels %.1f us  plan %.1f us  bwd_seg %.1f" % (d["ms_per_step"], f, pl, k["k_draw_bwd_seg"]["avg_us"]))
PY
  done
done
