#!/bin/bash
# rocprofv3 kernel trace of the fused loss kernels (run on the GPU box via gpurun)
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/lp
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/lp -- python $GRAFT_REPO_ROOT/tools/lab/time_hip_loss.py > /tmp/lp.log 2>&1
grep "hip gau_loss" /tmp/lp.log
f=$(find /tmp/lp -name "*kernel_stats.csv" | head -1)
python - "$f" <<PY
import csv, sys
for r in csv.DictReader(open(sys.argv[1])):
    if "ssim" in r["Name"] or "finalize" in r["Name"]:
        print("%-28s calls %s avg %.1f us min %.1f us" % (r["Name"][:28], r["Calls"], float(r["AverageNs"]) / 1e3, float(r["MinNs"]) / 1e3))
PY
