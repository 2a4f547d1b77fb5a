#!/bin/bash
# cache-side counters of the per-Gaussian kernels (one --pmc pass per group, kernel trace only)
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/pmc_tcp; mkdir -p $O; export PYTHONDONTWRITEBYTECODE=1
i=0
for C in "TCP_TCC_READ_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum TCP_TOTAL_READ_sum" "TCC_READ_sum TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum" "TA_BUSY_avr TCP_PENDING_STALL_CYCLES_sum TA_ADDR_STALLED_BY_TC_CYCLES_sum GRBM_GUI_ACTIVE" "TCP_TCC_WRITE_REQ_sum TCC_WRITE_sum TCP_TOTAL_WRITE_sum" "TCP_TCC_READ_REQ_LATENCY_sum TCP_TCP_TA_DATA_STALL_CYCLES_sum TCP_READ_TAGCONFLICT_STALL_CYCLES_sum"; do
  i=$((i+1))
  timeout 300 rocprofv3 --kernel-trace --pmc $C --output-format csv -d /tmp/t$i -- python $R/tools/profile_step.py --steps 3 > /tmp/t$i.log 2>&1
  f=$(find /tmp/t$i -name "*counter_collection.csv" | head -1)
  echo "== $C -> $f"; tail -2 /tmp/t$i.log
  [ -n "$f" ] && python - "$f" <<'PY'
import csv, sys, collections
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for r in csv.DictReader(open(sys.argv[1])):
    k = r["Kernel_Name"].split("(")[0].replace("void egs::", "").replace("egs::", "")[:40]
    acc[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
for k, d in acc.items():
    print("%-42s" % k, {c: round(sum(v) / len(v), 1) for c, v in d.items()}, "launches", len(next(iter(d.values()))))
PY
done 2>&1 | tee $O/tcp_counters.txt
