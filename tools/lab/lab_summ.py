#!/usr/bin/env python3
"""One-line summary of a bench.py JSON line (last line of the given file): tools/lab_*.sh use it."""
import json
import sys

for fn in sys.argv[1:]:
    try:
        line = open(fn).read().strip().splitlines()[-1]
        d = json.loads(line)
    except Exception as e:
        print(fn, "unreadable:", e)
        continue
    k = d.get("kernels", {})
    ku = lambda n: k.get(n, {}).get("ms_per_step")
    sort_ms = sum(k.get(n, {}).get("ms_per_step", 0) for n in ("k_radix_hist", "k_radix_rowscan", "k_radix_scatter"))
    print("%-28s ms %.4f busy %s ratio %s fwd %.4f | draw %s bwd %s pre_f %s pre_b %s sort %.4f emit %s scan %s order %s | ops %s redone %s"
          % (fn.split("/")[-1], d["ms_per_step"], d.get("gpu_busy_ms_per_step"), d.get("wall_over_gpu_busy"),
             d["fwd_only"]["ms"], ku("k_draw"), ku("k_draw_bwd"), ku("k_preprocess_fwd"), ku("k_preprocess_bwd"),
             sort_ms, ku("k_bin_emit"),
             round((ku("k_scan_partials") or 0) + (ku("k_scan_apply") or 0), 4), ku("k_tile_order"),
             d.get("ops_ms_per_step"), d.get("redone_steps")))
