#!/usr/bin/env python3
"""Timeline of the last profiled step from a rocprofv3 --kernel-trace CSV: per-kernel duration and the
idle gap before each kernel.   python tools/trace_timeline.py <dir with *_kernel_trace.csv>"""
import csv
import glob
import os
import sys

root = sys.argv[1]
files = glob.glob(os.path.join(root, "**", "*kernel_trace.csv"), recursive=True)
rows = []
for f in files:
    with open(f) as fh:
        for r in csv.DictReader(fh):
            rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"]))
rows.sort()
# the last step = from the last k_preprocess_fwd on
# (with the two-stream forward the side-stream part is k_preprocess_fwd<.., 2>: not a step boundary)
starts = [i for i, r in enumerate(rows) if "k_preprocess_fwd" in r[2] and ", 2>(" not in r[2]]
if len(starts) < 2:      # the seven-op surface: a step starts with project
    starts = [i for i, r in enumerate(rows) if "k_project" in r[2]]
if "--by-bwd" in sys.argv:   # (two-stream forward: k_preprocess_fwd runs twice per step) a step ends with the chain rule
    ends = [i for i, r in enumerate(rows) if "k_preprocess_bwd" in r[2]]
    starts = [ends[-3] + 1, ends[-2] + 1]
i0, i1 = starts[-2], starts[-1]
step = rows[i0:i1]
t0 = step[0][0]
busy = 0
prev_end = None
print("%-46s %9s %9s %9s" % ("kernel", "start_us", "dur_us", "gap_us"))
for s, e, name in step:
    gap = 0 if prev_end is None else (s - prev_end) / 1e3
    print("%-46s %9.1f %9.1f %9.1f" % (name[:46], (s - t0) / 1e3, (e - s) / 1e3, gap))
    busy += e - max(s, prev_end or s) if e > (prev_end or 0) else 0   # concurrent launches count once
    prev_end = max(e, prev_end or e)
span = (rows[i1][0] - t0) / 1e3
print("step span %.1f us, kernel busy %.1f us, idle %.1f us, launches %d" % (span, busy / 1e3, span - busy / 1e3, len(step)))
