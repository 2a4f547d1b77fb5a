#!/usr/bin/env python3
"""Summarise rocprofv3 --pmc CSVs: per kernel, per-launch averages of each counter."""
import collections, csv, glob, json, sys
res = collections.defaultdict(dict)
for d in sys.argv[1:]:
    for f in glob.glob(d + "/*counter_collection.csv"):
        agg = collections.defaultdict(lambda: collections.defaultdict(float)); disp = collections.defaultdict(set)
        for r in csv.DictReader(open(f)):
            k = r["Kernel_Name"].split("(")[0].replace("void ", "")
            agg[k][r["Counter_Name"]] += float(r["Counter_Value"]); disp[k].add(r["Dispatch_Id"])
        for k, v in agg.items():
            if "egs::" in k or "--any" in sys.argv:
                for c, x in v.items():
                    res[k][c] = x / len(disp[k])
                res[k]["launches"] = len(disp[k])
# kernel durations of the same passes (rocprofv3 --kernel-trace writes them next to the counters): what the cycle
# counters of a pass are divided by to get the clock of THAT pass
for d in sys.argv[1:]:
    for f in glob.glob(d + "/*kernel_trace.csv"):
        dur = collections.defaultdict(list)
        for r in csv.DictReader(open(f)):
            k = r["Kernel_Name"].split("(")[0].replace("void ", "")
            dur[k].append(float(r["End_Timestamp"]) - float(r["Start_Timestamp"]))
        for k, v in dur.items():
            if k in res:
                res[k].setdefault("duration_ns", sum(v) / len(v))
if __name__ == "__main__":
    for k, v in res.items():
        if "--all" in sys.argv or "draw" in k:
            print(k)
            for c in sorted(v):
                print("   %-28s %16.0f" % (c, v[c]))
    json.dump(res, open("/tmp/pmc_summary.json", "w"), indent=1)
