#!/usr/bin/env python3
"""profiles/pmc_traffic.json from a tools/pmc_summary.py JSON (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes).
   python tools/make_pmc_traffic.py gpurun_out/prof/pmc_fetch_write.json profiles/pmc_traffic.json"""
import json
import sys

src, dst = sys.argv[1], sys.argv[2]
d = json.load(open(src))
out = {"gaussians": 1000000, "width": 1920, "height": 1080,
       "source": "rocprofv3 --kernel-trace --pmc FETCH_SIZE / --pmc WRITE_SIZE (separate passes), tools/profile_step.py, "
                 "MI355X; tools/collect_profiles.sh",
       "correction": "FETCH_SIZE x2 (gfx950 tallies 128-B requests at 64 B for 16 B/lane loads, MI355X_MICROARCH.md "
                     "section HBM; calibrated on k_preprocess_fwd: 240 MB algorithmic reads -> FETCH_SIZE 119 MB, "
                     "WRITE_SIZE exact), WRITE_SIZE x1; counters are in KB",
       "kernels": {}}
for k, v in d.items():
    name = k.replace("egs::", "").split("<")[0]
    if "FETCH_SIZE" not in v or "WRITE_SIZE" not in v:
        continue
    e = out["kernels"].setdefault(name, {"FETCH_SIZE_KB": 0.0, "WRITE_SIZE_KB": 0.0, "variants": 0})
    # several template instances / call sites of one kernel name: keep the per-launch mean over all launches
    e["FETCH_SIZE_KB"] += v["FETCH_SIZE"] * v["launches"]
    e["WRITE_SIZE_KB"] += v["WRITE_SIZE"] * v["launches"]
    e["variants"] += v["launches"]
for name, e in out["kernels"].items():
    n = e.pop("variants")
    e["FETCH_SIZE_KB"] /= n
    e["WRITE_SIZE_KB"] /= n
    e["hbm_bytes_per_launch"] = int((2 * e["FETCH_SIZE_KB"] + e["WRITE_SIZE_KB"]) * 1024)
json.dump(out, open(dst, "w"), indent=1)
print({k: v["hbm_bytes_per_launch"] for k, v in out["kernels"].items()})
