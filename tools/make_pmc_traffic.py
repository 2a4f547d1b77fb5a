#!/usr/bin/env python3
"""profiles/pmc_traffic.json from tools/pmc_summary.py JSONs:
   python tools/make_pmc_traffic.py <fetch_write.json> <sq_counters.json or -> <ubench_counters.json or -> <out.json> [<valu_mix.json> [<scene>]]

* HBM bytes per launch from the rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes (separate passes);
* VALU issue utilisation per kernel from the SQ passes, CALIBRATED on single-instruction kernels
  (tools/lab/ubench_calib.hip): SQ_ACTIVE_INST_VALU counts 4 cycles for an instruction that really occupies the
  SIMD for 2.5 (add / mul / fma), 4.3 (DPP, cndmask, med3, min/max, cmp, readlane) and 8 for one that takes 8.5
  (exp, rcp, permlane swaps) -- "busy = counter x 4 / cycles" therefore reads 1.59 for a pure-FMA kernel at 100 %
  issue and cannot be quoted as a utilisation.  What is stored instead is the range the counters allow:
  (INSTS - Q) x c + Q x 8.5 cycles with Q = ACTIVE - INSTS quarter-rate instructions and c = 2.5 (every other
  instruction full rate) .. 4.3 (every other one half rate), over the SIMD cycles of the launch;
* `source_hash`: fingerprint of the kernel sources the passes ran on -- bench.py quotes these numbers only when it
  runs the same sources."""
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from bench import kernel_source_hash  # noqa: E402

src, sq, ub, dst = sys.argv[1:5]
# static instruction mix of the draw kernels (tools/valu_mix.py on the ISA of THESE sources): the share of half-rate
# instructions among the non-quarter-rate ones -- turns the band the counters leave into a point estimate
mix = json.load(open(sys.argv[5])) if len(sys.argv) > 5 else {}
MIX_OF = {"k_draw_bwd": "k_draw_bwdILb0ELb1ELb1ELi7ELb0", "k_draw": "k_drawILb0ELb1ELb1ELb1",
          "k_draw_bwd_seg": "k_draw_bwdILb0ELb1ELb1ELi7ELb1", "k_draw_seg": "k_draw_segILb1ELb1ELi0"}


def kname(k):
    """kernel name of the bench tables: the template instances of the segment path are kernels of their own"""
    base = k.replace("egs::", "").split("<")[0]
    args = [x.strip() for x in k[k.find("<") + 1:k.rfind(">")].split(",")] if "<" in k else []
    if base == "k_draw_seg" and args:
        return {"0": "k_draw_seg", "1": "k_draw_seg_fix", "2": "k_draw_seg_compose"}.get(args[-1], base)
    if base == "k_draw_bwd" and len(args) >= 5 and args[4] == "true":
        return "k_draw_bwd_seg"
    return base


d = json.load(open(src))
scene = sys.argv[6] if len(sys.argv) > 6 else "iid"
out = {"scene": scene, "gaussians": 1000000 if scene == "iid" else 1500000, "width": 1920, "height": 1080, "source_hash": kernel_source_hash(),
       "source": "rocprofv3 --kernel-trace --pmc FETCH_SIZE / --pmc WRITE_SIZE (separate passes) and SQ passes, "
                 "tools/profile_step.py, MI355X; tools/collect_profiles.sh",
       "correction": "FETCH_SIZE x2 (gfx950 tallies 128-B requests at 64 B for 16 B/lane loads, MI355X_MICROARCH.md "
                     "section HBM; calibrated on k_preprocess_fwd), WRITE_SIZE x1; counters are in KB",
       "kernels": {}}
for k, v in d.items():
    name = kname(k)
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
if sq != "-":
    for k, c in json.load(open(sq)).items():
        name = kname(k)
        if name not in out["kernels"] or "SQ_ACTIVE_INST_VALU" not in c or "GRBM_GUI_ACTIVE" not in c:
            continue
        insts, active = c["SQ_INSTS_VALU"], c["SQ_ACTIVE_INST_VALU"]
        q = max(0.0, active - insts)
        simd_cycles = 1024 * c["GRBM_GUI_ACTIVE"] / 8
        out["kernels"][name]["valu_insts_per_launch"] = int(insts)
        out["kernels"][name]["valu_busy_counter"] = round(active * 4 / simd_cycles, 3)
        out["kernels"][name]["valu_issue_util"] = [round(((insts - q) * 2.5 + q * 8.5) / simd_cycles, 3),
                                                   round(min(1.0, ((insts - q) * 4.3 + q * 8.5) / simd_cycles), 3)]
        e = out["kernels"][name]
        e["valu_quarter_rate_share"] = round(q / insts, 4) if insts else None
        e["simd_cycles_per_launch"] = int(simd_cycles)
        e["xcd_cycles_per_launch"] = int(c["GRBM_GUI_ACTIVE"] / 8)
        if "duration_ns" in c:       # the clock of the counter pass itself: cycles of one XCD over the kernel's duration
            e["duration_us_in_counter_pass"] = round(c["duration_ns"] / 1e3, 1)
            e["gui_clock_mhz"] = round(c["GRBM_GUI_ACTIVE"] / 8 / c["duration_ns"] * 1e3, 1)
        m = next((v for k, v in mix.items() if MIX_OF.get(name) and MIX_OF[name] in k), None)
        if m:
            h = m["half_share_of_non_quarter"]
            cyc = (insts - q) * (2.5 + 1.8 * h) + q * 8.5
            e["valu_half_rate_share_static"] = h
            e["valu_cycles_per_launch_model"] = int(cyc)
            # point estimate: calibrated cycles of the instructions the counters saw, with the static class mix, over the
            # SIMD cycles of the launch (both in the clock domain of the counters: no MHz enters)
            e["valu_frac"] = round(min(1.0, cyc / simd_cycles), 3)
        for k2 in ("SQ_WAIT_INST_ANY", "SQ_WAVE_CYCLES", "SQ_ACTIVE_INST_SCA", "SQ_ACTIVE_INST_LDS", "SQ_INSTS_SALU",
                   "SQ_INSTS_LDS", "SQ_INSTS_BRANCH"):
            if k2 in c:
                e[k2] = int(c[k2])
if ub != "-":
    cal = {}
    for k, c in json.load(open(ub)).items():
        cal[k.split("<")[-1].rstrip(">")] = round(c["SQ_ACTIVE_INST_VALU"] * 4 / (1024 * c["GRBM_GUI_ACTIVE"] / 8), 3)
    out["valu_busy_counter_at_full_issue"] = {"by_ubench_op_index": cal,
                                              "note": "tools/lab/ubench_calib.hip kernels k_ub<OP>: 0 v_fma, 1 v_exp, 2 v_rcp, "
                                                      "3/4 permlane swaps, 5 dpp add, 6 cndmask, 7 readlane, 8 mov_b64, 9 med3"}
json.dump(out, open(dst, "w"), indent=1)
print({k: (v["hbm_bytes_per_launch"], v.get("valu_issue_util")) for k, v in out["kernels"].items()})
