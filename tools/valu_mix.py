#!/usr/bin/env python3
"""Static VALU instruction mix of a kernel, by issue rate, from the ISA hipcc emits (csrc: `hipcc -S` / -save-temps).

    python tools/valu_mix.py <file.s> <kernel-name-substring> [<kernel-name-substring> ...]  ->  JSON on stdout

What it is for: the SQ counters give a kernel's VALU instruction count and the number of quarter-rate
ones (SQ_ACTIVE_INST_VALU - SQ_INSTS_VALU), but not how the rest splits into FULL-rate (2.5 SIMD cycles per wave64
instruction measured, tools/lab/ubench_calib.hip: v_add / v_mul / v_fma / v_sub / v_mad / v_lshl_add ...) and HALF-rate
ones (4.3: DPP forms, v_cndmask, v_med3, v_min / v_max, v_cmp*, v_readlane / v_readfirstlane, v_mov_b64, 64-bit
integer ops).  The draw kernels are one loop; the static share of half-rate instructions among the non-quarter ones of
that loop body narrows the issue-utilisation band the counters leave (0.81 .. 1.0) to a point with a stated error.
Classes follow profiles/r2_ubench_calib.txt / r4_ubench_exec.txt."""
import json
import re
import sys

QUARTER = ("v_exp_", "v_rcp_", "v_rsq_", "v_sqrt_", "v_log_", "v_sin_", "v_cos_", "v_permlane")
HALF_PREFIX = ("v_cndmask", "v_med3", "v_min", "v_max", "v_cmp", "v_readlane", "v_readfirstlane", "v_writelane",
               "v_mov_b64", "v_lshlrev_b64", "v_lshrrev_b64", "v_ashrrev_i64", "v_add_co", "v_addc_co", "v_subb_co",
               "v_mbcnt", "v_bfe", "v_bfi", "v_perm_b32", "v_alignbit", "v_cvt_", "v_mul_lo", "v_mul_hi", "v_mad_u64",
               "v_mad_i64", "v_lshl_add_u64", "v_pk_", "v_bitop3", "v_and_or", "v_or3", "v_xad", "v_swap")


def classify(line):
    op = line.split()[0]
    if not op.startswith("v_") or op.startswith(("v_mfma", "v_smfmac")):
        return None
    if op.startswith(QUARTER):
        return "quarter"
    if "_dpp" in op or " row_" in line or "quad_perm" in line or "row_mirror" in line or "wave_" in line:
        return "half"            # any VALU instruction in its DPP form
    if op.startswith(HALF_PREFIX):
        return "half"
    return "full"


def kernel_body(text, needle):
    m = re.search(r"^(\S*%s\S*):[^\n]*$" % re.escape(needle), text, re.M)
    if not m:
        raise SystemExit("no kernel symbol containing %r" % needle)
    start = m.end()
    end = text.index(".Lfunc_end", start)
    return m.group(1), text[start:end]


def main():
    text = open(sys.argv[1]).read()
    out = {}
    for needle in sys.argv[2:]:
        name, body = kernel_body(text, needle)
        c = {"full": 0, "half": 0, "quarter": 0, "salu": 0, "lds": 0, "vmem": 0}
        for ln in body.splitlines():
            ln = ln.strip()
            if not ln or ln.startswith((".", ";", "//")) or ln.endswith(":"):
                continue
            op = ln.split()[0]
            k = classify(ln)
            if k:
                c[k] += 1
            elif op.startswith("s_") and not op.startswith(("s_waitcnt", "s_nop", "s_barrier", "s_endpgm")):
                c["salu"] += 1
            elif op.startswith("ds_"):
                c["lds"] += 1
            elif op.startswith(("global_", "buffer_", "flat_", "scratch_")):
                c["vmem"] += 1
        valu = c["full"] + c["half"] + c["quarter"]
        c["valu"] = valu
        c["half_share_of_non_quarter"] = round(c["half"] / max(1, c["full"] + c["half"]), 4)
        c["symbol"] = name
        out[needle] = c
    print(json.dumps(out, indent=1))


if __name__ == "__main__":
    main()
