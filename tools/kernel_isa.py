#!/usr/bin/env python3
"""Per-kernel ISA digests of csrc/*.hip (gfx950), for refactors that must not change a kernel.

    python tools/kernel_isa.py dump  out.json     # {demangled kernel name: sha1 of its normalised instruction stream}
    python tools/kernel_isa.py diff  a.json b.json

Every .hip file is compiled device-only to assembly (`hipcc -S --cuda-device-only`, the Makefile's flags); a kernel's
text is what lies between its label and its `.Lfunc_end`; local labels (`.LBB<function>_<block>`) are renumbered
by function so that moving a kernel to another translation unit changes nothing.  Registers, metadata
(`.vgpr_count`, LDS size) and the instruction text are all part of the digest."""
import hashlib
import json
import os
import re
import subprocess
import sys
import tempfile

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(REPO, "easygaussiansplatting_amd", "csrc")
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-munsafe-fp-atomics", "-fno-slp-vectorize",
         "-Wno-unused-function"]


def demangle(names):
    out = subprocess.run(["c++filt"], input="\n".join(names), capture_output=True,
                         text=True).stdout.split("\n")
    return dict(zip(names, out))


def kernels_of(asm):
    """{mangled name: normalised text} for every function with an .amdhsa_kernel descriptor"""
    kern = set(re.findall(r"^\s*\.amdhsa_kernel\s+(\S+)", asm, re.M))
    res = {}
    for name in kern:
        m = re.search(r"^%s:[^\n]*\n(.*?)^\s*\.section\s+\.rodata" % re.escape(name), asm, re.M | re.S)
        if not m:
            continue
        body = m.group(1)
        body = re.sub(r"\.LBB\d+_(\d+)", r".LBB_\1", body)
        body = re.sub(r"\.Lfunc_(begin|end)\d+", r".Lfunc_\1", body)
        body = re.sub(r"\.Ltmp\d+", ".Ltmp", body)
        body = "\n".join(ln.split(";")[0].rstrip() for ln in body.split("\n") if ln.strip() and not ln.strip().startswith(";"))
        d = re.search(r"^\s*\.amdhsa_kernel\s+%s\s*\n(.*?)\.end_amdhsa_kernel" % re.escape(name), asm, re.M | re.S)
        desc = "\n".join(ln.strip() for ln in d.group(1).split("\n")
                         if any(k in ln for k in ("next_free_vgpr", "next_free_sgpr", "group_segment_fixed_size",
                                                  "private_segment_fixed_size", "accum_offset"))) if d else ""
        res[name] = body + "\n" + desc
    return res


def dump(out_path, sources=None):
    srcs = sources or sorted(f for f in os.listdir(CSRC) if f.endswith(".hip"))
    table = {}
    with tempfile.TemporaryDirectory() as td:
        for f in srcs:
            s = os.path.join(td, f + ".s")
            subprocess.run(["/opt/rocm/bin/hipcc", *FLAGS, "-S", "--cuda-device-only", os.path.join(CSRC, f), "-o", s],
                           check=True)
            ks = kernels_of(open(s).read())
            dm = demangle(list(ks))
            for k, body in ks.items():
                nm = dm[k]
                assert nm not in table, "kernel defined twice: " + nm
                table[nm] = {"sha1": hashlib.sha1(body.encode()).hexdigest(), "lines": body.count("\n"), "file": f}
    json.dump(table, open(out_path, "w"), indent=1, sort_keys=True)
    print("%d kernels -> %s" % (len(table), out_path))


def diff(a, b):
    A, B = json.load(open(a)), json.load(open(b))
    bad = 0
    for k in sorted(set(A) | set(B)):
        if k not in A: print("only in b:", k); bad += 1
        elif k not in B: print("only in a:", k); bad += 1
        elif A[k]["sha1"] != B[k]["sha1"]:
            print("DIFFERENT (%d vs %d lines): %s" % (A[k]["lines"], B[k]["lines"], k)); bad += 1
    print("%d kernels compared, %d differ" % (len(set(A) | set(B)), bad))
    return 1 if bad else 0


if __name__ == "__main__":
    if sys.argv[1] == "dump":
        dump(sys.argv[2], sys.argv[3:] or None)
    else:
        sys.exit(diff(sys.argv[2], sys.argv[3]))
