#!/usr/bin/env python3
"""Developer aid: compile one .hip file for gfx950 and print, per kernel, the register usage and the
instruction mix of its hottest (largest backward-branch) loop body.  Usage: tools/isa_loop.py file.hip [kernel-substring]"""
import collections, os, re, subprocess, sys, tempfile
src = os.path.abspath(sys.argv[1])
flt = sys.argv[2] if len(sys.argv) > 2 else ""
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
tmp = tempfile.mkdtemp()
subprocess.check_call(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-ffp-contract=off",
                       "-I" + root + "/include", "-I" + root + "/cudasift_amd/csrc", "-Wno-unused-value", "-save-temps",
                       "-c", src, "-o", tmp + "/o.o"], cwd=tmp, stderr=subprocess.DEVNULL)
asm = [f for f in os.listdir(tmp) if f.endswith("gfx950.s")][0]
s = open(os.path.join(tmp, asm)).read()
for m in re.finditer(r"^(_Z\w+):.*?\n(.*?)\.Lfunc_end", s, re.S | re.M):
    name, body = m.group(1), m.group(2)
    if flt not in name or "kernel" not in name:
        continue
    lines = body.split("\n")
    labels = {}
    for i, l in enumerate(lines):
        mm = re.match(r"^(\.LBB\w+):", l)
        if mm:
            labels[mm.group(1)] = i
    best = None
    for i, l in enumerate(lines):
        mm = re.match(r"\s+s_cbranch_\w+\s+(\.LBB\w+)", l)
        if mm and mm.group(1) in labels and labels[mm.group(1)] < i:
            span = (labels[mm.group(1)], i)
            if best is None or span[1] - span[0] > best[1] - best[0]:
                best = span
    vg = re.search(re.escape(name) + r".*?\.vgpr_count:\s+(\d+)", s, re.S)
    ag = re.search(r"\.amdhsa_accum_offset", body)
    print("==", name[:60])
    meta = re.search(r"\.name:\s+" + re.escape(name) + r"\n(.*?)\.wavefront_size", s, re.S)
    if meta:
        for k in ("sgpr_count", "vgpr_count", "agpr_count", "sgpr_spill_count", "vgpr_spill_count", "group_segment_fixed_size"):
            mm = re.search(r"\." + k + r":\s+(\d+)", meta.group(1))
            if mm:
                print("   %s=%s" % (k, mm.group(1)), end="")
        print()
    if best:
        ops = [l.split()[0] for l in lines[best[0]:best[1] + 1] if l.startswith("\t") and not l.strip().startswith((".", ";"))]
        c = collections.Counter(ops)
        valu = sum(v for k, v in c.items() if k.startswith("v_"))
        print("   hottest loop: %d instrs, VALU %d, SALU %d, vmem %d, lds %d" % (
            len(ops), valu, sum(v for k, v in c.items() if k.startswith("s_")),
            sum(v for k, v in c.items() if k.startswith(("global_", "buffer_", "flat_"))),
            sum(v for k, v in c.items() if k.startswith("ds_"))))
        print("   ", c.most_common(14))
