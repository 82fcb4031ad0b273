#!/usr/bin/env python3
"""Developer aid: every loop of one kernel with its instruction mix and the VALU time its body costs per trip
according to the measured issue costs (profiles/r02_valu_rates.txt: cycles per wave64 instruction per SIMD).
Usage: tools/isa_loops.py file.hip kernel-substring [-D...]"""
import collections, os, re, subprocess, sys, tempfile

COST = [  # (regex on the opcode, cycles)  first match wins
    (r"v_(rcp|rsq|sqrt|sin|cos|exp|log)_", 8.5),
    (r"v_pk_", 4.45),
    (r"v_(fma|fmac|mul|add|sub|subrev|mac|max|min)_f32", 2.85),
    (r"v_(add|sub|subrev)_u32|v_(add|sub)_co_u32", 2.85),
    (r"v_", 4.7),
]


def cost(op):
    for rx, c in COST:
        if re.match(rx, op):
            return c
    return 0.0


src = os.path.abspath(sys.argv[1])
flt = sys.argv[2]
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
tmp = tempfile.mkdtemp()
subprocess.check_call(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-ffp-contract=off",
                       "-I" + root + "/include", "-I" + root + "/cudasift_amd/csrc", "-Wno-unused-value", "-S",
                       "--cuda-device-only", src, "-o", tmp + "/o.s"] + sys.argv[3:], stderr=subprocess.DEVNULL)
s = open(tmp + "/o.s").read()
for m in re.finditer(r"^(_Z\w+):.*?\n(.*?)\.Lfunc_end", s, re.S | re.M):
    name, body = m.group(1), m.group(2)
    if flt not in name:
        continue
    lines = body.split("\n")
    labels = {}
    for i, l in enumerate(lines):
        mm = re.match(r"^(\.LBB\w+):", l)
        if mm:
            labels[mm.group(1)] = i
    loops = []
    for i, l in enumerate(lines):
        mm = re.match(r"\s+s_cbranch_\w+\s+(\.LBB\w+)", l) or re.match(r"\s+s_branch\s+(\.LBB\w+)", l)
        if mm and mm.group(1) in labels and labels[mm.group(1)] < i:
            loops.append((labels[mm.group(1)], i, mm.group(1)))
    # merge loops with the same header (keep the widest span)
    byhead = {}
    for a, b, lab in loops:
        if lab not in byhead or b > byhead[lab][1]:
            byhead[lab] = (a, b, lab)
    loops = sorted(byhead.values())
    print("==", name[:90])

    def ops_of(a, b, excl=()):
        out = []
        for j in range(a, b + 1):
            if any(x <= j <= y for x, y in excl):
                continue
            l = lines[j]
            if l.startswith("\t") and not l.strip().startswith((".", ";")):
                out.append(l.split()[0])
        return out

    allops = ops_of(0, len(lines) - 1)
    print("   whole kernel: %d instrs, VALU %d" % (len(allops), sum(1 for o in allops if o.startswith("v_"))))
    for a, b, lab in loops:
        depth = sum(1 for x, y, _ in loops if x <= a and b <= y) - 1
        inner = [(x, y) for x, y, _ in loops if a <= x and y <= b and (x, y) != (a, b)]
        ops = ops_of(a, b, inner)
        c = collections.Counter(ops)
        valu = [o for o in ops if o.startswith("v_")]
        cyc = sum(cost(o) for o in valu)
        print("   %s%s lines %d-%d: own body %d instrs: VALU %d (~%.0f cycles), SALU %d, LDS %d, VMEM %d, waitcnt %d, nop %d" % (
            "  " * depth, lab, a, b, len(ops), len(valu), cyc,
            sum(1 for o in ops if o.startswith("s_") and not o.startswith(("s_waitcnt", "s_nop"))),
            sum(1 for o in ops if o.startswith("ds_")), sum(1 for o in ops if o.startswith(("global_", "buffer_", "flat_", "scratch_"))),
            c["s_waitcnt"], c["s_nop"]))
        print("   %s     %s" % ("  " * depth, ", ".join("%s %d" % kv for kv in c.most_common(12) if kv[0].startswith("v_"))))
