#!/usr/bin/env python
"""Static instruction mix of the loops of one kernel in an ISA listing (droid-slam_amd/build/*.s): every backward branch closes a loop;
prints SALU / VALU / LDS / VMEM / MFMA counts of the instructions between the label and the branch.
usage: python scripts/loop_mix.py droid-slam_amd/build/corr_pyramid.s <mangled-name substring> [mfma count to filter on]"""
import re, sys
txt = open(sys.argv[1]).read()
for m in re.finditer(r"\.amdhsa_kernel (\S+)", txt):
    name = m.group(1)
    if sys.argv[2] not in name:
        continue
    i = txt.index(name + ":"); j = txt.index(".end_amdhsa_kernel", i)
    lines = [l.strip() for l in txt[i:j].split("\n")]
    labels = {mm.group(1): n for n, l in enumerate(lines) for mm in [re.match(r"^(\.LBB\d+_\d+):", l)] if mm}
    vg = re.search(re.escape(name) + r"\.num_vgpr, (\d+)", txt)
    print(name, "vgpr", vg and vg.group(1))
    for n, l in enumerate(lines):
        mm = re.match(r"^s_c?branch\S*\s+(\.LBB\d+_\d+)", l)
        if mm and mm.group(1) in labels and labels[mm.group(1)] < n:
            seg = [x for x in lines[labels[mm.group(1)]:n + 1] if x and not x.startswith((".", ";"))]
            cat = {}
            for x in seg:
                op = x.split()[0]
                c = "mfma" if "mfma" in op else "lds" if op.startswith("ds_") else "vmem" if op.startswith(("global_", "buffer_", "flat_")) else "salu" if op.startswith("s_") else "valu"
                cat[c] = cat.get(c, 0) + 1
            if len(sys.argv) < 4 or cat.get("mfma", 0) == int(sys.argv[3]):
                print("  loop %-10s lines %5d..%5d  %4d instructions  %s" % (mm.group(1), labels[mm.group(1)], n, len(seg), cat))
