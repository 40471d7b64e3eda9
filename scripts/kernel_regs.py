#!/usr/bin/env python
"""Register / scratch / LDS use of the kernels in an ISA listing produced by the in-tree build
(droid-slam_amd/build/*.s, written by build.py's audits or `hipcc -S --cuda-device-only`).
usage: python scripts/kernel_regs.py droid-slam_amd/build/conv.s [name-substring]"""
import re
import subprocess
import sys

txt = open(sys.argv[1]).read()
pat = sys.argv[2] if len(sys.argv) > 2 else ""
for m in re.finditer(r"\.amdhsa_kernel (\S+)(.*?)\.end_amdhsa_kernel", txt, re.S):
    name, body = m.group(1), m.group(2)
    if pat not in name:
        continue
    g = lambda k: (re.search(r"\.amdhsa_%s (\S+)" % k, body) or [None, "?"])[1]
    try:
        dn = subprocess.run(["/opt/rocm/lib/llvm/bin/llvm-cxxfilt", name], stdout=subprocess.PIPE, text=True).stdout.strip()
    except OSError:
        dn = name
    sc = re.search(re.escape(name) + r"\.private_seg_size, (\d+)", txt)
    vg = re.search(re.escape(name) + r"\.num_vgpr, (\d+)", txt)
    ag = re.search(re.escape(name) + r"\.num_agpr, (\d+)", txt)
    sg = re.search(re.escape(name) + r"\.numbered_sgpr, (\d+)", txt)
    print("%-110s vgpr %s agpr %s sgpr %s scratch %s lds %s" % (dn[:110], vg and vg.group(1), ag and ag.group(1), sg and sg.group(1),
                                                                sc and sc.group(1), g("group_segment_fixed_size")))
