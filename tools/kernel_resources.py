#!/usr/bin/env python3
"""Compiles search.hip for gfx950 with -save-temps and prints the resources of every kernel (VGPRs, SGPRs, LDS, scratch,
spills): what decides how many waves a CU holds.  Usage: python tools/kernel_resources.py [name-filter] [-D...]"""
import os, re, subprocess, sys, tempfile
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
flt = [a for a in sys.argv[1:] if not a.startswith("-")]
defs = [a for a in sys.argv[1:] if a.startswith("-")]
d = tempfile.mkdtemp()
subprocess.check_call(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-c", "-save-temps=obj", "-o", os.path.join(d, "s.o"),
                       os.path.join(ROOT, "kaamer_amd", "csrc", "search.hip")] + defs, cwd=d, stderr=subprocess.DEVNULL)
s = open([os.path.join(d, f) for f in os.listdir(d) if f.endswith("gfx950.s")][0]).read()
for body in re.split(r"\n  - \.agpr_count:|\n  - \.args:", s)[1:]:
    m = re.search(r"\.name:\s+(\S+)", body)
    if not m:
        continue
    name = m.group(1)
    if flt and not any(f in name for f in flt):
        continue
    g = lambda k: (re.search(r"\." + k + r":\s+(\d+)", body) or [None, "?"])[1]
    print("%-60s vgpr %4s sgpr %4s lds %6s scratch %4s spills v%s s%s" % (name[:60], g("vgpr_count"), g("sgpr_count"), g("group_segment_fixed_size"),
                                                                   g("private_segment_fixed_size"), g("vgpr_spill_count"), g("sgpr_spill_count")))
