#!/usr/bin/env python3
"""Per-kernel register / scratch / LDS usage of a HIP translation unit compiled for gfx950 (from the -save-temps assembly):
kernel_resources.py file.hip [extra hipcc flags...] -> name, VGPRs, AGPRs, SGPRs, spilled SGPRs / VGPRs, scratch bytes, static LDS bytes."""
import os
import re
import subprocess
import sys
import tempfile

src = os.path.abspath(sys.argv[1])
tmp = tempfile.mkdtemp(prefix="kres_")
cmd = ["hipcc", "-O3", "-std=c++17", "--offload-arch=gfx950", "-fPIC", "-ffp-contract=fast", "-save-temps", "-c", src, "-o", os.path.join(tmp, "o.o")] + sys.argv[2:]
subprocess.run(cmd, cwd=tmp, check=True, stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
asm = [f for f in os.listdir(tmp) if f.endswith("gfx950.s")][0]
text = open(os.path.join(tmp, asm)).read()
meta = text[text.index("amdhsa.kernels:"):]
rows = []
for blk in meta.split("  - .agpr_count:")[1:]:
    def g(key):
        m = re.search(r"\.%s:\s+(\S+)" % key, blk)
        return m.group(1) if m else "?"
    name = subprocess.run(["c++filt", g("name")], capture_output=True, text=True).stdout.strip()
    name = re.sub(r"\(.*", "", name)
    rows.append((name, g("vgpr_count"), blk.split()[0], g("sgpr_count"), g("sgpr_spill_count"), g("vgpr_spill_count"), g("private_segment_fixed_size"), g("group_segment_fixed_size")))
print("%-86s %5s %5s %5s %7s %7s %8s %7s" % ("kernel", "vgpr", "agpr", "sgpr", "s-spill", "v-spill", "scratch", "lds"))
for r in sorted(rows):
    print("%-86s %5s %5s %5s %7s %7s %8s %7s" % (r[0][-86:], r[1], r[2], r[3], r[4], r[5], r[6], r[7]))
