#!/usr/bin/env python3
"""Extract one kernel's gfx950 assembly from a -save-temps .s file: kasm.py file.s mangled_substring [out.s]; prints an opcode histogram."""
import collections
import re
import sys

lines = open(sys.argv[1]).read().split("\n")
sub = sys.argv[2]
start = next(i for i, l in enumerate(lines) if sub in l and not l.startswith(".") and not l.startswith("\t") and l.split(";")[0].strip().endswith(":"))
end = next(i for i in range(start, len(lines)) if lines[i].startswith(".Lfunc_end"))
body = lines[start : end + 1]
if len(sys.argv) > 3:
    open(sys.argv[3], "w").write("\n".join(body))
ops = collections.Counter(l.split()[0] for l in body if l.startswith("\t") and re.match(r"\t(v_|s_|global_|ds_|buffer_|scratch_|flat_)", l))
print(lines[start], len(body), "lines")
for k, v in ops.most_common(40):
    print("%6d %s" % (v, k))
