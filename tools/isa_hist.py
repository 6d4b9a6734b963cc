#!/usr/bin/env python3
"""Opcode histogram of one kernel in a hipcc -S --cuda-device-only listing:  isa_hist.py file.s <mangled-name-substring> [--dump]"""
import re, sys
from collections import Counter
src = open(sys.argv[1]).read().split("\n")
pat = sys.argv[2]
start = next(i for i, l in enumerate(src) if re.match(r"^[A-Za-z_][\w.$]*:", l) and pat in l.split(":")[0])
end = next(i for i in range(start, len(src)) if src[i].startswith(".Lfunc_end"))
body = [l.strip() for l in src[start + 1:end]]
ins = [l for l in body if l and not l.startswith((".", ";")) and not l.endswith(":")]
if "--dump" in sys.argv:
    print("\n".join(body))
else:
    c = Counter(l.split()[0] for l in ins)
    print(src[start], len(ins))
    for k, v in c.most_common(40):
        print("%6d %s" % (v, k))
