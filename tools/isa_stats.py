#!/usr/bin/env python3
"""Static instruction mix of the kernels in a hipcc -S listing (gfx950): tools/isa_stats.py file.s [name-filter]"""
import re
import sys
from collections import Counter

t = open(sys.argv[1]).read()
flt = sys.argv[2] if len(sys.argv) > 2 else ""
labels = [(m.start(), m.group(1)) for m in re.finditer(r"^(_Z\w+):", t, re.M)]
for pos, name in labels:
    if flt not in name:
        continue
    end = t.find("s_endpgm", pos)
    body = t[pos:end]
    ins = [l.strip().split()[0] for l in body.split("\n")[1:]
           if l.strip() and not l.strip().startswith((";", ".")) and not l.strip().split()[0].endswith(":")]
    c = Counter(ins)
    grp = lambda p: sum(v for k, v in c.items() if k.startswith(p))
    print("%s\n   total %d  valu %d  ds %d  salu %d  vmem %d" % (name[:70], len(ins), grp("v_"), grp("ds_"), grp("s_"),
                                                               grp("global_") + grp("buffer_") + grp("flat_")))
    print("   ", c.most_common(18))
