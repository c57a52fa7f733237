#!/usr/bin/env python3
"""Static instruction mix per kernel from a `hipcc -S` listing: tools/isa_mix.py file.s [name-substring ...]"""
import collections
import re
import sys

s = open(sys.argv[1]).read()
want = sys.argv[2:]
parts = re.split(r"\n(?=_Z[\w]+:)", s)
for f in parts:
    m = re.match(r"(_Z\w+):", f)
    if not m or ".amdhsa_kernel" not in f and "s_endpgm" not in f:
        continue
    name = m.group(1)
    if want and not any(w in name for w in want):
        continue
    body = f.split("s_endpgm")[0]
    ins = [l.strip().split()[0] for l in body.split("\n") if l.startswith("\t") and l.strip() and not l.strip().startswith((".", ";"))]
    c = collections.Counter(ins)
    g = collections.Counter()
    for k, v in c.items():
        if "f64" in k: grp = "f64"
        elif any(x in k for x in ("v_div", "v_rcp", "v_sqrt", "v_rsq")): grp = "div/rcp/sqrt"
        elif k.startswith("s_load"): grp = "s_load"
        elif k.startswith(("global", "flat", "scratch", "buffer")): grp = "vmem"
        elif k.startswith("ds_"): grp = "lds"
        elif k.startswith("v_"): grp = "valu"
        elif k.startswith("s_"): grp = "salu"
        else: grp = "other"
        g[grp] += v
    meta = re.search(r"\.vgpr_count:\s*(\d+)", f)
    print(f"{name}: {len(ins)} instructions  {dict(g)}")
    print("   ", c.most_common(18))
