#!/usr/bin/env python3
"""VGPRs / scratch bytes / LDS bytes per kernel from a `hipcc -S` listing: tools/kernel_regs.py file.s"""
import re, sys
s = open(sys.argv[1]).read()
for m in re.finditer(r"\.amdhsa_kernel (\S+)(.*?)\.end_amdhsa_kernel", s, re.S):
    body = m.group(2)
    g = lambda k: re.search(r"\.amdhsa_" + k + r" (\d+)", body)
    name = m.group(1)
    print(f"{name[:70]:70s} vgpr {g('next_free_vgpr').group(1):>4s} scratch {g('private_segment_fixed_size').group(1):>5s} lds {g('group_segment_fixed_size').group(1):>6s}")
