#!/usr/bin/env python
"""Prints the register / spill figures of the k_primary instantiations from a hipcc -Rpass-analysis=kernel-resource-usage log."""
import re, sys
txt = open(sys.argv[1]).read()
for b in re.split(r"remark: [^\n]*Function Name: ", txt)[1:]:
    name = b.split('\n')[0]
    if 'k_primary' not in name:
        continue
    m = re.search(r"k_primaryILb(\d)ELi(\d+)ELb(\d)ELi(\d)", name)
    g = lambda k: (re.search(k + r": (\d+)", b) or [None, '?'])[1]
    print("k_primary<stats=%s, feat=%s, plain=%s, occ=%s>  VGPR %s  spillV %s  SGPR spill %s  scratch %s B  occ %s" % (
        m.group(1), m.group(2), m.group(3), m.group(4), g("VGPRs"), g("VGPRs Spill"), g("SGPRs Spill"), g(r"ScratchSize \[bytes/lane\]"), g(r"Occupancy \[waves/SIMD\]")))
